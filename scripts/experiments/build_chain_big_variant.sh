#!/bin/bash
# bash scripts/experiments/build_chain_big_variant.sh <name> "<extra -D flags>" [source file]
# A/B builds of ONE kernel file: libswn_hip_<name>.so = the default build's objects with chain_big.o recompiled from [source file]
# (default: the tree's chain_big.hip) under the extra flags.  Select at run time with SWN_LIB=$PWD/switch_nerf_amd/libswn_hip_<name>.so.
set -e
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
NAME=$1; DEFS=$2; SRC=${3:-$ROOT/switch_nerf_amd/csrc/chain_big.hip}
OBJ=$ROOT/switch_nerf_amd/build; VOBJ=$ROOT/switch_nerf_amd/build_$NAME
[ -f $OBJ/route.o ] || { echo "run switch_nerf_amd/build.sh first"; exit 1; }
mkdir -p $VOBJ
/opt/rocm/bin/hipcc $DEFS --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-variable -Wno-unused-but-set-variable \
  -ffp-contract=fast-honor-pragmas -I$ROOT/switch_nerf_amd/csrc -c $SRC -o $VOBJ/chain_big.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJ/{elementwise,gate_mfma,route,chain,chain_wide,chain_wide2,chain_cat,wgrad,sampling,mip,bounds,hashgrid,rayops}.o \
  $VOBJ/chain_big.o -o $ROOT/switch_nerf_amd/libswn_hip_$NAME.so
echo "built libswn_hip_$NAME.so"
