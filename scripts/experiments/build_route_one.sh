#!/bin/bash
# libswn_hip_routeone.so = the default build's objects with route.o recompiled under -DSWN_EXP_ROUTE_ONE (swn_route_top1x modes 1 / 2:
# scripts/experiments/route_one.inc).  SWN_LIB=$PWD/switch_nerf_amd/libswn_hip_routeone.so SWN_ROUTE_MODE=2 python bench.py ...
set -e
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
OBJ=$ROOT/switch_nerf_amd/build; VOBJ=$ROOT/switch_nerf_amd/build_routeone
mkdir -p $VOBJ
/opt/rocm/bin/hipcc -DSWN_EXP_ROUTE_ONE --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-variable -Wno-unused-but-set-variable \
  -ffp-contract=fast-honor-pragmas -c $ROOT/switch_nerf_amd/csrc/route.hip -o $VOBJ/route.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJ/{elementwise,gate_mfma,chain,chain_big,chain_wide,chain_wide2,chain_cat,wgrad,sampling,mip,bounds,hashgrid,rayops}.o \
  $VOBJ/route.o -o $ROOT/switch_nerf_amd/libswn_hip_routeone.so
echo "built libswn_hip_routeone.so"
