#!/bin/bash
# Build libswn_hip.so (fp32 + bf16) and libswn_hip_f16.so (fp32 + fp16: the same sources with -DSWN_HALF_F16) for gfx950, in-tree.
# hipcc cross-compiles without a GPU.  SWN_ONLY=bf16 skips the fp16 library (kernel experiments).
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
cd "$HERE/csrc"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
BASEFLAGS="${SWN_DEFS:-} --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-variable -Wno-unused-but-set-variable -ffp-contract=fast-honor-pragmas"
build_one() {   # $1 = object directory, $2 = extra flags, $3 = output library
  local OBJ="$HERE/$1" FLAGS="$BASEFLAGS $2" OUT="$HERE/$3"
  mkdir -p "$OBJ"
  local pids=()
  newer() { [ ! -f "$2" ] || [ "$1" -nt "$2" ] || [ common.hpp -nt "$2" ] || [ pe_store.hpp -nt "$2" ] || [ ../../include/swn.h -nt "$2" ]; }
  for f in elementwise gate_mfma route chain chain_big wgrad sampling mip bounds hashgrid rayops; do
    if newer $f.hip "$OBJ/$f.o"; then $HIPCC $FLAGS -c $f.hip -o "$OBJ/$f.o" & pids+=($!); fi
  done
  # chain.hip a second time: the 512-feature geometry; a third time: the concat-skip layer mode of the dense NeRF trunk (kept out of
  # the default build's register budget)
  if newer chain.hip "$OBJ/chain_wide.o"; then $HIPCC $FLAGS -DSWN_WIDE=1 -c chain.hip -o "$OBJ/chain_wide.o" & pids+=($!); fi
  if newer chain.hip "$OBJ/chain_wide2.o"; then $HIPCC $FLAGS -DSWN_WIDE=2 -c chain.hip -o "$OBJ/chain_wide2.o" & pids+=($!); fi
  if newer chain.hip "$OBJ/chain_cat.o"; then $HIPCC $FLAGS -DSWN_CONCAT=1 -c chain.hip -o "$OBJ/chain_cat.o" & pids+=($!); fi
  for p in "${pids[@]}"; do wait $p; done
  $HIPCC --offload-arch=gfx950 -shared -fPIC "$OBJ"/{elementwise,gate_mfma,route,chain,chain_big,chain_wide,chain_wide2,chain_cat,wgrad,sampling,mip,bounds,hashgrid,rayops}.o -o "$OUT"
  echo "built $OUT"
}
# SWN_VARIANT=name (experiments): the bf16 library built with SWN_DEFS into libswn_hip_<name>.so / build_<name>/ - select it at run time
# with SWN_LIB=switch_nerf_amd/libswn_hip_<name>.so; the default libraries are left alone
if [ -n "${SWN_VARIANT:-}" ]; then build_one "build_${SWN_VARIANT}" "" "libswn_hip_${SWN_VARIANT}.so"; exit 0; fi
build_one build "" libswn_hip.so
if [ "${SWN_ONLY:-}" != "bf16" ]; then build_one build_f16 "-DSWN_HALF_F16" libswn_hip_f16.so; fi
