#!/bin/bash
# Build libswn_hip.so (gfx950 only) in-tree.  hipcc cross-compiles without a GPU.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
cd "$HERE/csrc"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="${SWN_DEFS:-} --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-variable -Wno-unused-but-set-variable -ffp-contract=fast-honor-pragmas"
mkdir -p "$HERE/build"
pids=()
for f in elementwise route chain chain_big wgrad sampling mip bounds hashgrid; do
  if [ ! -f "$HERE/build/$f.o" ] || [ "$f.hip" -nt "$HERE/build/$f.o" ] || [ common.hpp -nt "$HERE/build/$f.o" ] || [ pe_store.hpp -nt "$HERE/build/$f.o" ] || [ ../../include/swn.h -nt "$HERE/build/$f.o" ]; then
    $HIPCC $FLAGS -c $f.hip -o "$HERE/build/$f.o" &
    pids+=($!)
  fi
done
# chain.hip a second time: the 512-feature geometry
if [ ! -f "$HERE/build/chain_wide.o" ] || [ chain.hip -nt "$HERE/build/chain_wide.o" ] || [ common.hpp -nt "$HERE/build/chain_wide.o" ] || [ ../../include/swn.h -nt "$HERE/build/chain_wide.o" ]; then
  $HIPCC $FLAGS -DSWN_WIDE=1 -c chain.hip -o "$HERE/build/chain_wide.o" &
  pids+=($!)
fi
# ... and a third time: the concat-skip layer mode of the dense NeRF trunk (kept out of the default build's register budget)
if [ ! -f "$HERE/build/chain_cat.o" ] || [ chain.hip -nt "$HERE/build/chain_cat.o" ] || [ common.hpp -nt "$HERE/build/chain_cat.o" ] || [ ../../include/swn.h -nt "$HERE/build/chain_cat.o" ]; then
  $HIPCC $FLAGS -DSWN_CONCAT=1 -c chain.hip -o "$HERE/build/chain_cat.o" &
  pids+=($!)
fi
for p in "${pids[@]}"; do wait $p; done
$HIPCC --offload-arch=gfx950 -shared -fPIC "$HERE"/build/{elementwise,route,chain,chain_big,chain_wide,chain_cat,wgrad,sampling,mip,bounds,hashgrid}.o -o "$HERE/libswn_hip.so"
echo "built $HERE/libswn_hip.so"
