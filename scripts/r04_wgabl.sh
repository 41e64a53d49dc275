#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
for lib in "" switch_nerf_amd/libswn_hip_wgabl2.so switch_nerf_amd/libswn_hip_wgabl3.so switch_nerf_amd/libswn_hip_wgstream.so; do
  echo "== lib ${lib:-default}"
  PERM=none SWN_LIB=$lib timeout 300 python scripts/wgrad_check.py 2>&1 | grep -v amdgpu.ids | grep "balanced\|router\|batched"
done | tee gpurun_out/r04/wgabl.log
