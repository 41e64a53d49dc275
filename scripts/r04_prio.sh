#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
for rep in 1 2; do
for lib in switch_nerf_amd/libswn_hip_noprio.so "" switch_nerf_amd/libswn_hip_prio3.so; do
  echo "== rep $rep lib ${lib:-default(prio1)}"
  SWN_LIB=$lib timeout 300 python scripts/chainq_timing.py 7 2>&1 | grep "segments 16"
  SWN_LIB=$lib timeout 300 python scripts/tailfuse_check.py full time 2>&1 | grep "unfused train"
  SWN_LIB=$lib timeout 300 python scripts/headfuse_check.py full time 2>&1 | grep "fused"
done; done | tee gpurun_out/r04/prio.log
