#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
for z in "" 1; do
  echo "== ZERO=$z"
  ZERO=$z PERM=none timeout 300 python scripts/wgrad_check.py 2>&1 | grep -v amdgpu.ids | grep "balanced\|router"
done | tee gpurun_out/r04/wgz.log
