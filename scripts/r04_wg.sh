#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
PERM=none timeout 300 python scripts/wgrad_check.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04/wg_none.log
timeout 300 python scripts/wgrad_check.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04/wg.log
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -m gpu -k wgrad 2>&1 | tail -3
