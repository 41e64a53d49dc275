#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_determinism_gpu.py -q -x -m gpu -k "wgrad or determinism" 2>&1 | tail -2
for lib in switch_nerf_amd/libswn_hip_oldwg.so ""; do echo "== ${lib:-default}"; SWN_LIB=$lib PERM=none timeout 300 python scripts/wgrad_check.py 2>&1 | grep "balanced\|router\|batched"; done
bash scripts/r04_ablib.sh switch_nerf_amd/libswn_hip_oldwg.so
