#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_dense_gpu.py -m gpu -q -x 2>&1 | grep -E "Error|assert|passed|failed|rror" | head -20
bash scripts/r04_prof.sh skipk | head -14
