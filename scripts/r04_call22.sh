#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "front_chains or geometry_bit_exact or fused_combine" 2>&1 | grep -E "Error|assert|passed|failed|rror" | head -20
bash scripts/r04_prof.sh yidx | head -12
