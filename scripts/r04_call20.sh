#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_parallel_gpu.py -m gpu -q -x -k "expert_parallel or two_ranks or split_backward" 2>&1 | grep -E "Error|assert|passed|failed|rror" | head -20
