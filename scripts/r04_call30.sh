#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "persistent_chain_shapes" 2>&1 | grep -E "Error|assert|passed|failed|rror|differ" | head -20
