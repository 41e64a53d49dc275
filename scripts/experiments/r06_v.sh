#!/bin/bash
# round 6, GPU call V: the whole GPU suite on the build with the wide router and the owner-tail exchange
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
timeout 3000 python -m pytest tests/ -m gpu -q -x 2>&1 | tail -15
