#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout 300 python scripts/chain9_probe.py 2>&1 | tee gpurun_out/r04/c9.log
