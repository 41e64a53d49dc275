#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout 300 python scripts/headfuse_check.py ${1:-small} $2 2>&1 | tee gpurun_out/r04/hf_${1:-small}.log
