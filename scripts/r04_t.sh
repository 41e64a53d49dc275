#!/bin/bash
# bash scripts/r04_t.sh "<pytest args>" : a pytest selection on the box
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout 1200 python -m pytest $1 -q -x -m gpu -s 2>&1 | tail -${2:-30} | tee gpurun_out/r04/t.log
