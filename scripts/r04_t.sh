#!/bin/bash
# bash scripts/r04_t.sh <tail lines> <pytest args ...> : a pytest selection on the box
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
N=$1; shift
timeout 1200 python -m pytest "$@" -q -x -m gpu -s 2>&1 | tail -$N | tee gpurun_out/r04/t.log
