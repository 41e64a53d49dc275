#!/bin/bash
# per-kernel table of the step (eager launches, no side-stream overlap): bash scripts/r04_prof.sh <tag> [env assignments...]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=$1; shift
O=gpurun_out/r04; mkdir -p $O
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-balanced --graph off --no-events"
env SWN_NO_OVERLAP=1 "$@" rocprofv3 --kernel-trace --stats -d gpurun_out/p_$TAG -o step -- $B > $O/prof_$TAG.log 2>&1
python scripts/prof_summary.py $(find gpurun_out/p_$TAG -name "*.db" | head -1) 28 > $O/prof_$TAG.md
tail -1 $O/prof_$TAG.log | grep '^{' | cut -c1-300 >> $O/prof_$TAG.md
rm -rf gpurun_out/p_$TAG
head -32 $O/prof_$TAG.md
