#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
B="python bench.py --rays 1024 --steps 100 --warmup 20 --no-cpu-baseline --no-balanced --graph off --no-events"
env SWN_NO_OVERLAP=1 rocprofv3 --kernel-trace --stats -d gpurun_out/p_1024 -o step -- $B > $O/prof_1024.log 2>&1
python scripts/prof_summary.py $(find gpurun_out/p_1024 -name "*.db" | head -1) 60 > $O/prof_1024.md
rm -rf gpurun_out/p_1024
cat $O/prof_1024.md
python bench.py --rays 1024 --steps 100 --warmup 20 --no-cpu-baseline --no-balanced --no-events 2>/dev/null | tail -1 | cut -c1-160
