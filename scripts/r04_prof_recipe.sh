#!/bin/bash
# kernel table of one recipe: bash scripts/r04_prof_recipe.sh <tag> <bench flags...>
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=$1; shift
O=gpurun_out/r04; mkdir -p $O
env SWN_NO_OVERLAP=1 rocprofv3 --kernel-trace --stats -d gpurun_out/p_$TAG -o step -- python bench.py "$@" --steps 6 --warmup 2 --no-cpu-baseline --no-balanced --graph off --no-events > $O/prof_$TAG.log 2>&1
python scripts/prof_summary.py $(find gpurun_out/p_$TAG -name "*.db" | head -1) 24 > $O/prof_$TAG.md
rm -rf gpurun_out/p_$TAG
head -28 $O/prof_$TAG.md | cut -c1-170
