#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout 1200 python -m pytest tests -q -x -m gpu -k "wide or mission or sixteen or 512" 2>&1 | tail -5
for w in 1 ""; do
  echo "== SWN_NO_WIDE2=$w"
  env ${w:+SWN_NO_WIDE2=1} timeout 300 python bench.py --mip --samples 257 --rays 3328 --chunk 212992 --model-dim 512 --experts 16 --steps 10 --warmup 3 --no-cpu-baseline --no-balanced 2>/dev/null | tail -1 | cut -c1-170
done
