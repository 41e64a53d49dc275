#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
for pv in 1; do
  echo "== SWN_HASH_PRIV=$pv"
  env ${pv:+SWN_HASH_PRIV=1} timeout 600 python -m pytest tests/test_hash_gpu.py -q -x -m gpu 2>&1 | tail -2
  env ${pv:+SWN_HASH_PRIV=1} timeout 300 python bench.py --hash --capacity-factor 1.25 --steps 10 --warmup 3 --no-cpu-baseline --no-balanced --graph off --no-events 2>/dev/null | tail -1 | cut -c1-170
done
