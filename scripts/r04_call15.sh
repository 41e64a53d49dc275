#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "fused_combine or front_chains_on_the_persistent" 2>&1 | tail -15
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-balanced --no-events"
for g in 1 7 1 7; do
  SWN_TAIL_GEOM=$g $B 2>/dev/null | tail -1 | python -c "
import sys,json; j=json.loads(sys.stdin.read()); print('tail geom $g: step', j['ms_per_step'], 'loss', j['config']['loss'])"
done
