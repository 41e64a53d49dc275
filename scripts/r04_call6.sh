#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-balanced"
B2="python bench.py --rays 1024 --steps 100 --warmup 20 --no-cpu-baseline --no-balanced --no-events"
B3="python bench.py --eval --steps 50 --warmup 10 --no-cpu-baseline"
for rep in 1 2; do
for g in 4 7; do
  SWN_CHAIN_GEOM=$g $B 2>/dev/null | tail -1 | python -c "
import sys,json; j=json.loads(sys.stdin.read()); k=j['kernels']
print('geom $g rep $rep: step', j['ms_per_step'], 'eager', j['config']['eager_ms_per_step'], 'fwd', k['expert_fwd']['ms'], 'bwd', k['expert_bwd']['ms'], 'wgrad', k['expert_wgrad']['ms'], 'nosave', k['expert_fwd_nosave']['ms'], k['expert_fwd_nosave']['mfma_frac'], 'kept', j['config']['kept_token_fraction_mean'])"
  SWN_CHAIN_GEOM=$g $B2 2>/dev/null | tail -1 | python -c "
import sys,json; j=json.loads(sys.stdin.read()); print('geom $g rep $rep: 1024 rays', j['ms_per_step'])"
  SWN_CHAIN_GEOM=$g $B3 2>/dev/null | tail -1 | python -c "
import sys,json; j=json.loads(sys.stdin.read()); print('geom $g rep $rep: eval', j['ms_per_step'])"
done; done > $O/c6.log 2>&1
cat $O/c6.log
