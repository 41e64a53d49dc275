#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "front_chains_on_the_persistent or (geometry_bit_exact and (6 or 7))" 2>&1 | tail -15 > $O/c7_pytest.log
cat $O/c7_pytest.log
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-balanced"
for g in 1 7 1 7; do
  SWN_FRONT_GEOM=$g $B 2>/dev/null | tail -1 | python -c "
import sys,json; j=json.loads(sys.stdin.read()); k=j['kernels']
print('front geom $g: step', j['ms_per_step'], 'eager', j['config']['eager_ms_per_step'], 'loss', j['config']['loss'], 'kept', j['config']['kept_token_fraction_mean'])"
done > $O/c7_bench.log 2>&1
cat $O/c7_bench.log
