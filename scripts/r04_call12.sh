#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
for st in 0 2 4 8; do
echo "== stagger $st"
SWN_CHAINQ_STAGGER=$st SWN_LIB=switch_nerf_amd/libswn_hip_timing.so timeout 300 python scripts/chainq_front_phases.py 2>&1 | grep "train wave"
done > $O/c12.log 2>&1
cat $O/c12.log
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-balanced --no-events"
for st in 0 4 0 4; do
  SWN_CHAINQ_STAGGER=$st $B 2>/dev/null | tail -1 | python -c "
import sys,json; j=json.loads(sys.stdin.read()); print('stagger $st: step', j['ms_per_step'])"
done
