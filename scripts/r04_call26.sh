#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-balanced"
for rep in 1 2 3; do
for v in default nt; do
  E=""; [ $v = nt ] && E="SWN_LIB=switch_nerf_amd/libswn_hip_wgnt.so"
  env $E $B 2>/dev/null | tail -1 | python -c "
import sys,json; j=json.loads(sys.stdin.read()); k=j['kernels']; print('$v rep $rep: step', j['ms_per_step'], 'wgrad', k['expert_wgrad']['ms'])"
done; done
