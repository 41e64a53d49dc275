#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
R="--mip --samples 257 --rays 3328 --chunk 212992 --model-dim 512 --experts 16 --no-cpu-baseline --no-balanced --steps 6 --warmup 2"
for rep in 1 2; do
for v in default nofuse wide128; do
  case $v in
    default) E="" ;;
    nofuse) E="SWN_NO_FUSED_HEADS=1" ;;
    wide128) E="SWN_NO_FUSED_HEADS=1 SWN_LIB=switch_nerf_amd/libswn_hip_wide128.so" ;;
  esac
  env $E timeout 280 python bench.py $R 2>/dev/null | tail -1 | python -c "
import sys,json; j=json.loads(sys.stdin.read()); print('$v rep $rep: step', j['ms_per_step'], 'loss', j['config']['loss'])"
done; done
