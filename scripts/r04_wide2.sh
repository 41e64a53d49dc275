#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for rep in 1 2; do for lib in "" switch_nerf_amd/libswn_hip_w2r3.so switch_nerf_amd/libswn_hip_w2r4.so; do
  echo -n "rep $rep ${lib:-default(ring2)}: "
  SWN_LIB=$lib timeout 300 python bench.py --mip --samples 257 --rays 3328 --chunk 212992 --model-dim 512 --experts 16 --steps 10 --warmup 3 --no-cpu-baseline --no-balanced 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"
done; done
