#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for rep in 1 2; do for lib in "" switch_nerf_amd/libswn_hip_cat2.so; do
  echo -n "rep $rep ${lib:-default(3 per CU)}: dense "
  SWN_LIB=$lib timeout 300 python bench.py --dense --steps 10 --warmup 3 --no-cpu-baseline --no-balanced 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"
  echo -n "      bg "
  SWN_LIB=$lib timeout 300 python bench.py --bg --steps 10 --warmup 3 --no-cpu-baseline --no-balanced 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"
done; done
