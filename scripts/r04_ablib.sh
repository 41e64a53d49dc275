#!/bin/bash
# A/B of two builds of the library on ONE box: bash scripts/r04_ablib.sh <other .so>
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-balanced --no-events"
for rep in 1 2 3; do for lib in "$1" ""; do
  echo -n "rep $rep ${lib:-default}: "
  SWN_LIB=$lib $B 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"
done; done
