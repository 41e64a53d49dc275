#!/bin/bash
# A/B of environment switches on ONE box (box-to-box spread is +-1.5 %): bash scripts/ab_env.sh "VAR1=1" "VAR2=1" ...  (each run twice, interleaved)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-balanced --no-events"
B2="python bench.py --rays 1024 --steps 100 --warmup 20 --no-cpu-baseline --no-balanced --no-events"
for rep in 1 2; do
  for cfg in "" "$@"; do
    a=$(env $cfg $B 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
    b=$(env $cfg $B2 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
    echo "rep $rep [${cfg:-default}] 8192 rays: $a ms   1024 rays: $b ms" | tee -a gpurun_out/ab/ab.log
  done
done
