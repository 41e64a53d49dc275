#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -m gpu -k "chain" 2>&1 | tail -3
for rep in 1 2; do
  timeout 300 python scripts/chainq_timing.py 7 2>&1 | grep "segments 16"
  timeout 300 python scripts/tailfuse_check.py full time 2>&1 | grep "unfused train"
  timeout 300 python scripts/headfuse_check.py full time 2>&1 | grep "fused"
done | tee gpurun_out/r04/bv.log
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-balanced --no-events 2>/dev/null | tail -1 | cut -c1-180
