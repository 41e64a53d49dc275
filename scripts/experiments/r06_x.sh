#!/bin/bash
# round 6, GPU call X: HBM traffic of the Mission Bay recipe's launches (FETCH_SIZE / WRITE_SIZE passes): do the 256-column block GEMMs of the
# 512-wide weight gradients fetch their operands twice?
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
MB="python bench.py --mip --samples 257 --rays 3328 --chunk 212992 --model-dim 512 --experts 16 --steps 1 --warmup 0 --no-cpu-baseline --no-balanced --graph off --no-events"
for c in FETCH_SIZE WRITE_SIZE; do
  SWN_NO_OVERLAP=1 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/p_x$c -- $MB > $O/x_$c.log 2>&1
  python scripts/pmc_summary.py gpurun_out/p_x$c > $O/x_pmc_mb_$c.txt
  rm -rf gpurun_out/p_x$c
  cut -c1-200 $O/x_pmc_mb_$c.txt | head -30
done
grep '^{' $O/x_FETCH_SIZE.log | tail -1 | cut -c1-600
