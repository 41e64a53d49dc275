#!/bin/bash
# Collects the round's rocprofv3 evidence on a GPU box into gpurun_out/ (summaries only; raw databases are deleted).
#   bash scripts/collect_profiles.sh
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-balanced --graph off"
# (a) per-kernel time of the step (events on: expert weight gradients on the main stream, like the default timed region)
rocprofv3 --kernel-trace --stats -d gpurun_out/p_step -o step -- $B > gpurun_out/p_step.log 2>&1
python scripts/prof_summary.py $(find gpurun_out/p_step -name "*.db" | head -1) 40 > gpurun_out/r02_kernel_stats_step.md
tail -1 gpurun_out/p_step.log | grep '^{' >> gpurun_out/r02_kernel_stats_step.md
rm -rf gpurun_out/p_step
# (b) HBM traffic: two counter-only passes
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/p_$c -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --routing balanced --no-events --graph off > gpurun_out/p_$c.log 2>&1
  python scripts/pmc_summary.py gpurun_out/p_$c > gpurun_out/r02_pmc_$c.txt
  tail -1 gpurun_out/p_$c.log | grep '^{' >> gpurun_out/r02_pmc_$c.txt
  rm -rf gpurun_out/p_$c
done
# (c) SQ counters of the expert chains (one pass, 8 SQ slots)
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --output-format csv -d gpurun_out/p_sq -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --routing balanced --no-events --graph off > gpurun_out/p_sq.log 2>&1
python scripts/pmc_summary.py gpurun_out/p_sq chainp > gpurun_out/r02_pmc_sq_chainb.txt
python scripts/pmc_summary.py gpurun_out/p_sq wgrad_kernel >> gpurun_out/r02_pmc_sq_chainb.txt
python scripts/pmc_summary.py gpurun_out/p_sq gate_ >> gpurun_out/r02_pmc_sq_chainb.txt
rm -rf gpurun_out/p_sq
# (d) the bench lines
python bench.py > gpurun_out/r02_bench_default.json 2> gpurun_out/r02_bench_default.err
python bench.py --steps 20 --warmup 5 > gpurun_out/r02_bench_20_5.json 2>/dev/null
python bench.py --rays 1024 --steps 100 --warmup 20 --no-cpu-baseline --no-balanced --graph on > gpurun_out/r02_bench_1024rays_graph.json 2>/dev/null
python bench.py --rays 1024 --steps 100 --warmup 20 --no-cpu-baseline --no-balanced --graph off --no-events > gpurun_out/r02_bench_1024rays_eager.json 2>/dev/null
ls -la gpurun_out | head -30
