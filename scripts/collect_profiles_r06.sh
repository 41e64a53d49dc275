#!/bin/bash
# Round 6: the round's rocprofv3 evidence on a GPU box -> gpurun_out/r06/ (summaries only; raw databases are deleted), every table from
# ONE build (bench.py stamps config.csrc_sha256; scripts/make_traffic.py ties profiles/traffic.json to it).
#   bash scripts/collect_profiles_r06.sh
set -x
R=r06
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/$R; mkdir -p $O
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-balanced --graph off --no-events"
# (a) per-kernel time of the step, eager launches, without the side-stream overlap (kernel durations undisturbed by concurrent kernels)
SWN_NO_OVERLAP=1 rocprofv3 --kernel-trace --stats -d gpurun_out/p_step2 -o step -- $B > $O/p_step_noov.log 2>&1
python scripts/prof_summary.py $(find gpurun_out/p_step2 -name "*.db" | head -1) 45 > $O/${R}_kernel_stats_step_no_overlap.md
grep '^{' $O/p_step_noov.log | tail -1 >> $O/${R}_kernel_stats_step_no_overlap.md
rm -rf gpurun_out/p_step2
# (b) HBM traffic of the TIMED workload (the router's own routing): two counter-only passes, each followed by its bench line
# ONE step per pass (no warm-up): every dispatch of a kernel in the pass has the same kept rows - the router's kept fraction moves from
# step to step (0.52 at the first step, ~0.8 after twenty), and the table stores bytes per launch AT the pass's kept rows
P1="python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-balanced --no-events --graph off"
P="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-balanced --no-events --graph off"
for c in FETCH_SIZE WRITE_SIZE; do
  SWN_NO_OVERLAP=1 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/p_$c -- $P1 > $O/p_$c.log 2>&1
  python scripts/pmc_summary.py gpurun_out/p_$c > $O/${R}_pmc_$c.txt
  grep '^{' $O/p_$c.log | tail -1 >> $O/${R}_pmc_$c.txt
  rm -rf gpurun_out/p_$c
done
mkdir -p profiles; cp $O/${R}_pmc_FETCH_SIZE.txt $O/${R}_pmc_WRITE_SIZE.txt profiles/
python scripts/make_traffic.py $R > $O/make_traffic.log 2>&1; cp profiles/traffic.json $O/traffic.json
# (c) SQ counters of the expert kernels (one pass, 8 SQ slots) and of the save-free expert chain (inference forward)
SWN_NO_OVERLAP=1 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --output-format csv -d gpurun_out/p_sq -- $P > $O/p_sq.log 2>&1
python scripts/pmc_summary.py gpurun_out/p_sq chainq > $O/${R}_pmc_sq_experts.txt
python scripts/pmc_summary.py gpurun_out/p_sq wgrad_stream >> $O/${R}_pmc_sq_experts.txt
rm -rf gpurun_out/p_sq
# (c2) L2 -> L1 requests of the chains (the weight stream through the CU's vector memory path)
SWN_NO_OVERLAP=1 rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/p_l2 -- $P > $O/p_l2.log 2>&1
python scripts/pmc_summary.py gpurun_out/p_l2 chain > $O/${R}_pmc_l2_chains.txt
rm -rf gpurun_out/p_l2
# (c3) kernel table of the 1024-rays-per-GPU share (8-GPU strong scaling)
SWN_NO_OVERLAP=1 rocprofv3 --kernel-trace --stats -d gpurun_out/p_1024 -o step -- python bench.py --rays 1024 --steps 30 --warmup 5 --no-cpu-baseline --no-balanced --graph off --no-events > $O/p_1024.log 2>&1
python scripts/prof_summary.py $(find gpurun_out/p_1024 -name "*.db" | head -1) 45 > $O/${R}_kernel_stats_1024rays.md
rm -rf gpurun_out/p_1024
# (d) the bench lines (traffic.json of THIS build is in place: roofline.traffic is reported)
python bench.py > $O/${R}_bench_default.json 2> $O/${R}_bench_default.err
python bench.py --steps 20 --warmup 5 > $O/${R}_bench_20_5.json 2>/dev/null
python bench.py --rays 1024 --steps 100 --warmup 20 --no-cpu-baseline --no-balanced --graph on > $O/${R}_bench_1024rays_graph.json 2>/dev/null
python bench.py --rays 1024 --steps 100 --warmup 20 --no-cpu-baseline --no-balanced --graph off --no-events > $O/${R}_bench_1024rays_eager.json 2>/dev/null
python bench.py --eval --steps 50 --warmup 10 --no-cpu-baseline > $O/${R}_bench_eval_graph.json 2>/dev/null
ls -la $O | head -60
# (e) round 6: the other recipes whose kernels changed - hash-grid input (binned backward) and Mission Bay's per-GPU share (512-feature chains)
H="python bench.py --hash --capacity-factor 1.25 --dtype fp16 --steps 6 --warmup 3 --no-cpu-baseline --no-balanced --graph off --no-events"
SWN_NO_OVERLAP=1 rocprofv3 --kernel-trace --stats -d gpurun_out/p_hash -o hash -- $H > $O/p_hash.log 2>&1
python scripts/prof_summary.py $(find gpurun_out/p_hash -name "*.db" | head -1) 30 > $O/${R}_kernel_stats_hash.md
rm -rf gpurun_out/p_hash
MB="python bench.py --mip --samples 257 --rays 3328 --chunk 212992 --model-dim 512 --experts 16 --steps 6 --warmup 3 --no-cpu-baseline --no-balanced --graph off --no-events"
SWN_NO_OVERLAP=1 rocprofv3 --kernel-trace --stats -d gpurun_out/p_mb -o mb -- $MB > $O/p_mb.log 2>&1
python scripts/prof_summary.py $(find gpurun_out/p_mb -name "*.db" | head -1) 30 > $O/${R}_kernel_stats_mission_bay.md
rm -rf gpurun_out/p_mb
# (f) the recipes' bench lines
( for fl in "" "--bg" "--bg --fine 512" "--dense" "--hash --capacity-factor 1.25" "--hash --capacity-factor 1.25 --dtype fp16" "--eval" "--fine 512" "--mip --samples 257" "--mip --samples 257 --rays 3328 --chunk 212992 --model-dim 512 --experts 16" "--dtype fp16"; do
    echo "### bench.py $fl"; python bench.py $fl --no-cpu-baseline --no-balanced --steps 10 --warmup 3 2>/dev/null | tail -1; done ) > $O/${R}_bench_recipes.txt
# (g) the one-GPU rehearsal of the multi-GPU step over RCCL (world-1 loopback)
python bench.py --gpus 1 --loopback --rays 1024 --steps 50 --warmup 10 --no-cpu-baseline --no-balanced --no-events > $O/${R}_bench_loopback_dp_1024rays.json 2>/dev/null
python bench.py --gpus 1 --loopback --parallelism ep --steps 10 --warmup 3 --no-cpu-baseline --no-balanced --no-events > $O/${R}_bench_loopback_ep.json 2>/dev/null
python bench.py --gpus 1 --loopback --parallelism ep --ep-owner-tail --steps 10 --warmup 3 --no-cpu-baseline --no-balanced --no-events > $O/${R}_bench_loopback_ep_owner_tail.json 2>/dev/null
# (h) the 512-feature router against fp64 + timing, both 16-bit builds
GATE_G=512 GATE_E=16 python scripts/gate_check.py 2>&1 | grep -v amdgpu > $O/${R}_gate_check_512x16.txt
GATE_G=512 GATE_E=16 GATE_DTYPE=fp16 python scripts/gate_check.py 2>&1 | grep -v amdgpu > $O/${R}_gate_check_512x16_fp16.txt
ls -la $O | head -80
