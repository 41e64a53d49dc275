#!/bin/bash
# round 4, GPU call 1: the GPU suite on the tier-1 changes, a fresh bench line, L2-side counters of the dense chains
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -40 > $O/c1_pytest.log
timeout 300 python -m pytest tests/test_fullsize_gpu.py -m gpu -q -s -k "bf16_vs_autocast" 2>&1 | grep -E "e_hip|ratio|worst|passed|failed|Error" | head -60 > $O/c1_threeway.log
timeout 400 python bench.py --steps 20 --warmup 5 > $O/c1_bench.json 2> $O/c1_bench.err
timeout 300 python bench.py --rays 1024 --steps 100 --warmup 20 --no-cpu-baseline --no-balanced --no-events > $O/c1_bench_1024.json 2>/dev/null
rocprofv3 -L > $O/c1_avail.txt 2>&1
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --routing balanced --no-events --graph off"
SWN_NO_OVERLAP=1 timeout 300 rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/p_l2 -- $B > $O/c1_p_l2.log 2>&1
python scripts/pmc_summary.py gpurun_out/p_l2 chain_kernel > $O/c1_pmc_l2_chains.txt 2>&1
python scripts/pmc_summary.py gpurun_out/p_l2 chainp >> $O/c1_pmc_l2_chains.txt 2>&1
rm -rf gpurun_out/p_l2
SWN_NO_OVERLAP=1 timeout 300 rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum --output-format csv -d gpurun_out/p_l1 -- $B > $O/c1_p_l1.log 2>&1
python scripts/pmc_summary.py gpurun_out/p_l1 chain_kernel > $O/c1_pmc_l1_chains.txt 2>&1
rm -rf gpurun_out/p_l1
tail -5 $O/c1_pytest.log; cat $O/c1_threeway.log | tail -30; cut -c1-400 $O/c1_bench.json; cut -c1-300 $O/c1_bench_1024.json; cat $O/c1_pmc_l2_chains.txt | head -40
