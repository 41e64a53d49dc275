set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout 1200 python -m pytest tests/test_autograd_gpu.py tests/test_fullsize_gpu.py tests/test_hash_gpu.py tests/test_model_gpu.py -x -q -k "autograd or graph_train or external_optimizer or autocast or configs4 or ragged or runner or full_segment" 2>&1 | tail -25 > gpurun_out/r03/t4_pytest.log
timeout 500 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r03/t4_bench.log 2>&1
timeout 300 python bench.py --eval --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r03/t4_bench_eval.log 2>&1
rocprofv3 --kernel-trace --stats -d gpurun_out/p1024 -o s -- python bench.py --rays 1024 --steps 30 --warmup 5 --no-cpu-baseline --no-balanced --graph off --no-events > gpurun_out/r03/t4_p1024.log 2>&1
python scripts/prof_summary.py $(find gpurun_out/p1024 -name "*.db" | head -1) 45 > gpurun_out/r03/t4_kernel_stats_1024.md
rm -rf gpurun_out/p1024
cat gpurun_out/r03/t4_pytest.log; tail -1 gpurun_out/r03/t4_bench.log; tail -1 gpurun_out/r03/t4_bench_eval.log | cut -c1-400; cat gpurun_out/r03/t4_kernel_stats_1024.md
