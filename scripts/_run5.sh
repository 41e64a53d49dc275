set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout 1500 python -m pytest tests/test_parallel_gpu.py tests/test_hash_gpu.py tests/test_model_gpu.py tests/test_kernels_gpu.py tests/test_moe_gpu.py -q -k "parallel or two_ranks or bench_launches or expert_parallel or configs4 or wgrad or route or moe or train_step" 2>&1 | tail -40 > gpurun_out/r03/t5_pytest.log
timeout 300 python bench.py --rays 1024 --steps 100 --warmup 20 --no-cpu-baseline --no-balanced > gpurun_out/r03/t5_bench_1024.log 2>&1
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r03/t5_bench.log 2>&1
cat gpurun_out/r03/t5_pytest.log; tail -1 gpurun_out/r03/t5_bench_1024.log | cut -c1-250;  tail -1 gpurun_out/r03/t5_bench.log | cut -c1-250
