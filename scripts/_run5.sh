set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout 1500 python -m pytest tests/test_parallel_gpu.py tests/test_hash_gpu.py tests/test_model_gpu.py tests/test_kernels_gpu.py -x -q -k "parallel or two_ranks or bench_launches or expert_parallel or configs4 or wgrad" 2>&1 | tail -30 > gpurun_out/r03/t5_pytest.log
cat gpurun_out/r03/t5_pytest.log
