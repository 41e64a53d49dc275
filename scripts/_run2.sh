set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
timeout 300 python scripts/memset_graph_repro.py all > gpurun_out/r03/t2_memset_repro.log 2>&1
timeout 200 python scripts/wgrad_check.py > gpurun_out/r03/t2_wgrad_new.log 2>&1
timeout 1500 python scripts/graph_probe.py all > gpurun_out/r03/t2_probe.log 2>&1
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_fp16_gpu.py tests/test_autograd_gpu.py tests/test_parallel_gpu.py tests/test_background_gpu.py -x -q 2>&1 | tail -15 > gpurun_out/r03/t2_pytest.log
cat gpurun_out/r03/t2_memset_repro.log gpurun_out/r03/t2_wgrad_new.log; tail -30 gpurun_out/r03/t2_probe.log | cut -c1-600; cat gpurun_out/r03/t2_pytest.log
