set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py -k "wgrad or expert_mlp" -x -q 2>&1 | tail -15 > gpurun_out/r03/t1_pytest_wgrad.log
timeout 200 python scripts/wgrad_check.py > gpurun_out/r03/t1_wgrad_new.log 2>&1
SWN_WGRAD_LEGACY=1 timeout 200 python scripts/wgrad_check.py > gpurun_out/r03/t1_wgrad_legacy.log 2>&1
timeout 1500 python scripts/graph_probe.py all > gpurun_out/r03/t1_probe.log 2>&1
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r03/t1_bench.log 2>&1
tail -5 gpurun_out/r03/t1_pytest_wgrad.log; cat gpurun_out/r03/t1_wgrad_new.log gpurun_out/r03/t1_wgrad_legacy.log; tail -30 gpurun_out/r03/t1_probe.log; tail -3 gpurun_out/r03/t1_bench.log
