set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout 900 python -m pytest tests/test_determinism_gpu.py tests/test_kernels_gpu.py -q -x -k "determin or twin or graph_replay or emb_grad or heads or gate" 2>&1 | tail -30 > gpurun_out/r03/t14_pytest.log
timeout 300 python scripts/determinism_check.py bf16 2048 > gpurun_out/r03/t14_det.log 2>&1
timeout 600 python scripts/graph_flake_probe.py eager 40 > gpurun_out/r03/t14_flake_eager.log 2>&1
cat gpurun_out/r03/t14_pytest.log; cat gpurun_out/r03/t14_det.log | head -40; tail -5 gpurun_out/r03/t14_flake_eager.log
