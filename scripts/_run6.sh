set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout 600 python -m pytest tests/test_hash_gpu.py -q -k "configs4" 2>&1 | tail -5 > gpurun_out/r03/t6_pytest.log
SWN_LIB=$PWD/switch_nerf_amd/libswn_hip_bm128.so timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-balanced --no-events > gpurun_out/r03/t6_bench_bm128.log 2>&1
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-balanced --no-events > gpurun_out/r03/t6_bench_default.log 2>&1
rocprofv3 --kernel-trace --stats -d gpurun_out/p_step -o s -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-balanced --graph off --no-events > gpurun_out/r03/t6_p_step.log 2>&1
python scripts/prof_summary.py $(find gpurun_out/p_step -name "*.db" | head -1) 45 > gpurun_out/r03/t6_kernel_stats_step.md
rm -rf gpurun_out/p_step
SWN_LIB=$PWD/switch_nerf_amd/libswn_hip_bm128.so rocprofv3 --kernel-trace --stats -d gpurun_out/p_step2 -o s -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-balanced --graph off --no-events > gpurun_out/r03/t6_p_step_bm128.log 2>&1
python scripts/prof_summary.py $(find gpurun_out/p_step2 -name "*.db" | head -1) 14 > gpurun_out/r03/t6_kernel_stats_step_bm128.md
rm -rf gpurun_out/p_step2
cat gpurun_out/r03/t6_pytest.log; tail -1 gpurun_out/r03/t6_bench_bm128.log | cut -c1-200; tail -1 gpurun_out/r03/t6_bench_default.log | cut -c1-200; head -30 gpurun_out/r03/t6_kernel_stats_step.md; cat gpurun_out/r03/t6_kernel_stats_step_bm128.md
