set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -30 > gpurun_out/r03/t10_pytest_all.log
timeout 300 python bench.py --rays 1024 --steps 100 --warmup 20 --no-cpu-baseline --no-balanced > gpurun_out/r03/t10_bench_1024.log 2>&1
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r03/t10_bench.log 2>&1
cat gpurun_out/r03/t10_pytest_all.log; tail -1 gpurun_out/r03/t10_bench_1024.log | cut -c1-250; tail -1 gpurun_out/r03/t10_bench.log | cut -c1-250
