set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > gpurun_out/r03/t3_pytest_all.log
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r03/t3_bench.log 2>&1
timeout 300 python bench.py --eval --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r03/t3_bench_eval.log 2>&1
timeout 300 python bench.py --rays 1024 --steps 50 --warmup 10 --no-cpu-baseline --no-balanced > gpurun_out/r03/t3_bench_1024.log 2>&1
cat gpurun_out/r03/t3_pytest_all.log; tail -2 gpurun_out/r03/t3_bench.log; tail -1 gpurun_out/r03/t3_bench_eval.log; tail -1 gpurun_out/r03/t3_bench_1024.log
