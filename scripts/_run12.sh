cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
SWN_NO_OVERLAP=1 rocprofv3 --kernel-trace --stats -d gpurun_out/p12 -o s -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-balanced --graph off --no-events > gpurun_out/r03/t12_p.log 2>&1
python scripts/prof_summary.py $(find gpurun_out/p12 -name "*.db" | head -1) 40 > gpurun_out/r03/t12_kernel_stats.md
rm -rf gpurun_out/p12
tail -1 gpurun_out/r03/t12_p.log | cut -c1-200
cat gpurun_out/r03/t12_kernel_stats.md | cut -c1-150
