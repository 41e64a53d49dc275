set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
O=gpurun_out/r03
SWN_NO_OVERLAP=1 rocprofv3 --kernel-trace --stats -d gpurun_out/p_1024 -o step -- python bench.py --rays 1024 --steps 30 --warmup 5 --no-cpu-baseline --no-balanced --graph off --no-events > $O/p_1024.log 2>&1
python scripts/prof_summary.py $(find gpurun_out/p_1024 -name "*.db" | head -1) 60 > $O/p1024_noov.md
rm -rf gpurun_out/p_1024
cat $O/p1024_noov.md
