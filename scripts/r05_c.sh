#!/bin/bash
# round 5, GPU call C: the one-launch routing with coherent (sc1) data instead of L2 write-back / invalidate fences at the barriers
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -x -k "route" 2>&1 | tail -15 > $O/c_route.log
tail -3 $O/c_route.log
if ! grep -q " passed" $O/c_route.log || grep -q "failed\|error" $O/c_route.log; then echo ROUTE_TESTS_NOT_GREEN; exit 1; fi
for i in 1 2; do
  timeout 300 python bench.py --rays 1024 --steps 100 --warmup 20 --no-cpu-baseline --no-balanced --graph on --no-events > $O/c_1024_graph_$i.json 2>/dev/null
  SWN_ROUTE_MULTI=1 timeout 300 python bench.py --rays 1024 --steps 100 --warmup 20 --no-cpu-baseline --no-balanced --graph on --no-events > $O/c_1024_graph_multi_$i.json 2>/dev/null
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-balanced --no-events > $O/c_step_$i.json 2>/dev/null
  SWN_ROUTE_MULTI=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-balanced --no-events > $O/c_step_multi_$i.json 2>/dev/null
done
SWN_NO_OVERLAP=1 rocprofv3 --kernel-trace --stats -d gpurun_out/p_1024 -o step -- python bench.py --rays 1024 --steps 30 --warmup 5 --no-cpu-baseline --no-balanced --graph off --no-events > $O/c_p1024.log 2>&1
python scripts/prof_summary.py $(find gpurun_out/p_1024 -name "*.db" | head -1) 50 > $O/c_kernel_stats_1024rays.md
rm -rf gpurun_out/p_1024
SWN_NO_OVERLAP=1 rocprofv3 --kernel-trace --stats -d gpurun_out/p_full -o step -- python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-balanced --graph off --no-events > $O/c_pfull.log 2>&1
python scripts/prof_summary.py $(find gpurun_out/p_full -name "*.db" | head -1) 50 > $O/c_kernel_stats_step.md
rm -rf gpurun_out/p_full
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/c_*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], "ms/step", j["ms_per_step"], "value", j["value"])
    except Exception as e:
        print(f, "ERR", e)
PY
grep -i "route" $O/c_kernel_stats_1024rays.md $O/c_kernel_stats_step.md | cut -c1-170
