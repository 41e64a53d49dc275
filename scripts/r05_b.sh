#!/bin/bash
# round 5, GPU call B: the one-launch routing (swn_route_top1x) - its twin test first (under a short timeout: a grid barrier that never
# opens must not take the box), then the suite, the bench lines and the 1024-rays-per-GPU kernel table
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -x -k "route" 2>&1 | tail -15 > $O/b_route.log
cat $O/b_route.log | tail -5
if ! grep -q " passed" $O/b_route.log || grep -q "failed\|error" $O/b_route.log; then echo ROUTE_TESTS_NOT_GREEN; exit 1; fi
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -30 > $O/b_pytest.log
tail -6 $O/b_pytest.log
timeout 400 python bench.py --steps 20 --warmup 5 > $O/b_bench.json 2> $O/b_bench.err
for i in 1 2; do
  timeout 300 python bench.py --rays 1024 --steps 100 --warmup 20 --no-cpu-baseline --no-balanced --graph on > $O/b_1024_graph_$i.json 2>/dev/null
  SWN_ROUTE_MULTI=1 timeout 300 python bench.py --rays 1024 --steps 100 --warmup 20 --no-cpu-baseline --no-balanced --graph on > $O/b_1024_graph_multi_$i.json 2>/dev/null
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-balanced --no-events > $O/b_step_$i.json 2>/dev/null
  SWN_ROUTE_MULTI=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-balanced --no-events > $O/b_step_multi_$i.json 2>/dev/null
done
SWN_NO_OVERLAP=1 rocprofv3 --kernel-trace --stats -d gpurun_out/p_1024 -o step -- python bench.py --rays 1024 --steps 30 --warmup 5 --no-cpu-baseline --no-balanced --graph off --no-events > $O/b_p1024.log 2>&1
python scripts/prof_summary.py $(find gpurun_out/p_1024 -name "*.db" | head -1) 50 > $O/b_kernel_stats_1024rays.md
rm -rf gpurun_out/p_1024
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/b_*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], "ms/step", j["ms_per_step"], "value", j["value"])
    except Exception as e:
        print(f, "ERR", e)
PY
grep -i "route\|fill_u32\|laux" $O/b_kernel_stats_1024rays.md | cut -c1-150
