#!/bin/bash
# round 5, GPU call G: swn_route_top1x modes - 0 = the 20 per-phase kernels, 1 = route_one_kernel per phase (9 launches), 2 = one launch
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
timeout 400 python -m pytest tests/test_kernels_gpu.py -q -x -k "route" 2>&1 | tail -15 > $O/g_route.log
tail -3 $O/g_route.log
if ! grep -q " passed" $O/g_route.log || grep -q "failed\|error" $O/g_route.log; then echo ROUTE_TESTS_NOT_GREEN; cat $O/g_route.log; exit 1; fi
for i in 1 2 3; do
  for m in 0 1; do
    SWN_ROUTE_MODE=$m timeout 300 python bench.py --rays 1024 --steps 100 --warmup 20 --no-cpu-baseline --no-balanced --no-events > $O/g_1024_m${m}_$i.json 2>/dev/null
    SWN_ROUTE_MODE=$m timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-balanced --no-events > $O/g_step_m${m}_$i.json 2>/dev/null
  done
done
for m in 0 1; do
SWN_ROUTE_MODE=$m SWN_NO_OVERLAP=1 rocprofv3 --kernel-trace --stats -d gpurun_out/p_g$m -o step -- python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-balanced --graph off --no-events > $O/g_p$m.log 2>&1
python scripts/prof_summary.py $(find gpurun_out/p_g$m -name "*.db" | head -1) 60 > $O/g_kernel_stats_m$m.md
rm -rf gpurun_out/p_g$m
done
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/g_*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], "ms/step", j["ms_per_step"], "value", j["value"])
    except Exception as e: print(f, "ERR", e)
PY
grep -i "route\|laux\|fill_u32" $O/g_kernel_stats_m0.md $O/g_kernel_stats_m1.md | cut -c1-160
