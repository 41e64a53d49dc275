#!/bin/bash
# round 5, GPU call D: hash-grid backward in level-major order (A/B against SWN_HASH_POINT_MAJOR=1), the routing tests on the default
# (per-phase) path and the opt-in one-launch twin, the two-rank bench line with the expert-parallel probe
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
timeout 600 python -m pytest tests/test_hash_gpu.py tests/test_kernels_gpu.py -q -k "hash or route" 2>&1 | tail -6 > $O/d_hash_route.log
tail -3 $O/d_hash_route.log
H="python bench.py --hash --capacity-factor 1.25 --dtype fp16 --steps 10 --warmup 3 --no-cpu-baseline --no-balanced"
for i in 1 2; do
  timeout 300 $H > $O/d_hash_level_$i.json 2>/dev/null
  SWN_HASH_POINT_MAJOR=1 timeout 300 $H > $O/d_hash_point_$i.json 2>/dev/null
done
SWN_NO_OVERLAP=1 rocprofv3 --kernel-trace --stats -d gpurun_out/p_hash -o step -- $H --graph off --no-events --steps 5 > $O/d_phash.log 2>&1
python scripts/prof_summary.py $(find gpurun_out/p_hash -name "*.db" | head -1) 14 > $O/d_kernel_stats_hash.md
rm -rf gpurun_out/p_hash
timeout 900 python -m pytest tests/test_parallel_gpu.py -q -x 2>&1 | tail -8 > $O/d_parallel.log
tail -4 $O/d_parallel.log
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/d_*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], "ms/step", j["ms_per_step"], "value", j["value"], "eager", j["config"]["eager_ms_per_step"])
    except Exception as e:
        print(f, "ERR", e)
PY
head -8 $O/d_kernel_stats_hash.md | cut -c1-150
