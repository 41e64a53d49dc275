#!/bin/bash
# round 6, GPU call M: work stealing between the per-XCD tile queues of chainq_kernel: twins + stress, A/B against the previous file
# (libswn_hip_nosteal.so) at 1024 rays per GPU (graph) and at full size, kernel table at 1024 rays
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_graph_gpu.py tests/test_determinism_gpu.py -m gpu -q -x 2>&1 | tail -6 > $O/m_tests.log
tail -6 $O/m_tests.log
for rep in 1 2 3; do
  for v in default nosteal; do
    L=$PWD/switch_nerf_amd/libswn_hip_$v.so; [ $v = default ] && L=$PWD/switch_nerf_amd/libswn_hip.so
    SWN_LIB=$L timeout 300 python bench.py --rays 1024 --steps 100 --warmup 20 --no-cpu-baseline --no-balanced --no-events > $O/m_1024_${v}_$rep.json 2>/dev/null
    SWN_LIB=$L timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-balanced > $O/m_full_${v}_$rep.json 2>/dev/null
  done
done
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/m_*_*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); k=j.get("kernels") or {}
        print(f.split("/")[-1], "ms/step", j["ms_per_step"], "kept", j["config"]["kept_token_fraction_mean"], {n:k[n]["ms"] for n in ("expert_fwd","expert_bwd") if n in k})
    except Exception as e: print(f, "ERR", e)
PY
SWN_NO_OVERLAP=1 rocprofv3 --kernel-trace --stats -d gpurun_out/p_m -o s -- python bench.py --rays 1024 --steps 30 --warmup 5 --no-cpu-baseline --no-balanced --graph off --no-events > $O/m_prof.log 2>&1
python scripts/prof_summary.py $(find gpurun_out/p_m -name "*.db" | head -1) 12 > $O/m_kernel_stats_1024rays.md
rm -rf gpurun_out/p_m
head -12 $O/m_kernel_stats_1024rays.md | cut -c1-150
