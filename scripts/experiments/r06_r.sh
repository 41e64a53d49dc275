#!/bin/bash
# round 6, GPU call S: owner-tail expert parallelism with the sign-bit kernels and aliased buffers at W = 1: tests, timing, kernel table
# expert-parallel path and the eager data-parallel step; kernel table of the owner-tail step
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-balanced --no-events"
timeout 300 $B --graph off > $O/s_dp_eager.json 2>/dev/null
timeout 300 $B --parallelism ep > $O/s_ep_local.json 2>/dev/null
timeout 300 $B --parallelism ep --ep-owner-tail > $O/s_ot_local.json 2>/dev/null
python - <<PY
import json
for f in ["s_dp_eager", "s_ep_local", "s_ot_local"]:
    try:
        j=json.loads([l for l in open("$O/"+f+".json").read().splitlines() if l.startswith("{")][-1]); print(f, j["ms_per_step"], j["value"], j["config"].get("parallelism"), j["config"]["loss"])
    except Exception as e: print(f, "ERR", e)
PY
SWN_NO_OVERLAP=1 rocprofv3 --kernel-trace --stats -d gpurun_out/p_s -o s -- $B --parallelism ep --ep-owner-tail --steps 6 > $O/s_prof.log 2>&1
python scripts/prof_summary.py $(find gpurun_out/p_s -name "*.db" | head -1) 45 > $O/s_kernel_stats_owner_tail.md
rm -rf gpurun_out/p_s
head -48 $O/s_kernel_stats_owner_tail.md | cut -c1-150
