#!/bin/bash
# round 6, GPU call T: owner-tail expert parallelism v2 (one all-to-all per payload over all segments, tail_bias_row in the fused launch, index
# kernels instead of torch indexing): tests, timing against the kept-rows mode and the eager data-parallel step (aliased and RCCL loopback)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -q -x -k "sign_bits or scatter_rows or ray_bias or expert_parallel or fused_tail or tail" 2>&1 | tail -6
timeout 900 python -m pytest tests/test_parallel_gpu.py tests/test_rccl_gpu.py -m gpu -q -x 2>&1 | tail -6
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-balanced --no-events"
timeout 300 $B --graph off > $O/t_dp_eager.json 2>/dev/null
timeout 300 $B --parallelism ep > $O/t_ep_local.json 2>/dev/null
timeout 300 $B --parallelism ep --ep-owner-tail > $O/t_ot_local.json 2>/dev/null
timeout 300 $B --gpus 1 --loopback --parallelism ep > $O/t_ep_loopback.json 2>/dev/null
timeout 300 $B --gpus 1 --loopback --parallelism ep --ep-owner-tail > $O/t_ot_loopback.json 2>$O/t_ot_loopback.err
python - <<PY
import json
for f in ["t_dp_eager", "t_ep_local", "t_ot_local", "t_ep_loopback", "t_ot_loopback"]:
    try:
        j=json.loads([l for l in open("$O/"+f+".json").read().splitlines() if l.startswith("{")][-1]); print(f, j["ms_per_step"], j["value"], j["config"].get("parallelism"), j["config"]["loss"])
    except Exception as e: print(f, "ERR", e)
PY
SWN_NO_OVERLAP=1 rocprofv3 --kernel-trace --stats -d gpurun_out/p_t -o t -- $B --parallelism ep --ep-owner-tail --steps 6 > $O/t_prof.log 2>&1
python scripts/prof_summary.py $(find gpurun_out/p_t -name "*.db" | head -1) 45 > $O/t_kernel_stats_owner_tail.md
rm -rf gpurun_out/p_t
head -40 $O/t_kernel_stats_owner_tail.md | cut -c1-150
