#!/bin/bash
# round 6, GPU call Y: the weight-gradient stream kernel's operand loads non-temporal (the build) against temporal (-DSWN_WG_NT=0, libswn_hip_wgt.so)
# on the Mission Bay recipe, whose 512-wide weight gradients are cut into 256-column blocks (every operand block is read by TWO jobs: a
# non-temporal line is the first to leave the L2 the second reader hopes to hit) - step time x 2 interleaved, FETCH_SIZE of the variant
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
MB="python bench.py --mip --samples 257 --rays 3328 --chunk 212992 --model-dim 512 --experts 16 --steps 10 --warmup 3 --no-cpu-baseline --no-balanced --no-events"
for i in 1 2; do
  timeout 300 $MB > $O/y_mb_nt_$i.json 2>/dev/null
  SWN_LIB=$GRAFT_REPO_ROOT/switch_nerf_amd/libswn_hip_wgt.so timeout 300 $MB > $O/y_mb_t_$i.json 2>/dev/null
done
for i in 1 2; do
  timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-balanced --no-events > $O/y_full_nt_$i.json 2>/dev/null
  SWN_LIB=$GRAFT_REPO_ROOT/switch_nerf_amd/libswn_hip_wgt.so timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-balanced --no-events > $O/y_full_t_$i.json 2>/dev/null
done
python - <<PY
import json
for f in ["y_mb_nt_1", "y_mb_t_1", "y_mb_nt_2", "y_mb_t_2", "y_full_nt_1", "y_full_t_1", "y_full_nt_2", "y_full_t_2"]:
    try:
        j=json.loads([l for l in open("$O/"+f+".json").read().splitlines() if l.startswith("{")][-1]); print(f, j["ms_per_step"], j["config"]["loss"])
    except Exception as e: print(f, "ERR", e)
PY
MB1="python bench.py --mip --samples 257 --rays 3328 --chunk 212992 --model-dim 512 --experts 16 --steps 1 --warmup 0 --no-cpu-baseline --no-balanced --graph off --no-events"
SWN_LIB=$GRAFT_REPO_ROOT/switch_nerf_amd/libswn_hip_wgt.so SWN_NO_OVERLAP=1 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/p_y -- $MB1 > $O/y_FETCH.log 2>&1
python scripts/pmc_summary.py gpurun_out/p_y wgrad_stream > $O/y_pmc_mb_FETCH_SIZE_temporal.txt; rm -rf gpurun_out/p_y
cat $O/y_pmc_mb_FETCH_SIZE_temporal.txt | cut -c1-150
