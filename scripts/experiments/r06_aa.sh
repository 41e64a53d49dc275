#!/bin/bash
# round 6, GPU call AA: (i) route_scan_kernel<8, 128> for segments of 65-128 tiles (Mission Bay: 104) - routing tests; (ii) the 512-feature
# tail forward chain on the 8-wave kernel with the heads as their own launch (SWN_NO_FUSED_HEADS=1) against the 4-wave chain with fused heads:
# Mission Bay recipe, interleaved x 2, one box; kernel table of the unfused form
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_fullsize_gpu.py -m gpu -q -x -k "route or mission or wide" 2>&1 | tail -4
MB="python bench.py --mip --samples 257 --rays 3328 --chunk 212992 --model-dim 512 --experts 16 --steps 10 --warmup 3 --no-cpu-baseline --no-balanced --no-events"
for i in 1 2; do
  timeout 300 $MB > $O/aa_mb_fused_$i.json 2>/dev/null
  SWN_NO_FUSED_HEADS=1 timeout 300 $MB > $O/aa_mb_unfused_$i.json 2>/dev/null
done
python - <<PY
import json
for f in ["aa_mb_fused_1", "aa_mb_unfused_1", "aa_mb_fused_2", "aa_mb_unfused_2"]:
    try:
        j=json.loads([l for l in open("$O/"+f+".json").read().splitlines() if l.startswith("{")][-1]); print(f, j["ms_per_step"], j["config"]["loss"])
    except Exception as e: print(f, "ERR", e)
PY
MBP="python bench.py --mip --samples 257 --rays 3328 --chunk 212992 --model-dim 512 --experts 16 --steps 6 --warmup 3 --no-cpu-baseline --no-balanced --graph off --no-events"
SWN_NO_FUSED_HEADS=1 SWN_NO_OVERLAP=1 rocprofv3 --kernel-trace --stats -d gpurun_out/p_aa -o mb -- $MBP > $O/aa_p_mb.log 2>&1
python scripts/prof_summary.py $(find gpurun_out/p_aa -name "*.db" | head -1) 22 > $O/aa_kernel_stats_mission_bay_unfused.md
rm -rf gpurun_out/p_aa
cut -c1-140 $O/aa_kernel_stats_mission_bay_unfused.md
