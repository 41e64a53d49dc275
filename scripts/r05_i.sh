#!/bin/bash
# round 5, GPU call I: does the router's read of g hit the Infinity Cache when it starts with the rows the front chain wrote last?
# variants: rev = gate_fwd walks its tiles backwards; y0 = the chains' OUTPUT stores are write-back instead of non-temporal; revy0 = both
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
for i in 1 2; do
  for v in prod rev revy0 y0; do
    L=$PWD/switch_nerf_amd/libswn_hip.so; [ $v != prod ] && L=$PWD/switch_nerf_amd/libswn_hip_$v.so
    SWN_LIB=$L timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-balanced --no-events > $O/i_${v}_$i.json 2>/dev/null
  done
done
for v in prod rev revy0 y0; do
  L=$PWD/switch_nerf_amd/libswn_hip.so; [ $v != prod ] && L=$PWD/switch_nerf_amd/libswn_hip_$v.so
  SWN_LIB=$L SWN_NO_OVERLAP=1 rocprofv3 --kernel-trace --stats -d gpurun_out/p_i$v -o step -- python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-balanced --graph off --no-events > $O/i_p$v.log 2>&1
  python scripts/prof_summary.py $(find gpurun_out/p_i$v -name "*.db" | head -1) 12 > $O/i_kernel_stats_$v.md
  rm -rf gpurun_out/p_i$v
done
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/i_*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], "ms/step", j["ms_per_step"])
    except Exception as e: print(f, "ERR", e)
PY
grep -h "gate_fwd\|Bf16, 3, true\|Bf16, 7, true\|Bf16, 8, true\|Bf16, 6, true\|gate_bwd" $O/i_kernel_stats_*.md | cut -c1-140
