#!/bin/bash
# round 6, GPU call E: binned hash backward v2 (all levels per workgroup, ballot ranks on dense levels, 4 items in flight in the tile pass):
# tests, recipe timing, kernel table; phase timers of the 8-wave 512-feature chain (timing build)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
timeout 600 python -m pytest tests/test_hash_gpu.py -m gpu -q 2>&1 | tail -15 > $O/e_tests.log
tail -15 $O/e_tests.log
H="python bench.py --hash --capacity-factor 1.25 --dtype fp16 --steps 10 --warmup 3 --no-cpu-baseline --no-balanced"
for rep in 1 2; do
  timeout 300 $H > $O/e_hash_binned_$rep.json 2>$O/e_hash_binned_$rep.err
done
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/e_hash_*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], "ms/step", j["ms_per_step"], "loss", j["config"]["loss"])
    except Exception as e: print(f, "ERR", e, open(f.replace(".json",".err")).read()[-800:])
PY
SWN_NO_OVERLAP=1 rocprofv3 --kernel-trace --stats -d gpurun_out/p_e -o hash -- $H --graph off --no-events --steps 6 > $O/e_prof.log 2>&1
python scripts/prof_summary.py $(find gpurun_out/p_e -name "*.db" | head -1) 40 > $O/e_kernel_stats_hash.md
rm -rf gpurun_out/p_e
head -24 $O/e_kernel_stats_hash.md | cut -c1-160
for m in bare fwd bwd; do
  SWN_LIB=$PWD/switch_nerf_amd/libswn_hip_timingw.so timeout 300 python scripts/experiments/chain_wide_timing.py $m 2>&1 | tail -12 | tee -a $O/e_chain_wide_timing.txt
  timeout 300 python scripts/experiments/chain_wide_timing.py $m 2>&1 | head -1 | tee -a $O/e_chain_wide_timing.txt
done
