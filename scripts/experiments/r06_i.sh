#!/bin/bash
# round 6, GPU call I (+ conflict-free LDS writes and reads in the epilogue): 8-wave 512-feature chain with the write-out inside the next EPILOGUE (column blocks), bias / mask hoisted:
# phase timers (timing build), chain tests, Mission Bay recipe (3 runs), kernel table
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_fullsize_gpu.py tests/test_model_gpu.py -m gpu -q -x -k "wide or 512 or mission or configs3 or mip" 2>&1 | tail -8 > $O/i_tests.log
tail -8 $O/i_tests.log
for m in bare fwd bwd; do
  SWN_LIB=$PWD/switch_nerf_amd/libswn_hip_timingw.so timeout 300 python scripts/experiments/chain_wide_timing.py $m 2>&1 | grep -v amdgpu.ids | tail -11 | tee -a $O/i_chain_wide_timing.txt
  timeout 300 python scripts/experiments/chain_wide_timing.py $m 2>&1 | grep "^mode" | tee -a $O/i_chain_wide_timing.txt
done
MB="python bench.py --mip --samples 257 --rays 3328 --chunk 212992 --model-dim 512 --experts 16 --steps 10 --warmup 3 --no-cpu-baseline --no-balanced"
for rep in 1 2 3; do timeout 400 $MB > $O/i_mb_$rep.json 2>$O/i_mb_$rep.err; done
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/i_mb_*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], "ms/step", j["ms_per_step"], "loss", j["config"]["loss"])
    except Exception as e: print(f, "ERR", e, open(f.replace(".json",".err")).read()[-800:])
PY
SWN_NO_OVERLAP=1 rocprofv3 --kernel-trace --stats -d gpurun_out/p_i -o mb -- $MB --graph off --no-events --steps 6 > $O/i_prof.log 2>&1
python scripts/prof_summary.py $(find gpurun_out/p_i -name "*.db" | head -1) 30 > $O/i_kernel_stats_mission_bay.md
rm -rf gpurun_out/p_i
head -20 $O/i_kernel_stats_mission_bay.md | cut -c1-170
