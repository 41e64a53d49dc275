#!/bin/bash
# round 6, GPU call AB: pack_weights_batched with 16-byte loads (backward-data layout) and one 16-byte store per lane: equality test, chain tests,
# kernel time in the Mission Bay recipe and the headline (rocprofv3 tables), steps x 2
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -q -x -k "pack or chain or train or golden" 2>&1 | tail -4
MB="python bench.py --mip --samples 257 --rays 3328 --chunk 212992 --model-dim 512 --experts 16 --steps 10 --warmup 3 --no-cpu-baseline --no-balanced --no-events"
for i in 1 2; do timeout 300 $MB > $O/ab_mb_$i.json 2>/dev/null; timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-balanced --no-events > $O/ab_full_$i.json 2>/dev/null; done
python - <<PY
import json
for f in ["ab_mb_1", "ab_mb_2", "ab_full_1", "ab_full_2"]:
    try:
        j=json.loads([l for l in open("$O/"+f+".json").read().splitlines() if l.startswith("{")][-1]); print(f, j["ms_per_step"], j["config"]["loss"])
    except Exception as e: print(f, "ERR", e)
PY
MBP="python bench.py --mip --samples 257 --rays 3328 --chunk 212992 --model-dim 512 --experts 16 --steps 6 --warmup 3 --no-cpu-baseline --no-balanced --graph off --no-events"
SWN_NO_OVERLAP=1 rocprofv3 --kernel-trace --stats -d gpurun_out/p_ab -o mb -- $MBP > $O/ab_p_mb.log 2>&1
python scripts/prof_summary.py $(find gpurun_out/p_ab -name "*.db" | head -1) 40 | grep -i "pack\|adam\|scan\|total" 
rm -rf gpurun_out/p_ab
SWN_NO_OVERLAP=1 rocprofv3 --kernel-trace --stats -d gpurun_out/p_ab2 -o st -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-balanced --graph off --no-events > $O/ab_p_full.log 2>&1
python scripts/prof_summary.py $(find gpurun_out/p_ab2 -name "*.db" | head -1) 45 | grep -i "pack\|adam\|total"
rm -rf gpurun_out/p_ab2
