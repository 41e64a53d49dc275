#!/bin/bash
# round 6, GPU call Z: per-operand cache policy of the weight-gradient stream kernel (shared column blocks: default policy; everything else
# non-temporal as before): tests, Mission Bay recipe and the headline x 2, FETCH_SIZE of the recipe's weight-gradient launches
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_fullsize_gpu.py -m gpu -q -x -k "wgrad or mission or wide" 2>&1 | tail -4
MB="python bench.py --mip --samples 257 --rays 3328 --chunk 212992 --model-dim 512 --experts 16 --steps 10 --warmup 3 --no-cpu-baseline --no-balanced --no-events"
for i in 1 2; do timeout 300 $MB > $O/z_mb_$i.json 2>/dev/null; timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-balanced --no-events > $O/z_full_$i.json 2>/dev/null; done
python - <<PY
import json
for f in ["z_mb_1", "z_mb_2", "z_full_1", "z_full_2"]:
    try:
        j=json.loads([l for l in open("$O/"+f+".json").read().splitlines() if l.startswith("{")][-1]); print(f, j["ms_per_step"], j["config"]["loss"])
    except Exception as e: print(f, "ERR", e)
PY
MB1="python bench.py --mip --samples 257 --rays 3328 --chunk 212992 --model-dim 512 --experts 16 --steps 1 --warmup 0 --no-cpu-baseline --no-balanced --graph off --no-events"
SWN_NO_OVERLAP=1 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/p_z -- $MB1 > $O/z_FETCH.log 2>&1
python scripts/pmc_summary.py gpurun_out/p_z wgrad_stream > $O/z_pmc_mb_FETCH_SIZE.txt; rm -rf gpurun_out/p_z
cat $O/z_pmc_mb_FETCH_SIZE.txt | cut -c1-150
