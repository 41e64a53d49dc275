#!/bin/bash
# round 6, GPU call N: hash tile pass with 16-byte loads, static per-ray-bias epilogue variant (4-wave 512-feature tail chain): tests + recipes
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
timeout 1500 python -m pytest tests/test_hash_gpu.py tests/test_kernels_gpu.py tests/test_dense_gpu.py tests/test_background_gpu.py -m gpu -q -x 2>&1 | tail -5 > $O/n_tests.log
tail -5 $O/n_tests.log
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_fullsize_gpu.py -m gpu -q -x -k "mission or 512 or mip or hash or configs" 2>&1 | tail -4
for rep in 1 2; do
  timeout 300 python bench.py --hash --capacity-factor 1.25 --dtype fp16 --steps 10 --warmup 3 --no-cpu-baseline --no-balanced > $O/n_hash_$rep.json 2>/dev/null
  timeout 400 python bench.py --mip --samples 257 --rays 3328 --chunk 212992 --model-dim 512 --experts 16 --steps 10 --warmup 3 --no-cpu-baseline --no-balanced > $O/n_mb_$rep.json 2>/dev/null
done
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/n_*_*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], "ms/step", j["ms_per_step"], "loss", j["config"]["loss"])
    except Exception as e: print(f, "ERR", e)
PY
SWN_NO_OVERLAP=1 rocprofv3 --kernel-trace --stats -d gpurun_out/p_n -o h -- python bench.py --hash --capacity-factor 1.25 --dtype fp16 --steps 6 --warmup 3 --no-cpu-baseline --no-balanced --graph off --no-events > $O/n_prof.log 2>&1
python scripts/prof_summary.py $(find gpurun_out/p_n -name "*.db" | head -1) 16 | grep -i "hash\|kernel |" | cut -c1-150
rm -rf gpurun_out/p_n
