#!/bin/bash
# round 6, GPU call K: compile-time epilogue variants in the other chain.hip builds (64-row fp32, 512-feature 4-wave, concat-skip): A/B against
# the dynamic epilogue (libswn_hip_dynepi.so = -DSWN_STATIC_EPI=0) on the recipes that run them; chain / dense / bg tests on the new build
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_dense_gpu.py tests/test_background_gpu.py -m gpu -q -x 2>&1 | tail -6 > $O/k_tests.log
tail -6 $O/k_tests.log
run() { # name, flags
  for rep in 1 2; do
    timeout 400 python bench.py $2 --steps 10 --warmup 3 --no-cpu-baseline --no-balanced > $O/k_$1_static_$rep.json 2>$O/k_$1_static_$rep.err
    SWN_LIB=$PWD/switch_nerf_amd/libswn_hip_dynepi.so timeout 400 python bench.py $2 --steps 10 --warmup 3 --no-cpu-baseline --no-balanced > $O/k_$1_dyn_$rep.json 2>$O/k_$1_dyn_$rep.err
  done
}
run dense "--dense"
run bg "--bg"
run mb "--mip --samples 257 --rays 3328 --chunk 212992 --model-dim 512 --experts 16"
run fp32 "--dtype fp32 --rays 2048"
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/k_*_*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], "ms/step", j["ms_per_step"], "loss", j["config"]["loss"])
    except Exception as e: print(f, "ERR", e, open(f.replace(".json",".err")).read()[-500:])
PY
