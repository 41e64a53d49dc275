#!/bin/bash
# round 5, GPU call A: the suite on the SWN_KEEP build, the bench line, and the weight-sharing upper bound (libswn_hip_sharew.so:
# row group 1 of the expert launches skips its weight loads - results wrong, timing = what ANY sharing scheme could reach at most)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -30 > $O/a_pytest.log
tail -8 $O/a_pytest.log
timeout 400 python bench.py --steps 20 --warmup 5 > $O/a_bench.json 2> $O/a_bench.err
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-balanced"
for r in router balanced; do
  for rep in 1 2; do
    timeout 300 $B --routing $r > $O/a_prod_${r}_$rep.json 2>/dev/null
    SWN_LIB=$PWD/switch_nerf_amd/libswn_hip_sharew.so timeout 300 $B --routing $r > $O/a_sharew_${r}_$rep.json 2>/dev/null
  done
done
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/a_*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); k=j["kernels"]
        print(f.split("/")[-1], "step", j["ms_per_step"], {n:(k[n]["ms"], k[n]["mfma_frac"]) for n in ("expert_fwd","expert_bwd","expert_wgrad","expert_fwd_nosave","expert_gemm_nosave") if n in k})
    except Exception as e:
        print(f, "ERR", e)
PY
