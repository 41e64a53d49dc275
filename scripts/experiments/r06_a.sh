#!/bin/bash
# round 6, GPU call A: same-box A/B of chain_big.hip: round-4 holds (r4chain = the file at 15f2c65), round-5 HEAD
# (e0s0), and the two halves of the fix (e1 = round-4 order of the E-phase write-out with a structural hold of the last batch,
# s1 = the S phase's register set defined without instructions / trimmed for the fused tail); default build = e1s1.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-balanced --no-ep-probe"
for rep in 1 2 3; do
  for v in default r4chain e0s0 e1s0 e0s1; do
    L=$PWD/switch_nerf_amd/libswn_hip_$v.so; [ $v = default ] && L=$PWD/switch_nerf_amd/libswn_hip.so
    SWN_LIB=$L timeout 300 $B > $O/a_${v}_$rep.json 2>$O/a_${v}_$rep.err
  done
done
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/a_*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); k=j["kernels"]
        print(f.split("/")[-1], "step", j["ms_per_step"], {n:k[n]["ms"] for n in ("expert_fwd","expert_bwd","expert_wgrad","expert_fwd_nosave","expert_gemm_nosave","square_gemm_8192") if n in k})
    except Exception as e:
        print(f, "ERR", e)
PY
# the stress test + the twins on the default build
timeout 900 python -m pytest tests/test_graph_gpu.py tests/test_kernels_gpu.py -m gpu -q -x 2>&1 | tail -8 > $O/a_pytest.log
tail -8 $O/a_pytest.log
# RCCL world-1 loopback tests
timeout 1200 python -m pytest tests/test_rccl_gpu.py -m gpu -q 2>&1 | tail -40 > $O/a_rccl.log
tail -40 $O/a_rccl.log
