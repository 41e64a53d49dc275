#!/bin/bash
# round 6, GPU call AG: resident workgroups of the persistent chains (SWN_CHAINQ_WGS; default = one per CU = 256) on a power-bound chip: do
# fewer CUs at a higher clock stream as much?  bench step + the per-launch times, interleaved x 2
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-balanced --no-ep-probe"
for rep in 1 2; do
  for n in 256 248 240 224 208 192; do
    SWN_CHAINQ_WGS=$n timeout 300 $B > $O/ag_${n}_$rep.json 2>/dev/null
  done
done
python - <<PY
import json,glob
for n in (256,248,240,224,208,192):
    for rep in (1,2):
        try:
            j=json.loads(open("$O/ag_%d_%d.json"%(n,rep)).read().strip().splitlines()[-1]); k=j["kernels"]
            print(n, rep, "step", j["ms_per_step"], {q:round(k[q]["ms"],3) for q in ("expert_fwd","expert_bwd","expert_wgrad","expert_gemm_nosave") if q in k})
        except Exception as e:
            print(n, rep, "ERR", e)
PY
