#!/bin/bash
# round 6, GPU call U: the 512-feature router on the matrix pipe (gate_fwd_wide / gate_bwd_wide): parity against fp64, timing against the VALU
# kernels, the Mission Bay recipe with both
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "router_16bit or gate" 2>&1 | tail -15
GATE_G=512 GATE_E=16 timeout 300 python scripts/gate_check.py > $O/u_gate_mfma.log 2>&1; grep -v amdgpu $O/u_gate_mfma.log | cut -c1-230
GATE_G=512 GATE_E=16 SWN_GATE_VALU=1 timeout 300 python scripts/gate_check.py > $O/u_gate_valu.log 2>&1; grep "tokens x" $O/u_gate_valu.log
MB="python bench.py --mip --samples 257 --rays 3328 --chunk 212992 --model-dim 512 --experts 16 --steps 10 --warmup 3 --no-cpu-baseline --no-balanced --no-events"
for i in 1 2; do
  timeout 300 $MB > $O/u_mb_mfma_$i.json 2>/dev/null
  SWN_GATE_VALU=1 timeout 300 $MB > $O/u_mb_valu_$i.json 2>/dev/null
done
python - <<PY
import json
for f in ["u_mb_mfma_1", "u_mb_valu_1", "u_mb_mfma_2", "u_mb_valu_2"]:
    try:
        j=json.loads([l for l in open("$O/"+f+".json").read().splitlines() if l.startswith("{")][-1]); print(f, j["ms_per_step"], j["value"], j["config"]["loss"])
    except Exception as e: print(f, "ERR", e)
PY
