#!/bin/bash
# round 6, GPU call C: RCCL capture probe (hardened), the RCCL worker, the whole GPU suite on the hygiene build, bench line
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
timeout 1500 python scripts/experiments/rccl_capture_probe.py > $O/c_probe.log 2>&1
cat $O/c_probe.log | cut -c1-300
timeout 1800 python -m pytest tests -m gpu -q -x --deselect tests/test_rccl_gpu.py 2>&1 | tail -15 > $O/c_pytest.log
tail -15 $O/c_pytest.log
timeout 400 python bench.py --steps 20 --warmup 5 > $O/c_bench.json 2> $O/c_bench.err
python - <<PY
import json
j=json.loads(open("$O/c_bench.json").read().strip().splitlines()[-1]); k=j["kernels"]
print("step", j["ms_per_step"], j["value"], {n:k[n]["ms"] for n in ("expert_fwd","expert_bwd","expert_wgrad","expert_gemm_nosave") if n in k}, j["roofline"]["frac"])
PY
