#!/bin/bash
# round 6, GPU call L: the whole GPU suite on the current build + the headline bench line
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -25 > $O/l_pytest.log
tail -25 $O/l_pytest.log
timeout 400 python bench.py --steps 20 --warmup 5 > $O/l_bench.json 2> $O/l_bench.err
python - <<PY
import json
j=json.loads(open("$O/l_bench.json").read().strip().splitlines()[-1]); k=j["kernels"]
print("step", j["ms_per_step"], j["value"], j["scaling"], {n:k[n]["ms"] for n in ("expert_fwd","expert_bwd","expert_wgrad","expert_gemm_nosave") if n in k}, j["roofline"]["frac"], j["roofline"].get("traffic"))
PY
