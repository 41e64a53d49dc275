#!/bin/bash
# the GPU suite + the bench lines: bash scripts/r04_suite.sh <tag>
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
T=${1:-s}; O=gpurun_out/r04; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q ${PYTEST_X--x} 2>&1 | tail -25 > $O/${T}_pytest.log
tail -6 $O/${T}_pytest.log
timeout 400 python bench.py --steps 20 --warmup 5 > $O/${T}_bench.json 2> $O/${T}_bench.err
python - <<PY
import json
j=json.loads(open("$O/${T}_bench.json").read().strip().splitlines()[-1]); k=j["kernels"]
print("step", j["ms_per_step"], "value", j["value"], "eager", j["config"]["eager_ms_per_step"], "runner", j["config"]["runner_loop_ms_per_step"], "balanced", j["config"]["balanced_value"])
print({n: (k[n]["ms"], k[n]["mfma_frac"]) for n in ("expert_fwd","expert_bwd","expert_wgrad","expert_fwd_nosave")})
print("roofline", {a: j["roofline"][a] for a in ("bound","frac","achieved","attainable")}); print("cpu", j["cpu_baseline"])
PY
timeout 300 python bench.py --rays 1024 --steps 100 --warmup 20 --no-cpu-baseline --no-balanced --no-events 2>/dev/null | tail -1 | cut -c1-200
timeout 300 python bench.py --eval --steps 50 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-200
