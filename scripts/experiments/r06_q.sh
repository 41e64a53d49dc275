#!/bin/bash
# round 6, GPU call Q: owner-tail expert parallelism (dropped tokens stay on the source): tests; bench through the supervising wrapper
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
timeout 1200 python -m pytest tests/test_model_gpu.py -m gpu -q -x -k "expert_parallel" 2>&1 | tail -8
timeout 1500 python -m pytest tests/test_parallel_gpu.py tests/test_rccl_gpu.py -m gpu -q -x 2>&1 | tail -12
timeout 300 python bench.py --steps 20 --warmup 5 > $O/q_bench.json 2> $O/q_bench.err; echo "bench rc $? bytes $(stat -c %s $O/q_bench.json)"
for f in "" "--ep-owner-tail"; do
  timeout 300 python bench.py --gpus 1 --loopback --parallelism ep $f --steps 10 --warmup 3 --no-cpu-baseline --no-balanced --no-events > $O/q_lbep$f.json 2>/dev/null; echo "loopback ep $f rc $?"
done
python - <<PY
import json
for f in ["q_bench", "q_lbep", "q_lbep--ep-owner-tail"]:
    try:
        j=json.loads([l for l in open("$O/"+f+".json").read().splitlines() if l.startswith("{")][-1]); print(f, j["ms_per_step"], j["value"], j["config"].get("parallelism"), (j["config"].get("expert_parallel") or {}).get("bytes_leaving_this_gpu_per_step"))
    except Exception as e: print(f, "ERR", e)
PY
