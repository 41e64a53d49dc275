#!/bin/bash
# round 5, final: the GPU suite and the round's evidence from ONE build
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -30 > $O/final_pytest.log
tail -5 $O/final_pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash scripts/collect_profiles_r05.sh > gpurun_out/r05_collect.log 2>&1
tail -3 $O/make_traffic.log
python - <<PY
import json
for f in ("r05_bench_default","r05_bench_20_5","r05_bench_1024rays_graph","r05_bench_eval_graph"):
    j=json.loads(open("$O/"+f+".json").read().strip().splitlines()[-1]); r=j.get("roofline") or {}
    print(f, j["ms_per_step"], j["value"], r.get("frac"), r.get("traffic"), (r.get("traffic_source") or "")[:50])
PY
