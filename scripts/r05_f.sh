#!/bin/bash
# round 5, GPU call F: the per-ray launches (ray_feat_fwd; ray_feat_wgrad + emb_grad) on the side stream - suite + A/B
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -12 > $O/f_pytest.log
tail -4 $O/f_pytest.log
for i in 1 2 3; do
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-balanced --no-events > $O/f_step_$i.json 2>/dev/null
  SWN_NO_SIDE_SMALL=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-balanced --no-events > $O/f_step_off_$i.json 2>/dev/null
  timeout 300 python bench.py --rays 1024 --steps 100 --warmup 20 --no-cpu-baseline --no-balanced --no-events > $O/f_1024_$i.json 2>/dev/null
  SWN_NO_SIDE_SMALL=1 timeout 300 python bench.py --rays 1024 --steps 100 --warmup 20 --no-cpu-baseline --no-balanced --no-events > $O/f_1024_off_$i.json 2>/dev/null
done
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/f_*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], "ms/step", j["ms_per_step"], "value", j["value"])
    except Exception as e: print(f, "ERR", e)
PY
