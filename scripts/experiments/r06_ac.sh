#!/bin/bash
# round 6, GPU call AC: bench.py's N > 1 line with BOTH expert-parallel probes behind the data-parallel headline (kept rows per segment; tail on
# the expert's rank), rehearsed over the world-1 RCCL loopback group
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
for r in 1024 8192; do
python bench.py --gpus 1 --loopback --rays $r --steps 30 --warmup 10 --no-cpu-baseline --no-balanced --no-events 2>$O/ac_probe_$r.err | tail -1 > $O/ac_probe_$r.json
python - <<PY
import json
j=json.loads(open("$O/ac_probe_$r.json").read())
print($r, j["ms_per_step"], j["value"], j["scaling"])
for k in ("expert_parallel","expert_parallel_owner_tail"):
    x=j["config"].get(k) or {}; print(k, {q:x.get(q) for q in ("ms_per_step","value","owner_tail","collectives_per_step","error","loss","bytes_leaving_this_gpu_per_step")})
PY
done
