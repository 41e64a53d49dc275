#!/bin/bash
# round 6, GPU call D: RCCL capture workaround + teardown variants, RCCL tests, binned hash backward (tests, A/B against the atomics, kernel table), suite
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
timeout 900 python scripts/experiments/rccl_capture_probe.py --only a2a_sync_delgraph,a2a_sync_abort,ep_all_to_all_v,ep_exchange_counts > $O/d_probe.log 2>&1
cat $O/d_probe.log | cut -c1-300
timeout 1200 python -m pytest tests/test_rccl_gpu.py tests/test_hash_gpu.py -m gpu -q 2>&1 | tail -40 > $O/d_tests.log
tail -40 $O/d_tests.log
H="python bench.py --hash --capacity-factor 1.25 --dtype fp16 --steps 10 --warmup 3 --no-cpu-baseline --no-balanced"
for rep in 1 2; do
  timeout 300 $H > $O/d_hash_binned_$rep.json 2>$O/d_hash_binned_$rep.err
  SWN_HASH_BWD=atomic timeout 300 $H > $O/d_hash_atomic_$rep.json 2>$O/d_hash_atomic_$rep.err
done
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/d_hash_*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], "ms/step", j["ms_per_step"], "loss", j["config"]["loss"])
    except Exception as e: print(f, "ERR", e, open(f.replace(".json",".err")).read()[-800:])
PY
rocprofv3 --kernel-trace --stats -d gpurun_out/p_d -o hash -- $H --graph off --no-events --steps 6 > $O/d_prof.log 2>&1
python scripts/prof_summary.py $(find gpurun_out/p_d -name "*.db" | head -1) 40 > $O/d_kernel_stats_hash.md
rm -rf gpurun_out/p_d
head -30 $O/d_kernel_stats_hash.md | cut -c1-160
timeout 1800 python -m pytest tests -m gpu -q -x --deselect tests/test_rccl_gpu.py --deselect tests/test_hash_gpu.py 2>&1 | tail -15 > $O/d_pytest.log
tail -15 $O/d_pytest.log
