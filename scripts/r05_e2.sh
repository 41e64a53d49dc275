#!/bin/bash
# the two HBM counter passes again (one step each) + traffic.json + the bench lines that read it
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
R=r05; O=gpurun_out/$R; mkdir -p $O
P1="python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-balanced --no-events --graph off"
for c in FETCH_SIZE WRITE_SIZE; do
  SWN_NO_OVERLAP=1 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/p_$c -- $P1 > $O/p_$c.log 2>&1
  python scripts/pmc_summary.py gpurun_out/p_$c > $O/${R}_pmc_$c.txt
  grep '^{' $O/p_$c.log | tail -1 >> $O/${R}_pmc_$c.txt
  rm -rf gpurun_out/p_$c
done
cp $O/${R}_pmc_FETCH_SIZE.txt $O/${R}_pmc_WRITE_SIZE.txt profiles/
python scripts/make_traffic.py $R > $O/make_traffic.log 2>&1; cp profiles/traffic.json $O/traffic.json
python bench.py > $O/${R}_bench_default.json 2> $O/${R}_bench_default.err
python bench.py --steps 20 --warmup 5 > $O/${R}_bench_20_5.json 2>/dev/null
tail -3 $O/make_traffic.log; python - <<PY
import json
j=json.loads(open("$O/${R}_bench_20_5.json").read().strip().splitlines()[-1]); r=j["roofline"]
print(j["ms_per_step"], r["frac"], r["traffic"], r.get("hbm_frac_measured"), r["traffic_source"][:60]); print({k:(v["ms"],v.get("hbm_measured_over_alg")) for k,v in j["kernels"].items() if isinstance(v,dict) and "ms" in v})
PY
