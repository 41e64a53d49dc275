#!/bin/bash
# round 6, GPU call AD: soak of the final build - the default bench line six times (the driver's command), the suite once more in another order
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
for i in 1 2 3 4 5 6; do
  python bench.py --steps 20 --warmup 5 > $O/ad_bench_$i.json 2> $O/ad_bench_$i.err; echo "run $i rc $? $(python -c "import json;j=json.loads(open('$O/ad_bench_$i.json').read().splitlines()[-1]);print(j['ms_per_step'], j['roofline']['frac'], j['cpu_baseline']['value'])" 2>&1 | tail -1)"
done
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --deselect tests/test_rccl_gpu.py 2>&1 | tail -3
timeout 600 python -m pytest tests/test_rccl_gpu.py -m gpu -q 2>&1 | tail -2
