#!/bin/bash
# round 6, GPU call W: does `python bench.py` die when it is the first process behind a rocprofv3 counter pass?  (both SIGSEGVs of the
# evidence collection were the first plain bench.py behind the --pmc passes)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
export PYTHONFAULTHANDLER=1 SWN_BENCH_CHILD=1     # (no supervising parent: the child's own exit status)
P1="python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-balanced --no-events --graph off"
for i in 1 2 3; do
  SWN_NO_OVERLAP=1 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/p_w -- $P1 > $O/w_pmc_$i.log 2>&1; echo "pmc pass $i rc $?"
  rm -rf gpurun_out/p_w
  timeout 300 python bench.py > $O/w_default_$i.json 2> $O/w_default_$i.err; echo "default behind pmc $i rc $? bytes $(stat -c %s $O/w_default_$i.json)"
  grep -v amdgpu $O/w_default_$i.err | tail -25 | cut -c1-200
  timeout 300 python bench.py > $O/w_default_b$i.json 2> $O/w_default_b$i.err; echo "default again $i rc $? bytes $(stat -c %s $O/w_default_b$i.json)"
done
