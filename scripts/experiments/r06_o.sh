#!/bin/bash
# round 6, GPU call O: the intermittent SIGSEGV of `python bench.py` seen in the evidence collection: default bench x 6, loopback x 3 each,
# with PYTHONFAULTHANDLER (a crash prints its Python stack)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
export PYTHONFAULTHANDLER=1
for i in 1 2 3 4 5 6; do
  timeout 300 python bench.py > $O/o_default_$i.json 2> $O/o_default_$i.err; echo "default $i rc $? bytes $(stat -c %s $O/o_default_$i.json)"
  grep -v amdgpu.ids $O/o_default_$i.err | tail -25
done
for i in 1 2 3; do
  timeout 300 python bench.py --gpus 1 --loopback --rays 1024 --steps 50 --warmup 10 --no-cpu-baseline --no-balanced --no-events > $O/o_lbdp_$i.json 2> $O/o_lbdp_$i.err; echo "loopback dp $i rc $? bytes $(stat -c %s $O/o_lbdp_$i.json)"
  grep -v "amdgpu.ids\|hostname of the client" $O/o_lbdp_$i.err | tail -25
  timeout 300 python bench.py --gpus 1 --loopback --parallelism ep --steps 10 --warmup 3 --no-cpu-baseline --no-balanced --no-events > $O/o_lbep_$i.json 2> $O/o_lbep_$i.err; echo "loopback ep $i rc $? bytes $(stat -c %s $O/o_lbep_$i.json)"
  grep -v "amdgpu.ids\|hostname of the client" $O/o_lbep_$i.err | tail -25
done
