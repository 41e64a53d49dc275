#!/bin/bash
# round 6, GPU call P: (1) graph-drop stress with and without Tensor.record_stream on graph-pool tensors (the bench.py SIGSEGV hypothesis);
# (2) the default bench line x 8; (3) expert parallelism with the tail on the expert's rank: single-rank and two-rank tests
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
for i in 1 2 3; do
  SWN_EXP_RECORD_STREAM=1 timeout 300 python scripts/experiments/graph_drop_stress.py 40 > $O/p_stress_on_$i.log 2>&1; echo "record_stream ON  run $i rc $?: $(grep -v amdgpu $O/p_stress_on_$i.log | tail -2 | tr '\n' ' ' | cut -c1-200)"
  timeout 300 python scripts/experiments/graph_drop_stress.py 40 > $O/p_stress_off_$i.log 2>&1; echo "record_stream off run $i rc $?: $(grep -v amdgpu $O/p_stress_off_$i.log | tail -1 | cut -c1-200)"
done
export PYTHONFAULTHANDLER=1
for i in 1 2 3 4 5 6 7 8; do
  timeout 300 python bench.py > $O/p_default_$i.json 2> $O/p_default_$i.err; echo "default $i rc $? bytes $(stat -c %s $O/p_default_$i.json)"
done
timeout 1200 python -m pytest tests/test_model_gpu.py -m gpu -q -x -k "expert_parallel" 2>&1 | tail -25
timeout 1500 python -m pytest tests/test_parallel_gpu.py -m gpu -q -x 2>&1 | tail -25
