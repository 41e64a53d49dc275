#!/bin/bash
# round 5, GPU call K: the gate-noise branch (swn_gate_fwd_noise, MoELayer / SwitchNeRF gate_noise) - its tests, then the whole suite
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -q -x -k "gate_noise or gate_fwd_with" 2>&1 | tail -25 > $O/k_noise.log
tail -25 $O/k_noise.log
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -8 > $O/k_pytest.log
tail -4 $O/k_pytest.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-balanced --no-events 2>/dev/null | tail -1 | cut -c1-200
