#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_graph_gpu.py -m gpu -q -x -k "geometry_bit_exact or fused_combine or front_chains or 200_back_to_back" 2>&1 | grep -E "Error|assert|passed|failed|differ" | head -20
timeout 200 python scripts/chainq_timing.py 4 7 2>&1 | grep "segments 16"
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-balanced"
$B 2>/dev/null | tail -1 | python -c "
import sys,json; j=json.loads(sys.stdin.read()); k=j['kernels']
print('step', j['ms_per_step'], 'fwd', k['expert_fwd']['ms'], 'bwd', k['expert_bwd']['ms'], 'nosave', k['expert_fwd_nosave']['ms'], k['expert_fwd_nosave']['mfma_frac'])"
