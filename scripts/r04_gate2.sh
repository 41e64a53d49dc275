#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
cat > /tmp/gt.py <<'PY'
import sys, torch
sys.path.insert(0, '.')
from switch_nerf_amd import ops as o
P, G, E = 852992, 512, 16
g = torch.randn(P, G, device='cuda').to(torch.bfloat16)
lw, lb = torch.ones(G, device='cuda'), torch.zeros(G, device='cuda')
wg = torch.randn(E, G, device='cuda') * 0.05
f = lambda: o.gate_fwd(g, lw, lb, wg)
f(); torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10): f()
b.record(); torch.cuda.synchronize()
print("gate_fwd 512 x 16, 852992 rows: %.3f ms" % (a.elapsed_time(b) / 10))
PY
for lib in switch_nerf_amd/libswn_hip_tb1.so "" switch_nerf_amd/libswn_hip_tb4.so; do echo "lib ${lib:-default(tb2)}"; SWN_LIB=$lib python /tmp/gt.py 2>&1 | grep gate_fwd; done
