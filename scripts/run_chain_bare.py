import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from switch_nerf_amd import ops as o
dev = torch.device('cuda'); dt = torch.bfloat16
E, M, CAP, NSEG = 8, 256, 16384, 16
NG = NSEG * E; ROWS = NG * CAP
h0 = torch.randn(ROWS, M, device=dev).to(dt)
W = [o.pack_weights(torch.randn(E, M, M, device=dev).mul_(1 / 16), dt, True) for _ in range(8)]
y = torch.empty(ROWS, M, dtype=dt, device=dev)
layers = [o.Layer(W[l], None) for l in range(8)]
for _ in range(3):
    o.mlp_chain(h0, layers, y, n_groups=NG, n_wsets=E, group_stride=CAP, tag=1)
torch.cuda.synchronize()
