import sys, os, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import synth
from oracle import switchnerf_oracle as O
from switch_nerf_amd import ops
N, S = 37, 64
rays, _, _ = synth.make_rays(7, N)
r = torch.from_numpy(rays)
zr = O.sample_z(r[:, 6:7], r[:, 7:8], S, 0.0, None)
xyz = (r[:, None, :3] + r[:, None, 3:6] * zr[:, :, None]).reshape(-1, 3)
t = torch.linspace(0, 1, S)
z, pe, pd = ops.sample_pe(r.cuda(), t.cuda(), None, 0.0, S, 12, 4, torch.float32, 128, 32)
pe = pe.cpu()
print("z equal:", torch.equal(z.cpu(), zr))
print("xyz bit-equal:", torch.equal(pe[:, :3], xyz), "n diff", (pe[:, :3] != xyz).sum().item())
x64 = pe[:, :3].double()
for k in (0, 5, 11):
    f = 2.0 ** k
    s64 = torch.sin(f * x64); c64 = torch.cos(f * x64)
    gs = pe[:, 3 + 6 * k: 6 + 6 * k].double(); gc = pe[:, 6 + 6 * k: 9 + 6 * k].double()
    ts = torch.sin(f * pe[:, :3]).double(); tc = torch.cos(f * pe[:, :3]).double()
    print(f"k={k}: gpu sin err {(gs - s64).abs().max():.3e} cos err {(gc - c64).abs().max():.3e} | torch-cpu sin err {(ts - s64).abs().max():.3e} cos err {(tc - c64).abs().max():.3e}")
