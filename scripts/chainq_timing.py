"""Expert chains at full size: geometry 4 (one launch slot per 256-row tile) against geometry 7 (persistent workgroups, tile queue).
python scripts/chainq_timing.py [geometries, default "4 7"]   -> ms per launch, MFMA fraction (of 2.5 PFLOP/s) on the kept rows."""
import os
import sys
import torch
sys.path.insert(0, '.')
from switch_nerf_amd import ops as o
dev, dt = torch.device('cuda'), torch.bfloat16
M, E, L, CAP = 256, 8, 7, 16384
geoms = [int(a) for a in sys.argv[1:]] or [4, 7]
torch.manual_seed(0)
ZERO = bool(os.environ.get("ZERO"))      # all-zero weights and inputs: what the matrix pipe's POWER costs (nothing toggles)
Wm = [torch.randn(E, M, M, device=dev).mul_(0.0 if ZERO else 1 / 16) for _ in range(L)]
Wf = [o.pack_weights(w, dt, True) for w in Wm]
Wb = [o.pack_weights(w, dt, False) for w in Wm]
B = [torch.randn(E, M, device=dev).mul_(0.1) for _ in range(L)]


def bench(fn, n=5):
    fn(); fn()
    torch.cuda.synchronize()
    best = None
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            fn()
        b.record()
        torch.cuda.synchronize()
        t = a.elapsed_time(b) / n
        best = t if best is None else min(best, t)
    return best


for nseg, pattern in ((16, "full"), (16, "router"), (2, "full"), (2, "router")):
    NG = nseg * E
    ROWS = NG * CAP
    fill = {"full": [1.0] * 8, "router": [1.0, 1.0, 1.0, 0.66, 0.66, 0.66, 0.66, 0.66]}[pattern]
    counts = torch.tensor([int(CAP * fill[g % E]) for g in range(NG)], dtype=torch.int32, device=dev)
    kept = int(counts.sum().item())
    h0 = torch.randn(ROWS, M, device=dev).mul_(0.0 if ZERO else 1.0).to(dt)
    perm = torch.randperm(ROWS, device=dev).int()
    saves = [torch.empty(ROWS, M, dtype=dt, device=dev) for _ in range(L - 1)]
    dz = [torch.empty(ROWS, M, dtype=dt, device=dev) for _ in range(L - 1)]
    masks = [torch.zeros(o.chain_mask_words(dt, NG, CAP, M), dtype=torch.int32, device=dev) for _ in range(L - 1)]
    y = torch.empty(ROWS, M, dtype=dt, device=dev)
    dx = torch.empty(ROWS, M, dtype=dt, device=dev)
    fwd = [o.Layer(Wf[l], B[l], relu=1 if l < L - 1 else 0, skip=(l == 3), save=saves[l] if l < L - 1 else None,
                   mask=masks[l] if l < L - 1 else None) for l in range(L)]
    inf = [o.Layer(Wf[l], B[l], relu=1 if l < L - 1 else 0, skip=(l == 3)) for l in range(L)]
    bwd = [o.Layer(Wb[l], None, relu=2 if l > 0 else 0, mask=masks[l - 1] if l > 0 else None, save=dz[l - 1] if l > 0 else None)
           for l in range(L - 1, -1, -1)]
    kw = dict(n_groups=NG, n_wsets=E, group_stride=CAP, group_rows=counts, group_rows_clamp=CAP, x_gather=perm)
    for geom in geoms:
        res = {}
        res["train_fwd"] = bench(lambda: o.mlp_chain(h0, fwd, y, tag=1, geometry=geom, **kw))
        res["nosave"] = bench(lambda: o.mlp_chain(h0, inf, y, tag=1, geometry=geom, **kw))
        res["bwd"] = bench(lambda: o.mlp_chain(h0, bwd, dx, tag=2, geometry=geom, y_add=dz[3], **kw))
        fl = 2.0 * L * M * M * kept
        print(f"segments {nseg:2d} {pattern:6s} kept {kept:8d} geometry {geom}: " +
              ", ".join(f"{k} {v:.3f} ms ({fl / (v * 1e-3) / 2.5e15:.3f})" for k, v in res.items()), flush=True)
