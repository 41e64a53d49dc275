"""What would two more layers on the persistent expert chain cost?  (the tail layers folded into the expert launch: upper bound of the gain)
python scripts/chain9_probe.py   -> ms per launch of the 7-, 8- and 9-layer chains, training forward and backward, full size."""
import sys
import torch
sys.path.insert(0, '.')
from switch_nerf_amd import ops as o
dev, dt = torch.device('cuda'), torch.bfloat16
M, E, CAP, nseg = 256, 8, 16384, 16
torch.manual_seed(0)


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


NG = nseg * E
ROWS = NG * CAP
h0 = torch.randn(ROWS, M, device=dev).to(dt)
perm = torch.randperm(ROWS, device=dev).int()
for pattern in ("full", "router"):
    fill = {"full": [1.0] * 8, "router": [1.0, 1.0, 1.0, 0.66, 0.66, 0.66, 0.66, 0.66]}[pattern]
    counts = torch.tensor([int(CAP * fill[g % E]) for g in range(NG)], dtype=torch.int32, device=dev)
    for L in (7, 8, 9):
        Wm = [torch.randn(E, M, M, device=dev).mul_(1 / 16) for _ in range(L)]
        Wf = [o.pack_weights(w, dt, True) for w in Wm]
        Wb = [o.pack_weights(w, dt, False) for w in Wm]
        B = [torch.randn(E, M, device=dev).mul_(0.1) for _ in range(L)]
        saves = [torch.empty(ROWS, M, dtype=dt, device=dev) for _ in range(L - 1)]
        dz = [torch.empty(ROWS, M, dtype=dt, device=dev) for _ in range(L - 1)]
        masks = [torch.zeros(o.chain_mask_words(dt, NG, CAP, M), dtype=torch.int32, device=dev) for _ in range(L - 1)]
        y = torch.empty(ROWS, M, dtype=dt, device=dev)
        dx = torch.empty(ROWS, M, dtype=dt, device=dev)
        fwd = [o.Layer(Wf[l], B[l], relu=1 if l < L - 1 else 0, skip=(l == 3), save=saves[l] if l < L - 1 else None,
                       mask=masks[l] if l < L - 1 else None) for l in range(L)]
        bwd = [o.Layer(Wb[l], None, relu=2 if l > 0 else 0, mask=masks[l - 1] if l > 0 else None, save=dz[l - 1] if l > 0 else None)
               for l in range(L - 1, -1, -1)]
        kw = dict(n_groups=NG, n_wsets=E, group_stride=CAP, group_rows=counts, group_rows_clamp=CAP, x_gather=perm)
        tf = bench(lambda: o.mlp_chain(h0, fwd, y, tag=1, geometry=7, **kw))
        tb = bench(lambda: o.mlp_chain(h0, bwd, dx, tag=2, geometry=7, y_add=dz[3], **kw))
        print(f"{pattern:6s} layers {L}: train_fwd {tf:.3f} ms, bwd {tb:.3f} ms", flush=True)
        del Wm, Wf, Wb, saves, dz
