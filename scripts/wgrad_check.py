"""Expert weight-gradient launch (7 layers, 128 groups x 16384 rows) under different routing balances: ms and ns per kept row.
python scripts/wgrad_check.py [splits ...]"""
import sys, torch
sys.path.insert(0, '.')
from switch_nerf_amd import ops as o
dev, dt = torch.device('cuda'), torch.bfloat16
M, E, L, CAP, NSEG = 256, 8, 7, 16384, 16
NG = NSEG * E
ROWS = NG * CAP
torch.manual_seed(0)
import os
_mk = (lambda: torch.zeros(ROWS, M, device=dev, dtype=dt)) if os.environ.get("ZERO") else (lambda: torch.randn(ROWS, M, device=dev).to(dt))
acts = [_mk() for _ in range(L)]      # ZERO=1: all-zero operands (what the matrix pipe's POWER costs: zeros toggle nothing)
dzs = [_mk() for _ in range(L)]
perm = torch.randperm(ROWS, device=dev).int()
import os
if os.environ.get("PERM") == "identity":        # gather order experiments: the two gathered operands read in row order ...
    perm = torch.arange(ROWS, device=dev, dtype=torch.int32)
elif os.environ.get("PERM") == "sorted":        # ... or ascending inside every (segment, expert) group (what a token-ordered row layout would give)
    perm = perm.view(NG, CAP).sort(dim=1).values.reshape(-1).contiguous()
dw = [torch.zeros(E, M, M, device=dev) for _ in range(L)]
db = [torch.zeros(E, M, device=dev) for _ in range(L)]
if os.environ.get("PERM") == "none":            # no gathered operand at all (what the index loads of the two gathered streams cost)
    perm = None
items = [(acts[l], dzs[l], dw[l], db[l], perm if l == 0 else None, perm if l == L - 1 else None) for l in range(L)]
pats = {"balanced": [CAP] * 8,
        "all 79 %": [int(CAP * 0.79)] * 8,
        "half full, half 58 %": [CAP, int(CAP * 0.58)] * 4,
        "router-like (3 full, 5 x 66 %)": [CAP, CAP, CAP] + [int(CAP * 0.664)] * 5,
        "2 full, 6 x 72 %": [CAP, CAP] + [int(CAP * 0.72)] * 6}
for splits in [int(v) for v in sys.argv[1:]] or [2]:
    for name, pe in pats.items():
        counts = torch.tensor(pe * NSEG, dtype=torch.int32, device=dev)
        kept = int(counts.sum())
        f = lambda: o.wgrad_batched(items, n_groups=NG, n_wsets=E, group_stride=CAP, group_rows=counts, group_rows_clamp=CAP, n_splits=splits, tag=1)
        f(); f(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5):
            f()
        b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 5
        print(f"splits {splits} {name:32s} kept {kept / ROWS:.3f}: {ms:.3f} ms, {ms * 1e6 / kept:.3f} ns per kept row, {kept * 7 * 1024 / ms / 1e9:.2f} TB/s of operand reads")

# ---- the five dense-layer weight gradients of the step (2,097,152 points): per-layer launches vs the two batched launches
P, KP, H2 = 8192 * 256, 128, 128
t = lambda c: torch.randn(P, c, device=dev).to(dt)
h1, dh2, y, dh1, a1, dg, h0, dza1, pe, dh0 = t(M), t(H2), t(M), t(M), t(M), t(M), t(M), t(M), t(KP), t(M)
z = lambda *s: torch.zeros(*s, device=dev)
tail = [(h1, dh2, z(1, M, H2), None), (y, dh1, z(1, M, M), z(1, M))]
front = [(a1, dg, z(1, M, M), z(1, M)), (h0, dza1, z(1, M, M), z(1, M)), (pe, dh0, z(1, KP, M), z(1, M))]
nbytes = sum((a.shape[1] + b.shape[1]) * 2 * P for a, b, _w, _b in tail + front)


def per_layer():
    for a, b, w, bb in tail + front:
        o.wgrad(a, b, w, bb, n_splits=256)


def batched():
    o.wgrad_multi([(a, b, w, bb, None, None) for a, b, w, bb in tail])
    o.wgrad_multi([(a, b, w, bb, None, None) for a, b, w, bb in front])


for name, f in (("dense per-layer launches", per_layer), ("dense 2 batched launches", batched)):
    f(); f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        f()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 5
    print(f"{name}: {ms:.3f} ms, {nbytes / ms / 1e9:.2f} TB/s of operand reads")
