"""Phase timers of the front forward chain on geometry 7 (timing variant of the library, see scripts/chainq_phases.py)."""
import sys
import torch
sys.path.insert(0, '.')
from switch_nerf_amd import ops as o
dev, dt = torch.device('cuda'), torch.bfloat16
P = 2097152
torch.manual_seed(0)
pe = torch.randn(P, 128, device=dev).to(dt)
W0 = torch.randn(1, 128, 256, device=dev) / 11
W1 = torch.randn(1, 256, 256, device=dev) / 16
W2 = torch.randn(1, 256, 256, device=dev) / 16
b = [torch.randn(1, 256, device=dev) * 0.1 for _ in range(3)]
w0p, w1, w2 = o.pack_weights_padded(W0, dt, True, 256), o.pack_weights(W1, dt, True), o.pack_weights(W2, dt, True)
h0, a1, g = (torch.empty(P, 256, dtype=dt, device=dev) for _ in range(3))
mask = torch.zeros(o.chain_mask_words(dt, 1, P, 256), dtype=torch.int32, device=dev)
dbg = torch.zeros(4096 * 8, dtype=torch.int64, device=dev)
L = 3
for mode in ("train", "inference"):
    sv = mode == "train"
    layers = [o.Layer(w0p, b[0], save=h0), o.Layer(w1, b[1], relu=1, mask=mask if sv else None, save=a1 if sv else None), o.Layer(w2, b[2])]
    def f():
        o.mlp_chain(pe, layers, g, tag=3, geometry=7, x_features=128, y_add_gather=dbg.view(torch.int32))
    f(); torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); f(); e.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(e)
    for nm, sl in (("wave 0 (row group 0)", slice(0, 256)), ("wave 4 (row group 1)", slice(2048, 2048 + 256))):
        raw = dbg.view(4096, 8)[sl]
        t = raw.double().mean(0).tolist()
        tiles = (raw[:, 6] & 0xFFFF).double().mean().item()
        t_si = (raw[:, 6] >> 16).double().mean().item()
        clk = t[7] / (ms * 1e-3) / 1e9
        print(f"{mode} {nm}: {ms:.3f} ms, {tiles:.1f} tiles per workgroup, implied clock {clk:.2f} GHz; per tile: S {t[0] / tiles:.0f} (write-out issued at "
              f"{t[1] / tiles:.0f}, staging + claim issued at {t_si / tiles:.0f}); per layer: K {t[2] / tiles / L:.0f} + barrier {t[3] / tiles / L:.0f}, "
              f"E {t[4] / tiles / L:.0f} + barriers (E and S) {t[5] / tiles / L:.0f}; total per tile {t[7] / tiles:.0f}")
