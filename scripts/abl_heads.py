"""Fused forward launch (tag 7) at full size with and without the fused heads, outputs allocated once (scripts/tailfuse_check.py's
setup).  Usage: python scripts/abl_heads.py full"""
import sys, torch
sys.path.insert(0, '.')
exec(open('scripts/tailfuse_check.py').read().split("ya, h1a, h2a, rawa = unfused()")[0])
y = torch.empty(P, M, dtype=dt, device=dev); h1 = torch.empty(P, M, dtype=dt, device=dev); h2 = torch.empty(P, H2, dtype=dt, device=dev)
raw = torch.empty(P, 4, device=dev)
def run(hd, save=True):
    lys = expert_layers(save); lys[-1].save = y if save else None
    lys += [o.Layer(w1p, b1, save=h1 if save else None), o.Layer(w2pad, None, relu=1, rowbias=c_ray, rows_per_bias=S)]
    o.mlp_chain(h0, lys, h2 if save else None, tag=7, geometry=7, heads=heads(raw) if hd else None, tail=(L, gmax, drop_begin, dropped, H2), **kw)
def bench(fn, n=5):
    fn(); torch.cuda.synchronize(); best = 1e9
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n): fn()
        b.record(); torch.cuda.synchronize(); best = min(best, a.elapsed_time(b) / n)
    return best
for r in range(2):
    print(f"fused train: with heads {bench(lambda: run(True)):.3f} ms, without {bench(lambda: run(False)):.3f} ms;  "
          f"inference (no saves): with heads {bench(lambda: run(True, False)):.3f} ms")
