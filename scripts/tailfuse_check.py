"""The dense tail folded into the expert forward chain (swn_chain_desc.tail_first, tag 7) against the two launches it replaces
(expert chain on geometry 7 + the 64-row tail chain with fused heads).  python scripts/tailfuse_check.py [small|full] [time]"""
import sys
import torch
sys.path.insert(0, '.')
from switch_nerf_amd import ops as o
dev, dt = torch.device('cuda'), torch.bfloat16
mode = sys.argv[1] if len(sys.argv) > 1 else "small"
timing = "time" in sys.argv
M, E, L, H2, S = 256, 8, 7, 128, 64
n_seg, seg_tokens = (2, 8192) if mode == "small" else (16, 131072)
if mode == "full":
    S = 256
P = n_seg * seg_tokens
cap = seg_tokens // E
torch.manual_seed(0)
Wm = [torch.randn(E, M, M, device=dev).mul_(1 / 16) for _ in range(L)]
Wf = [o.pack_weights(w, dt, True) for w in Wm]
B = [torch.randn(E, M, device=dev).mul_(0.1) for _ in range(L)]
W1 = torch.randn(1, M, M, device=dev).mul_(1 / 16); b1 = torch.randn(1, M, device=dev).mul_(0.1)
W2 = torch.randn(1, M, H2, device=dev).mul_(1 / 16)
w1p = o.pack_weights(W1, dt, True)
w2p = o.pack_weights(W2, dt, True)
w2pad = o.pack_weights_padded(W2, dt, True, 0, 256)
ws = torch.randn(M, device=dev).mul_(0.1); bs = torch.randn(1, device=dev)
wc = torch.randn(3, H2, device=dev).mul_(0.1); bc = torch.randn(3, device=dev)
c_ray = torch.randn(P // S, H2, device=dev)
noise = torch.randn(P, device=dev)
h0 = torch.randn(P, M, device=dev).to(dt)
# routing: skewed so that tokens are dropped
probs = torch.tensor([3.0, 2.0, 1.0, 1.0, 1.0, 1.0, 0.5, 0.5], device=dev)
idx = torch.multinomial(probs, P, replacement=True).int()
gmax = torch.rand(P, device=dev) * 0.8 + 0.2
gates = torch.rand(P, E, device=dev)
loc, counts, perm, tok2row, _ = o.route_top1(idx, gmax, gates, seg_tokens, E, cap, True)
drop_begin, dropped = o.route_dropped(idx, loc, counts, seg_tokens, E, cap)
nd = int(drop_begin[-1].item())
ref_drop = (tok2row < 0).nonzero()[:, 0]
assert nd == ref_drop.numel(), (nd, ref_drop.numel())
assert torch.equal(torch.sort(dropped[:nd].long())[0], ref_drop), "dropped list"
print("tokens", P, "dropped", nd, flush=True)
ng, rows = n_seg * E, n_seg * E * cap
sv = True
saves = [torch.zeros(rows, M, dtype=dt, device=dev) for _ in range(L - 1)]
masks = [torch.zeros(o.chain_mask_words(dt, ng, cap, M), dtype=torch.int32, device=dev) for _ in range(L - 1)]


def expert_layers(save):
    return [o.Layer(Wf[l], B[l], relu=1 if l < L - 1 else 0, skip=(l == 3), save=saves[l] if (save and l < L - 1) else None,
                    mask=masks[l] if (save and l < L - 1) else None) for l in range(L)]


kw = dict(n_groups=ng, n_wsets=E, group_stride=cap, group_rows=counts.view(-1), group_rows_clamp=cap, x_gather=perm.view(-1))
heads = lambda raw: (ws, bs, wc, bc, noise, raw)


def unfused(save=True):
    eo = torch.zeros(rows, M, dtype=dt, device=dev)
    o.mlp_chain(h0, expert_layers(save), eo, tag=1, geometry=7, **kw)
    y = torch.zeros(P, M, dtype=dt, device=dev); h1 = torch.zeros(P, M, dtype=dt, device=dev); h2 = torch.zeros(P, H2, dtype=dt, device=dev)
    raw = torch.zeros(P, 4, device=dev)
    o.mlp_chain(eo, [o.Layer(w1p, b1, save=h1 if save else None), o.Layer(w2p, None, relu=1, rowbias=c_ray, rows_per_bias=S)], h2 if save else None,
                group_stride=P, x_gather=tok2row, x_save=y if save else None, x_scale=gmax, x_relu=True, tag=4, heads=heads(raw))
    return y, h1, h2, raw


def fused(save=True):
    y = torch.full((P, M), 7.0, dtype=dt, device=dev); h1 = torch.full((P, M), 7.0, dtype=dt, device=dev)
    h2 = torch.full((P, H2), 7.0, dtype=dt, device=dev)
    raw = torch.full((P, 4), 7.0, device=dev)
    lys = expert_layers(save)
    lys[-1].save = y if save else None
    lys += [o.Layer(w1p, b1, save=h1 if save else None), o.Layer(w2pad, None, relu=1, rowbias=c_ray, rows_per_bias=S)]
    o.mlp_chain(h0, lys, h2 if save else None, tag=7, geometry=7, heads=heads(raw), tail=(L, gmax, drop_begin, dropped, H2), **kw)
    return y, h1, h2, raw


ya, h1a, h2a, rawa = unfused()
sa = [s.clone() for s in saves]
ma = [m.clone() for m in masks]
for s in saves: s.zero_()
for m in masks: m.zero_()
yb, h1b, h2b, rawb = fused()
torch.cuda.synchronize()
for l in range(L - 1):
    assert torch.equal(sa[l], saves[l]), f"save {l}"
    assert torch.equal(ma[l], masks[l]), f"mask {l}"
print("expert saves / masks identical")
print("y identical:", torch.equal(ya, yb), " max diff", (ya.float() - yb.float()).abs().max().item())
for name, a, b in (("h1", h1a, h1b), ("h2", h2a, h2b)):
    dif = (a.float() - b.float()).abs()
    print(f"{name}: max diff {dif.max().item():.4g}, differing {float((dif > 0).float().mean()):.4%}, max |ref| {a.float().abs().max().item():.3g}")
print("raw max diff", (rawa - rawb).abs().max().item(), " sigma", (rawa[:, 3] - rawb[:, 3]).abs().max().item())
bad = ((rawa - rawb).abs().amax(1) > 0.05).nonzero()[:, 0]
if bad.numel():
    print("bad rows", bad[:10].tolist(), "dropped?", (tok2row[bad[:10]] < 0).tolist())
# inference form: nothing but raw
_, _, _, rawc = fused(save=False)
print("inference raw == training raw:", torch.equal(rawb, rawc))
if timing:
    def bench(fn, n=5):
        fn(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(n): fn()
            b.record(); torch.cuda.synchronize()
            best = min(best, a.elapsed_time(b) / n)
        return best
    print(f"unfused train {bench(unfused):.3f} ms, fused train {bench(fused):.3f} ms;  unfused eval {bench(lambda: unfused(False)):.3f}, "
          f"fused eval {bench(lambda: fused(False)):.3f}")
