"""The tail's backward layers in front of the expert backward chain (swn_chain_desc.head_layers, tag 8) against the two launches it
replaces (64-row tail backward chain with the combine backward in its write-out + expert backward chain on geometry 7).
python scripts/headfuse_check.py [small|full] [time]"""
import sys
import torch
sys.path.insert(0, '.')
from switch_nerf_amd import ops as o
dev, dt = torch.device('cuda'), torch.bfloat16
mode = sys.argv[1] if len(sys.argv) > 1 else "small"
timing = "time" in sys.argv
M, E, L, H2 = 256, 8, 7, 128
n_seg, seg_tokens = (2, 8192) if mode == "small" else (16, 131072)
P = n_seg * seg_tokens
cap = seg_tokens // E
torch.manual_seed(0)
Wm = [torch.randn(E, M, M, device=dev).mul_(1 / 16) for _ in range(L)]
Wb = [o.pack_weights(w, dt, False) for w in Wm]
W1 = torch.randn(1, M, M, device=dev).mul_(1 / 16)
W2 = torch.randn(1, M, H2, device=dev).mul_(1 / 16)
w1b, w2b, w2bpad = o.pack_weights(W1, dt, False), o.pack_weights(W2, dt, False), o.pack_weights_padded(W2, dt, False, 0, 256)
wsig = torch.randn(M, device=dev).mul_(0.1)
dsig = torch.randn(P, device=dev)
dh2 = torch.randn(P, H2, device=dev).to(dt)
y = torch.randn(P, M, device=dev).relu().to(dt)
probs = torch.tensor([3.0, 2.0, 1.0, 1.0, 1.0, 1.0, 0.5, 0.5], device=dev)
idx = torch.multinomial(probs, P, replacement=True).int()
gmax = torch.rand(P, device=dev) * 0.8 + 0.2
gates = torch.rand(P, E, device=dev)
loc, counts, perm, tok2row, _ = o.route_top1(idx, gmax, gates, seg_tokens, E, cap, True)
drop_begin, dropped = o.route_dropped(idx, loc, counts, seg_tokens, E, cap)
y[tok2row < 0] = 0          # (what the forward leaves for dropped tokens)
ng, rows = n_seg * E, n_seg * E * cap
masks = [torch.randint(-2 ** 31, 2 ** 31 - 1, (o.chain_mask_words(dt, ng, cap, M),), dtype=torch.int32, device=dev) for _ in range(L - 1)]
kw = dict(n_groups=ng, n_wsets=E, group_stride=cap, group_rows=counts.view(-1), group_rows_clamp=cap, x_gather=perm.view(-1))
print("tokens", P, "dropped", int(drop_begin[-1].item()), flush=True)


def expert_bwd(dz):
    return [o.Layer(Wb[l], None, relu=2 if l > 0 else 0, mask=masks[l - 1] if l > 0 else None, save=dz[l - 1] if l > 0 else None)
            for l in range(L - 1, -1, -1)]


def unfused():
    dh1 = torch.zeros(P, M, dtype=dt, device=dev); dout = torch.zeros(P, M, dtype=dt, device=dev)
    dgm = torch.zeros(P, device=dev)
    o.mlp_chain(dh2, [o.Layer(w2b, None, save=dh1), o.Layer(w1b, None)], dout, tag=5, combine=(y, dsig, wsig, gmax, dgm))
    dz = [torch.zeros(rows, M, dtype=dt, device=dev) for _ in range(L - 1)]
    dx = torch.zeros(rows, M, dtype=dt, device=dev)
    o.mlp_chain(dout, expert_bwd(dz), dx, y_add=dz[3], tag=2, geometry=7, **kw)
    return dh1, dout, dgm, dz, dx


def fused():
    dh1 = torch.full((P, M), 7.0, dtype=dt, device=dev); dzl = torch.zeros(rows, M, dtype=dt, device=dev)
    dgm = torch.full((P,), 7.0, device=dev)
    dz = [torch.zeros(rows, M, dtype=dt, device=dev) for _ in range(L - 1)]
    dx = torch.zeros(rows, M, dtype=dt, device=dev)
    o.mlp_chain(dh2, [o.Layer(w2bpad, None, save=dh1), o.Layer(w1b, None, save=dzl)] + expert_bwd(dz), dx, y_add=dz[3], tag=8, geometry=7,
                x_features=H2, combine=(y, dsig, wsig, gmax, dgm), head=(2, drop_begin, dropped), **kw)
    return dh1, dzl, dgm, dz, dx


a, b = unfused(), fused()
torch.cuda.synchronize()
print("dh1 identical:", torch.equal(a[0], b[0]), (a[0].float() - b[0].float()).abs().max().item())
pm = perm.view(-1).long()
valid = pm >= 0
ref_rows = a[1][pm.clamp(min=0)]
print("d_eo rows identical:", torch.equal(ref_rows[valid], b[1][valid]), (ref_rows[valid].float() - b[1][valid].float()).abs().max().item())
print("dgate max rel diff:", ((a[2] - b[2]).abs().max() / a[2].abs().max()).item(), " dropped zero:", float(b[2][tok2row < 0].abs().max()) if (tok2row < 0).any() else 0.0)
for l in range(L - 1):
    print(f"dz{l} identical:", torch.equal(a[3][l], b[3][l]))
print("dx identical:", torch.equal(a[4], b[4]))
# the sigma head's weight gradient from the same launch (comb_dwsig): += into a vector holding 1.0, twice (fixed order: identical bits)
dws = [torch.ones(M, device=dev) for _ in range(2)]
for t_ in dws:
    dgm2 = torch.zeros(P, device=dev); dz2 = [torch.zeros(rows, M, dtype=dt, device=dev) for _ in range(L - 1)]
    o.mlp_chain(dh2, [o.Layer(w2bpad, None, save=torch.zeros(P, M, dtype=dt, device=dev)), o.Layer(w1b, None, save=torch.zeros(rows, M, dtype=dt, device=dev))]
                + expert_bwd(dz2), torch.zeros(rows, M, dtype=dt, device=dev), y_add=dz2[3], tag=8, geometry=7, x_features=H2,
                combine=(y, dsig, wsig, gmax, dgm2, t_), head=(2, drop_begin, dropped), **kw)
kept = (tok2row >= 0)
ref = 1.0 + (dsig.double()[:, None] * y.double() * kept[:, None]).sum(0)
err = ((dws[0].double() - ref).abs().max() / ref.abs().max()).item()
print("dwsig rel err vs fp64:", err, " two launches identical:", torch.equal(dws[0], dws[1]), " dgate identical to the launch without it:", torch.equal(dgm2, b[2]))
assert err < 1e-5 and torch.equal(dws[0], dws[1]) and torch.equal(dgm2, b[2])
if timing:
    print("with comb_dwsig:", end=" ")
if timing:
    def bench(fn, n=5):
        fn(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n): fn()
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / n)
        return best
    dh1 = torch.zeros(P, M, dtype=dt, device=dev); dout = torch.zeros(P, M, dtype=dt, device=dev); dzl = torch.zeros(rows, M, dtype=dt, device=dev)
    dgm = torch.zeros(P, device=dev)
    dz = [torch.zeros(rows, M, dtype=dt, device=dev) for _ in range(L - 1)]
    dx = torch.zeros(rows, M, dtype=dt, device=dev)
    t_tail = bench(lambda: o.mlp_chain(dh2, [o.Layer(w2b, None, save=dh1), o.Layer(w1b, None)], dout, tag=5, combine=(y, dsig, wsig, gmax, dgm)))
    t_exp = bench(lambda: o.mlp_chain(dout, expert_bwd(dz), dx, y_add=dz[3], tag=2, geometry=7, **kw))
    t_fus = bench(lambda: o.mlp_chain(dh2, [o.Layer(w2bpad, None, save=dh1), o.Layer(w1b, None, save=dzl)] + expert_bwd(dz), dx, y_add=dz[3], tag=8,
                                      geometry=7, x_features=H2, combine=(y, dsig, wsig, gmax, dgm), head=(2, drop_begin, dropped), **kw))
    dwt = torch.zeros(M, device=dev)
    t_fus2 = bench(lambda: o.mlp_chain(dh2, [o.Layer(w2bpad, None, save=dh1), o.Layer(w1b, None, save=dzl)] + expert_bwd(dz), dx, y_add=dz[3], tag=8,
                                       geometry=7, x_features=H2, combine=(y, dsig, wsig, gmax, dgm, dwt), head=(2, drop_begin, dropped), **kw))
    print(f"tail backward {t_tail:.3f} ms + expert backward {t_exp:.3f} ms = {t_tail + t_exp:.3f};  fused {t_fus:.3f} ms;  fused + comb_dwsig (fill, run sums, reduce included) {t_fus2:.3f} ms")
