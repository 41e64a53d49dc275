"""Router kernels against a torch fp64 reference + timing.  python scripts/gate_check.py   (SWN_GATE_VALU=1: the VALU kernels)"""
import sys, os, torch
sys.path.insert(0, '.')
from switch_nerf_amd import ops as o, _lib
dev = torch.device('cuda')
HALF = torch.float16 if os.environ.get('GATE_DTYPE') == 'fp16' else torch.bfloat16
if HALF == torch.float16:
    _lib.use_half('f16')
torch.manual_seed(0)
G, E = int(os.environ.get('GATE_G', 256)), int(os.environ.get('GATE_E', 8))      # GATE_G=512 GATE_E=16: Mission Bay's router
which = ("VALU" if os.environ.get("SWN_GATE_VALU") else "MFMA") + (" fp16" if os.environ.get("GATE_DTYPE") == "fp16" else "")


def ref(g, ln_w, ln_b, wg):
    x = g.double()
    if ln_w is not None:
        x = torch.nn.functional.layer_norm(x, (G,), ln_w.double(), ln_b.double(), 1e-5)
    logits = x @ wg.double().t()
    return torch.softmax(logits, 1)


for P, mean_shift, E_ in ((5000, 0.0, E), (32768, 0.7, E), (33, 0.0, E), (4097, 3.0, E - 3), (131072, 0.2, E)):
    g = (torch.randn(P, G, device=dev) * 1.3 + mean_shift).to(HALF)
    ln_w = 1.0 + 0.2 * torch.randn(G, device=dev)
    ln_b = 0.1 * torch.randn(G, device=dev)
    wg = torch.randn(E_, G, device=dev) * 0.3
    for ln in (True, False):
        gates, idx, gmax, stats = o.gate_fwd(g, ln_w if ln else None, ln_b if ln else None, wg)
        torch.cuda.synchronize()
        pr = ref(g, ln_w if ln else None, ln_b if ln else None, wg)
        err = (gates.double() - pr).abs().max().item()
        ridx = pr.argmax(1)
        mis = (idx.long() != ridx)
        top2 = torch.topk(pr, 2, dim=1).values
        gap = (top2[:, 0] - top2[:, 1])[mis]
        gm_err = (gmax.double() - gates.double().gather(1, idx.long()[:, None])[:, 0]).abs().max().item()
        st_err = 0.0
        if ln:
            mu = g.double().mean(1); var = g.double().var(1, unbiased=False)
            st_err = max((stats[:, 0].double() - mu).abs().max().item(), ((stats[:, 1].double() - 1 / torch.sqrt(var + 1e-5)) * torch.sqrt(var + 1e-5)).abs().max().item())
        print(f"{which} P {P:7d} E {E_} shift {mean_shift} ln {int(ln)}: max |prob err| {err:.2e}, idx mismatches {int(mis.sum())} (largest top-2 gap among them "
              f"{gap.max().item() if mis.any() else 0:.1e}), gmax consistency {gm_err:.1e}, stats err {st_err:.1e}, sums {(gates.sum(1) - 1).abs().max().item():.1e}")

P = 2097152 * 256 // G
g = (torch.randn(P, G, device=dev) * 1.3).to(HALF)
ln_w = 1.0 + 0.2 * torch.randn(G, device=dev); ln_b = 0.1 * torch.randn(G, device=dev); wg = torch.randn(E, G, device=dev) * 0.3
for _ in range(2):
    o.gate_fwd(g, ln_w, ln_b, wg)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10):
    o.gate_fwd(g, ln_w, ln_b, wg)
b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b) / 10
print(f"{which} gate_fwd {P} tokens x {G}: {ms:.3f} ms = {P * G * 2 / ms / 1e9:.2f} TB/s of row reads")

# ---------------------------------------------------------------- backward: against torch autograd in fp64
print("backward")
for P, seg, E_, ln in ((8192, 4096, E, True), (8192, 8192, E, False), (4099 * 2, 4099, E // 2, True), (65536, 16384, E, True)):
    g = (torch.randn(P, G, device=dev) * 1.3 + 0.3).to(HALF)
    ln_w = (1.0 + 0.2 * torch.randn(G, device=dev)) if ln else None
    ln_b = (0.1 * torch.randn(G, device=dev)) if ln else None
    wg = torch.randn(E_, G, device=dev) * 0.3
    gates, idx, gmax, stats = o.gate_fwd(g, ln_w, ln_b, wg)
    n_seg = P // seg
    counts = torch.randint(0, seg, (n_seg, E_), device=dev, dtype=torch.int32)
    coef = torch.rand(n_seg, device=dev) * 1e-4
    dgmax = torch.randn(P, device=dev)
    d_wg = torch.zeros(E_, G, device=dev); d_lw = torch.zeros(G, device=dev); d_lb = torch.zeros(G, device=dev)
    dg = o.gate_bwd(g, ln_w, ln_b, wg, gates, idx, dgmax, stats, counts, coef, seg, d_wg, d_lw if ln else None, d_lb if ln else None)
    torch.cuda.synchronize()
    x = g.double().requires_grad_(True)
    W = wg.double().requires_grad_(True)
    lw = ln_w.double().requires_grad_(True) if ln else None
    lb = ln_b.double().requires_grad_(True) if ln else None
    xn = torch.nn.functional.layer_norm(x, (G,), lw, lb, 1e-5) if ln else x
    pr = torch.softmax(xn @ W.t(), 1)
    dp = coef.double().repeat_interleave(seg)[:, None] * counts.double().repeat_interleave(seg, 0)
    dp = dp + torch.nn.functional.one_hot(idx.long(), E_).double() * dgmax.double()[:, None]
    (pr * dp).sum().backward()
    def rel(a, b):
        return ((a.double() - b).abs().max() / b.abs().max()).item()
    line = f"{which} P {P} seg {seg} E {E_} ln {int(ln)}: dg rel err {rel(dg, x.grad):.2e}, d_wg {rel(d_wg, W.grad):.2e}"
    if ln:
        line += f", d_ln_w {rel(d_lw, lw.grad):.2e}, d_ln_b {rel(d_lb, lb.grad):.2e}"
    print(line)

P, seg = 2097152 * 256 // G, 131072
g = (torch.randn(P, G, device=dev) * 1.3).to(HALF)
gates, idx, gmax, stats = o.gate_fwd(g, ln_w, ln_b, wg) if False else o.gate_fwd(g, 1.0 + 0.2 * torch.randn(G, device=dev), 0.1 * torch.randn(G, device=dev), torch.randn(E, G, device=dev) * 0.3)
ln_w = 1.0 + 0.2 * torch.randn(G, device=dev); ln_b = 0.1 * torch.randn(G, device=dev); wg = torch.randn(E, G, device=dev) * 0.3
counts = torch.randint(0, seg, (P // seg, E), device=dev, dtype=torch.int32); coef = torch.rand(P // seg, device=dev) * 1e-4; dgmax = torch.randn(P, device=dev)
d_wg = torch.zeros(E, G, device=dev); d_lw = torch.zeros(G, device=dev); d_lb = torch.zeros(G, device=dev)
f = lambda: o.gate_bwd(g, ln_w, ln_b, wg, gates, idx, dgmax, stats, counts, coef, seg, d_wg, d_lw, d_lb)
f(); f(); torch.cuda.synchronize()
a.record()
for _ in range(10):
    f()
b.record(); torch.cuda.synchronize()
print(f"{which} gate_bwd (data path + parameter gradients) {P} tokens x {G}: {a.elapsed_time(b) / 10:.3f} ms")
