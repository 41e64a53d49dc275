"""The arithmetic of the matrix-pipe router kernels (switch_nerf_amd/csrc/gate_mfma.hip) restated with torch on the CPU: the LayerNorm
folded into the router weights and the fp32 weights split into bf16 terms reproduce the reference's LayerNorm -> fp32 router ->
softmax (models/nerf_moe.py:370-372, tutel_moe_layer_nobatch.py:105-126) to fp32 rounding.  No GPU, no library: this pins the
algorithm the kernels implement, the GPU tests pin the kernels."""
import torch


def _split_bf16(w, terms):
    parts, r = [], w.clone()
    for _ in range(terms):
        p = r.to(torch.bfloat16).float()
        parts.append(p)
        r = r - p
    return parts


def test_folded_layernorm_router_with_split_weights_matches_fp64():
    g = torch.Generator().manual_seed(0)
    P, G, E = 4096, 256, 8
    for shift in (0.0, 0.7, 3.0):
        x = (torch.randn(P, G, generator=g) * 1.3 + shift).to(torch.bfloat16)          # the gate input is bf16: exact in the MFMA
        ln_w = 1.0 + 0.2 * torch.randn(G, generator=g)
        ln_b = 0.1 * torch.randn(G, generator=g)
        wg = torch.randn(E, G, generator=g) * 0.3
        ref = torch.softmax(torch.nn.functional.layer_norm(x.double(), (G,), ln_w.double(), ln_b.double(), 1e-5) @ wg.double().t(), 1)
        # the kernel's algebra: W' = ln_w (.) wg split into hi + mid + lo (bf16); products of bf16 numbers are exact in fp32
        wp = wg * ln_w
        hi, mid, lo = _split_bf16(wp, 3)
        assert ((hi + mid + lo) - wp).abs().max() <= 2e-7 * wp.abs().max()              # 24 mantissa bits
        xf = x.float()
        s1 = (xf @ hi.t() + xf @ mid.t()) + xf @ lo.t()                                  # fp32 accumulation, three partial sums
        c1 = (hi + (mid + lo)).sum(1)                                                    # the sum of what was multiplied
        c0 = (wg * ln_b).sum(1)
        mean = xf.sum(1) / G                                                             # the row of ones in the same MFMA
        var = ((xf * xf).sum(1) / G - mean * mean).clamp(min=0)                          # E[x^2] - mean^2
        rstd = 1.0 / torch.sqrt(var + 1e-5)
        logits = rstd[:, None] * (s1 - mean[:, None] * c1[None]) + c0[None]
        pr = torch.softmax(logits, 1)
        assert (pr.double() - ref).abs().max().item() <= 2e-6
        top2 = torch.topk(ref, 2, dim=1).values
        clear = (top2[:, 0] - top2[:, 1]) > 1e-5
        assert torch.equal(pr.argmax(1)[clear], ref.argmax(1)[clear])


def test_router_backward_identities():
    """d_wg, d_ln_w and d_ln_b from M = dlogits^T xhat and DL = sum dlogits (gate_dwg_finalize_kernel) equal autograd."""
    g = torch.Generator().manual_seed(1)
    P, G, E = 2048, 256, 8
    x = (torch.randn(P, G, generator=g) * 1.3 + 0.3).double()
    ln_w = (1.0 + 0.2 * torch.randn(G, generator=g)).double().requires_grad_(True)
    ln_b = (0.1 * torch.randn(G, generator=g)).double().requires_grad_(True)
    wg = (torch.randn(E, G, generator=g) * 0.3).double().requires_grad_(True)
    dl = torch.randn(P, E, generator=g).double()                                          # any upstream gradient of the logits
    logits = torch.nn.functional.layer_norm(x, (G,), ln_w, ln_b, 1e-5) @ wg.t()
    (logits * dl).sum().backward()
    mean = x.mean(1, keepdim=True)
    xhat = (x - mean) / torch.sqrt(x.var(1, unbiased=False, keepdim=True) + 1e-5)
    M = dl.t() @ xhat                                                                     # [E, G]: one GEMM over the tokens
    DL = dl.sum(0)                                                                        # [E]
    d_wg = ln_w.detach()[None] * M + ln_b.detach()[None] * DL[:, None]
    d_ln_w = (wg.detach() * M).sum(0)
    d_ln_b = (wg.detach() * DL[:, None]).sum(0)
    for a, b in ((d_wg, wg.grad), (d_ln_w, ln_w.grad), (d_ln_b, ln_b.grad)):
        assert (a - b).abs().max().item() <= 1e-9 * max(1.0, b.abs().max().item())
