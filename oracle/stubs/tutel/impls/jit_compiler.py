"""Test-only stub: CPU restatement of the three Tutel sparse kernels (batched, capacity-padded ABI)."""
import torch

IS_HIP_EXTENSION = False


def _split(extra):
    samples, hidden, capacity = int(extra[0]), int(extra[1]), int(extra[2])
    return samples, hidden, capacity


def _g(gates, n):
    g = gates[:n]
    if g.dim() == 2:          # the "ones_helper" / fp16 [S,2] form: both columns equal
        g = g[:, 0]
    return g


def _rows(idx, loc, capacity, begin=None):
    idx = idx.long()
    loc = loc.long()
    if begin is None:
        keep = (idx >= 0) & (loc < capacity) & (loc >= 0)
        rows = idx.clamp(min=0) * capacity + loc
    else:
        keep = idx >= 0
        rows = begin.long()[idx.clamp(min=0)] + loc
    return keep, rows


def _fwd(gates, idx, loc, x, disp, extra, begin=None):
    n, h, cap = _split(extra)
    keep, rows = _rows(idx[:n], loc[:n], cap, begin)
    g = _g(gates, n).to(x.dtype)
    src = (g.unsqueeze(1) * x[:n].reshape(n, -1))[keep]
    disp.view(-1, src.shape[1]).index_add_(0, rows[keep], src)


def _bwd_data(gates, idx, loc, out, disp, extra, begin=None):
    n, h, cap = _split(extra)
    keep, rows = _rows(idx[:n], loc[:n], cap, begin)
    g = _g(gates, n).to(disp.dtype)
    d2 = disp.view(-1, out.shape[-1])
    o = out.view(-1, out.shape[-1])
    o[:n] = 0
    o[:n][keep] = g[keep].unsqueeze(1) * d2[rows[keep]]


def _bwd_gate(grad_gates, idx, loc, x, disp, extra, begin=None):
    n, h, cap = _split(extra)
    keep, rows = _rows(idx[:n], loc[:n], cap, begin)
    d2 = disp.view(-1, x.shape[-1])
    grad_gates[:n] = 0
    grad_gates[:n][keep] = (d2[rows[keep]] * x.view(-1, x.shape[-1])[:n][keep]).sum(1)


class JitCompiler:
    @staticmethod
    def generate_cpu_kernel(kernel_type):
        # nobatch ABI: (g, idx, loc, begin, x, disp, extra=[...])
        def fwd(g, i, l, b, x, d, extra):
            _fwd(g, i, l, x, d, extra, begin=b)

        def bwd_data(g, i, l, b, x, d, extra):
            _bwd_data(g, i, l, x, d, extra, begin=b)

        def bwd_gate(gg, i, l, b, x, d, extra):
            _bwd_gate(gg, i, l, x, d, extra, begin=b)
        return [fwd, bwd_data, bwd_gate][kernel_type]

    @staticmethod
    def generate_kernel(*a, **k):
        raise RuntimeError("GPU JIT is not available in the oracle stub")
