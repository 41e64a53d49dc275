"""Test-only stub: tutel.jit_kernels.sparse (batched, capacity-padded kernels)."""
from ..impls import jit_compiler as J


def create_forward(param_dtype, is_cuda=True):
    def f(g, i, l, x, d, extra):
        J._fwd(g, i, l, x, d, extra)
    return f


def create_backward_data(param_dtype, is_cuda=True):
    def f(g, i, l, x, d, extra):
        J._bwd_data(g, i, l, x, d, extra)
    return f


def create_backward_gate(param_dtype, is_cuda=True):
    def f(gg, i, l, x, d, extra):
        J._bwd_gate(gg, i, l, x, d, extra)
    return f
