"""Test-only stub: tutel.jit_kernels.gating."""
import torch


def torch_cumsum_sub_one(mask1):
    return torch.cumsum(mask1, dim=0) - 1


def fast_cumsum_sub_one(data, dim=0):
    return torch.cumsum(data, dim=dim) - 1
