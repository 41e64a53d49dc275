"""Test-only stub: single-process semantics of tutel.impls.communicate."""
import torch

TUTEL_GROUPING_CACHE = {}


def get_world_size(group=None):
    return 1


def get_world_rank(group=None):
    return 0


def all_to_all_single(x, group=None, **kw):
    return x


def simple_all_reduce(x, group=None, op=None, **kw):
    return x


class _Env:
    data_group = None
    model_group = None
    global_group = None
    global_size = 1
    global_rank = 0
    local_size = 1
    local_rank = 0
    local_device = torch.device("cpu")
    is_distributed = False
    mode = "single"


def create_groups_from_world(group_count, include_init=None):
    return _Env()


class AllToAllStatus:
    @staticmethod
    def init(*a, **k):
        pass


class PrimAllgather(torch.autograd.Function):
    @staticmethod
    def forward(ctx, group, x, fused=False):
        return x

    @staticmethod
    def backward(ctx, g):
        return None, g, None


def zero_gather(x, group=None):
    return x
