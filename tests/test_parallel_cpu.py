"""World-size-2 gloo test of the data-parallel host logic (ray sharding + flat-gradient all-reduce), CPU only."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from switch_nerf_amd import parallel
    r, w = parallel.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    b, e = parallel.shard_rays(8192, rank, world)
    flat = torch.full((1000,), float(rank + 1))
    flat[rank] = 100.0
    scale = parallel.make_grad_allreduce()(flat)
    out[rank] = (b, e, scale, flat[:3].tolist(), flat[5].item())
    dist.destroy_process_group()


def test_dp_allreduce_and_sharding_gloo():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, 29531 + os.getpid() % 200, out), nprocs=world, join=True)
    assert out[0][:2] == (0, 4096) and out[1][:2] == (4096, 8192)
    for r in range(world):
        b, e, scale, head, mid = out[r]
        assert scale == 0.5
        assert head == [102.0, 101.0, 3.0] and mid == 3.0     # identical summed buffer on both ranks
