"""Sanity: RCCL process group with one rank on the GPU box (init, all_reduce, barrier) - the N>1 bench path's API calls."""
import os, sys, torch
os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29517")
import torch.distributed as dist
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
t = torch.ones(4_000_000, device="cuda")
dist.all_reduce(t); dist.barrier(); torch.cuda.synchronize()
print("rccl single-rank ok", t[:2].tolist())
dist.destroy_process_group()
