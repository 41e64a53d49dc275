"""Launched by tests/test_parallel_gpu.py under torchrun (two ranks sharing cuda:0, gloo): the evaluation forward without token
dropping with the experts sharded over the ranks (unequal-split exchange) must equal the single-rank packed forward."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth  # noqa: E402
from switch_nerf_amd import parallel  # noqa: E402
from switch_nerf_amd.model import SwitchNeRF  # noqa: E402

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
parallel.init_from_env("gloo", dev)
dtype = torch.bfloat16 if sys.argv[1] == "bf16" else torch.float32
m = SwitchNeRF(synth.BUILDING, dtype=dtype, device=dev, batch_prioritized=False)
m.load_state_dict(synth.make_weights(91, synth.BUILDING, gate_scale=1.0))
N, S, chunk = 32, 128, 2048
rays, img, _ = synth.make_rays(92 + rank, N)             # every rank renders its own rays
rays, img = torch.from_numpy(rays).to(dev), torch.from_numpy(img).to(dev)
ref = m.forward_rays(rays, img, S, chunk, training=False, no_batch=True)["raw"].clone()
m.set_expert_parallel(parallel.ExpertParallel(rank, world, m.E))
got = m.forward_rays(rays, img, S, chunk, training=False, no_batch=True)["raw"]
ok = torch.equal(got, ref) and bool(torch.isfinite(got).all()) and float(got.abs().sum()) > 0
print(f"EP_EVAL rank {rank}: {'OK' if ok else 'MISMATCH'} max diff {(got - ref).abs().max().item():.3e}", flush=True)
torch.distributed.barrier()
torch.distributed.destroy_process_group()
sys.exit(0 if ok else 1)
