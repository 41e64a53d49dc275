"""Launched by tests/test_parallel_gpu.py under torchrun (two ranks sharing cuda:0, gloo): two optimizer steps with the experts SHARDED
over the ranks (kept rows travel, a rank updates only the experts it owns) followed by checkpoint.save_checkpoint on EVERY rank must
give the same checkpoint - parameters and Adam moments of all experts - as the same two steps in data-parallel mode (experts
replicated, whole gradient all-reduced).  ADVICE round 2: gather_expert_shards had never been called by anything."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth  # noqa: E402
from switch_nerf_amd import checkpoint, parallel  # noqa: E402
from switch_nerf_amd.model import SwitchNeRF  # noqa: E402

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
parallel.init_from_env("gloo", dev)
allreduce = parallel.make_grad_allreduce()
N, S, chunk = 64, 64, 1024
batches = []
for it in range(2):
    rays, img, rgbs = synth.make_rays(700 + 10 * it + rank, N)         # every rank trains on its own rays
    batches.append(tuple(torch.from_numpy(x).to(dev) for x in (rays, img, rgbs)))
cks = {}
for mode in ("dp", "ep"):
    m = SwitchNeRF(synth.BUILDING, dtype=torch.float32, device=dev)
    m.load_state_dict(synth.make_weights(701, synth.BUILDING, gate_scale=1.0))
    if mode == "ep":
        m.set_expert_parallel(parallel.ExpertParallel(rank, world, m.E))
    for rays, img, rgbs in batches:
        m.train_step(rgbs, rays, img, S, chunk, perturb=0.0, grad_allreduce=allreduce)
    cks[mode] = checkpoint.save_checkpoint(None, m, iteration=2)       # collective under EP: gathers every expert from its owner
ok = True
worst = 0.0
for k, v in cks["dp"]["model_state_dict"].items():
    d = (v - cks["ep"]["model_state_dict"][k]).abs().max().item() / (v.abs().max().item() + 1e-12)
    worst = max(worst, d)
    ok &= d <= 5e-5
sd, se = cks["dp"]["optimizers"]["nerf"]["state"], cks["ep"]["optimizers"]["nerf"]["state"]
for i in sd:
    for key in ("exp_avg", "exp_avg_sq"):
        d = (sd[i][key] - se[i][key]).abs().max().item() / (sd[i][key].abs().max().item() + 1e-20)
        worst = max(worst, d)
        ok &= d <= 5e-4
moved = (cks["ep"]["model_state_dict"]["module.layers.0.experts.0.weights.3"] - torch.from_numpy(
    synth.make_weights(701, synth.BUILDING, gate_scale=1.0)["layers.0.experts.0.weights.3"])).abs().amax(dim=(1, 2))
ok &= bool((moved > 0).all())                                          # every expert - also the other rank's - carries trained weights
print(f"EP_CKPT rank {rank}: {'OK' if ok else 'MISMATCH'} worst relative difference {worst:.3e}", flush=True)
torch.distributed.barrier()
torch.distributed.destroy_process_group()
sys.exit(0 if ok else 1)
