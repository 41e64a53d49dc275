"""Launched by tests/test_parallel_gpu.py under torchrun (two ranks sharing cuda:0, gloo) and by tests/test_rccl_gpu.py (ONE rank,
backend nccl = RCCL, loopback: `dp_overlap_worker.py nccl 50`): the data-parallel step with the backward
cut in two captured graphs and the all-reduce of the expert block issued on the side stream between them (graph.GraphedTrainStep,
split_backward = what DDP's gradient buckets do in the reference, runner.py:203-207) must train bit-identically to (a) the single-graph
step with one all-reduce behind the whole backward and (b) the eager step - three optimizer steps (argv[2]: more), every parameter and
Adam moment."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth  # noqa: E402
from switch_nerf_amd import parallel  # noqa: E402
from switch_nerf_amd.graph import GraphedTrainStep  # noqa: E402
from switch_nerf_amd.model import SwitchNeRF  # noqa: E402

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
backend = sys.argv[1] if len(sys.argv) > 1 else "gloo"
n_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
loop = world == 1           # one rank: the collectives are issued anyway (parallel.init_from_env(loopback=True))
parallel.init_from_env(backend, dev, loopback=loop)
assert torch.distributed.is_initialized() and torch.distributed.get_backend() == backend
allreduce = parallel.make_grad_allreduce(loopback=loop)
assert allreduce.active
N, S, chunk = 128, 64, 2048
batches = []
for it in range(3):
    rays, img, rgbs = synth.make_rays(800 + 10 * it + rank, N)         # every rank trains on its own rays
    batches.append(tuple(torch.from_numpy(x).to(dev) for x in (rays, img, rgbs)))
out = {}
flat0 = None
for mode in ("eager", "one_graph", "split"):
    m = SwitchNeRF(synth.BUILDING, dtype=torch.bfloat16, device=dev)
    m.load_state_dict(synth.make_weights(801, synth.BUILDING, gate_scale=1.0))
    if flat0 is None:
        flat0 = m.flat.clone()                                         # the initial parameters (the same in every mode)
    step = None
    if mode != "eager":         # (perturb / noise off: the three modes must see the same draws)
        step = GraphedTrainStep(m, batches[0][2], batches[0][0], batches[0][1], S, chunk, perturb=0.0, noise_std=0.0,
                                split_backward=(mode == "split"))
        assert step.split == (mode == "split") and (step.graph_b is not None) == (mode == "split")
    allreduce.profile = mode == "split"
    for it in range(n_steps):
        rays, img, rgbs = batches[it % 3]
        if step is None:
            m.train_step(rgbs, rays, img, S, chunk, perturb=0.0, grad_allreduce=allreduce)
        else:
            step(rgbs, rays, img, grad_allreduce=allreduce)
    torch.cuda.synchronize()
    rep = allreduce.report()
    if mode == "split":
        assert rep["collectives"] == 2 * n_steps, rep               # expert block + dense prefix per step
    out[mode] = (m.flat.clone(), m.m.clone(), m.v.clone(), m.step_count)
ok = True
for mode in ("one_graph", "split"):
    for a, b in zip(out["eager"][:3], out[mode][:3]):
        ok &= bool(torch.equal(a, b))
    ok &= out[mode][3] == out["eager"][3] == n_steps
moved = all(bool((out[mode][0] != flat0).any()) for mode in out)       # the optimizer steps changed the parameters in every mode
ok &= allreduce.stale_drains == 0
print(f"DP_OVERLAP rank {rank} ({backend}, {n_steps} steps): {'OK' if ok and moved else 'MISMATCH'}", flush=True)
torch.distributed.barrier()
torch.distributed.destroy_process_group()
sys.exit(0 if ok and moved else 1)
