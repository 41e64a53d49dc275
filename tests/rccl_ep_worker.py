"""Launched by tests/test_rccl_gpu.py: ONE rank, backend nccl (= RCCL), loopback (parallel.init_from_env(loopback=True)).  Every
collective of the expert-parallel step (tutel_moe_layer_nobatch.py:157-185 - the all-to-all in front of and behind the experts - and
the dense-prefix all-reduce) is issued for real on a world-1 RCCL communicator: same API calls, side stream, split sizes and graph
capture as at W > 1; the payload moves inside the GPU.
  (a) primitives: unequal-split all_to_all_single on the side stream, the equal-split form, the count exchange, the ragged evaluation
      exchange, GradAllReduce begin / finish around work on the launch stream;
  (b) the kept-rows step (unequal splits, eager) == the same step with no process group (buffers aliased, nothing moves), bit for bit;
  (c) the capacity-padded step CAPTURED into a hipGraph with the RCCL calls inside and replayed argv[1] (50) times == the eager padded
      step, bit for bit (every parameter and Adam moment)."""
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

n_replays = int(sys.argv[1]) if len(sys.argv) > 1 else 50
os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
parallel.init_from_env("nccl", dev, loopback=True)
import torch.distributed as dist  # noqa: E402
assert dist.is_initialized() and dist.get_backend() == "nccl" and dist.get_world_size() == 1
ok = True


def check(name, cond):
    global ok
    print(f"RCCL_EP {name}: {'OK' if cond else 'MISMATCH'}", flush=True)
    ok &= bool(cond)


# ---- (a) primitives ----
E = 8
ep = parallel.ExpertParallel(0, 1, E, padded=False, loopback=True)
assert not ep.local
side = torch.cuda.Stream(device=dev)
x = torch.randn(3000, 256, device=dev).to(torch.bfloat16)
send = torch.empty_like(x)
recv = torch.zeros_like(x)
send.copy_(x)                                                  # (queued on the launch stream: the collective must wait for it)
wait = ep.all_to_all_v(send[:2777], [2777], recv[:2777], [2777], side)
wait()
y = recv.clone()
torch.cuda.synchronize()
check("all_to_all_v unequal split on the side stream", torch.equal(y[:2777], x[:2777]) and float(y[2777:].abs().sum()) == 0.0)
r2, wait = ep.all_to_all(x, side)
wait()
check("all_to_all equal split", r2.data_ptr() != x.data_ptr() and torch.equal(r2, x))
counts = torch.randint(0, 400, (4, E), device=dev, dtype=torch.int32)
rc = ep.exchange_counts(counts, 256, side)()
check("exchange_counts", torch.equal(rc, counts.clamp(max=256)))
gc = torch.tensor([5, 0, 17, 300, 1, 0, 64, 9], device=dev, dtype=torch.int32)
rows = torch.randn(int(gc.sum()), 256, device=dev)
back, rcnt = ep.all_to_all_ragged(rows, gc)
back2, _ = ep.all_to_all_ragged(back, rcnt, recv_counts=gc)
check("all_to_all_ragged there and back", torch.equal(back, rows) and torch.equal(rcnt, gc) and torch.equal(back2, rows))
ar = parallel.make_grad_allreduce(loopback=True)
assert ar.active
g = torch.zeros(1 << 20, device=dev)
ref = torch.randn(1 << 20, device=dev)
g.copy_(ref)                                                   # producer on the launch stream
ar.begin(g[4096:], side)
g[:4096].mul_(2.0)                                             # work on the launch stream while the early part travels
scale = ar.finish(g[:4096])
g2 = g.clone()
torch.cuda.synchronize()
check("GradAllReduce begin / finish", scale == 1.0 and torch.equal(g2[4096:], ref[4096:]) and torch.equal(g2[:4096], ref[:4096] * 2.0)
      and ar.stale_drains == 0)

# ---- the model steps ----
N, S, chunk = 128, 64, 2048
batches = []
for it in range(3):
    rays, img, rgbs = synth.make_rays(900 + 10 * it, N)
    batches.append(tuple(torch.from_numpy(v).to(dev) for v in (rays, img, rgbs)))


def run(ep_obj, steps, graph=False, allreduce=None):
    m = SwitchNeRF(synth.BUILDING, dtype=torch.bfloat16, device=dev)
    m.load_state_dict(synth.make_weights(901, synth.BUILDING, gate_scale=1.0))
    m.set_expert_parallel(ep_obj)
    step = None
    if graph:
        step = GraphedTrainStep(m, batches[0][2], batches[0][0], batches[0][1], S, chunk, perturb=0.0, noise_std=0.0)
    losses = []
    for it in range(steps):
        rays, img, rgbs = batches[it % 3]
        if step is None:
            r = m.train_step(rgbs, rays, img, S, chunk, perturb=0.0, grad_allreduce=allreduce)
        else:
            r = step(rgbs, rays, img, grad_allreduce=allreduce)
        losses.append(float(r["loss"].item()))
    torch.cuda.synchronize()
    return m.flat.clone(), m.m.clone(), m.v.clone(), losses


def same(a, b):
    return all(torch.equal(p, q) for p, q in zip(a[:3], b[:3])) and a[3] == b[3]


# (b) kept rows only, unequal splits sized on the host: RCCL carries what the aliased buffers hold without it
local_kept = run(parallel.ExpertParallel(0, 1, E, padded=False), 3)
loop_kept = run(parallel.ExpertParallel(0, 1, E, padded=False, loopback=True), 3, allreduce=parallel.make_grad_allreduce(loopback=True))
check("kept-rows step over RCCL == no-collective step (3 optimizer steps)", same(local_kept, loop_kept) and loop_kept[3][0] > 0)
# (c) the padded step: eager over RCCL, then captured with the collectives inside and replayed
eager_pad = run(parallel.ExpertParallel(0, 1, E, padded=True, loopback=True), n_replays, allreduce=parallel.make_grad_allreduce(loopback=True))
graph_pad = run(parallel.ExpertParallel(0, 1, E, padded=True, loopback=True), n_replays, graph=True,
                allreduce=parallel.make_grad_allreduce(loopback=True))
check(f"padded step captured with its RCCL all-to-alls, {n_replays} replays == eager", same(eager_pad, graph_pad))
local_pad = run(parallel.ExpertParallel(0, 1, E, padded=True), 3)
check("padded step over RCCL == no-collective padded step", all(torch.equal(p, q) for p, q in zip(run(parallel.ExpertParallel(0, 1, E, padded=True, loopback=True), 3)[:3], local_pad[:3])))
print(f"RCCL_EP done: {'OK' if ok else 'MISMATCH'}", flush=True)
dist.barrier()
parallel.shutdown(exit_code=0 if ok else 1)      # (the teardown behind replayed graphs with captured all-to-alls: see parallel.shutdown)
sys.exit(0 if ok else 1)
