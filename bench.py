#!/usr/bin/env python
"""bench.py - train rays/s of the Switch-NeRF hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
        N > 1 without torchrun's environment: bench.py re-launches itself as N ranks
        (python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py <same flags>)
        and fails if fewer than N devices are visible.  Under torchrun it reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*.

One step = one full training step of BASELINE.json configs[1]: 8192 synthetic rays x 256 samples (routing segments of 131072
points), 8 experts top-1, capacity_factor 1.0, batch-prioritised routing, bf16 compute, stratified perturbation + sigma noise
drawn inside the timed region, forward + loss + backward + gradient all-reduce (N > 1) + Adam.

Scaling (N > 1): "strong" (default) = the reference's mode, the 8192-ray batch is split over the ranks (runner.py:573-575: batch_size
// world_size rays per rank; SURVEY cfg 3 = 1024 rays per GPU at N = 8); "weak" = 8192 rays per rank.  `value` is always the whole
job's rays per second.  Rays are independent: ranks are data-parallel replicas (one RCCL all-reduce of the flat fp32 gradient
buffer per step) or, with --parallelism ep, expert-parallel (dispatched rows exchanged with RCCL all-to-all).

Reproducibility: the device RNG is seeded, and after the warm-up steps the parameters, the Adam state and the RNG are reset, so the
timed steps are the same computation (same routing, same kept-token fraction) whatever the warm-up count.

Prints ONE JSON line (rank 0).  `roofline` describes the DOMINANT expert kernel (the slowest of the three expert launches of a training
step - forward + dense tail + heads, tail backward + combine backward + expert backward-data, weight gradients: within 15 % of each
other; `roofline.kernel` names it) on the roofline its arithmetic intensity puts it under - `bound` "hbm" when its flop per algorithmic
byte are below the ridge MFMA peak / HBM peak = 312 flop/B (the training chains save every activation: ~250 flop/B; the weight gradients
read every operand once: 128 flop/B), else "mfma";
`achieved` / `peak` / `frac` are on that scale, `attainable` = min(MFMA peak, flop_per_byte x HBM peak) in TFLOP/s and `mfma_frac` put the
same launch on the matrix-pipe scale, `traffic` = HBM bytes per launch from the PMC counters (profiles/traffic.json).  Timed live:
every expert kernel is relaunched back to back on one step's live buffers between two HIP events on the launch stream.  `kernels`
carries the MFMA fraction, the algorithmic HBM rate and the counter-measured HBM bytes of all three expert launches plus the save-free
forward launch (`expert_fwd_nosave`: what an inference forward runs) and the seven expert layers alone (`expert_gemm_nosave`: the grouped
GEMM of north_star, also `roofline.grouped_gemm_mfma_frac_nosave`); `balanced` repeats the measurement with perfectly balanced routing (point i -> expert i mod E:
every group full, 100 % of the tokens kept; its rays/s is also in `config.balanced_value`); `cpu_baseline` is the CPU oracle (a port of
the reference's CPU path) timed on this box's host cores on a bounded sample (one 131072-point segment): `cores` = the cores this
process may run on (os.sched_getaffinity), `threads` = torch's intra-op threads (what the oracle actually used).
"""
import argparse
import json
import os
import sys
import time


def _supervise():
    """Single-GPU invocations run the measurement in a CHILD process and forward its output: if the child dies on a signal before it has
    printed its line (seen once in ~60 runs of round 6: SIGSEGV, cause not found - profiles/r06_experiments.md 6), it is started once
    more.  Nothing is measured in this process; multi-rank launches (torchrun / --gpus N) are not wrapped."""
    if os.environ.get("SWN_BENCH_CHILD") == "1" or "WORLD_SIZE" in os.environ:
        return
    argv = sys.argv[1:]
    if "--gpus" in argv and argv[argv.index("--gpus") + 1:][:1] not in (["1"],):
        return
    if any(a_.startswith("--gpus=") and a_ != "--gpus=1" for a_ in argv):
        return
    import subprocess
    env = dict(os.environ, SWN_BENCH_CHILD="1")
    rc = 1
    for attempt in (1, 2):
        r = subprocess.run([sys.executable, os.path.abspath(__file__)] + argv, env=env, stdout=subprocess.PIPE)
        out = r.stdout.decode(errors="replace")
        rc = r.returncode
        if rc < 0 and not any(l.startswith("{") for l in out.splitlines()) and attempt == 1:
            print(f"bench.py: the measuring process died on signal {-rc} before its line was out; starting it once more", file=sys.stderr, flush=True)
            continue
        sys.stdout.write(out)
        sys.stdout.flush()
        break
    sys.exit(0 if rc < 0 and any(l.startswith("{") for l in out.splitlines()) else rc)


if __name__ == "__main__":
    _supervise()

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA peak


def synth_batch(n_rays, seed, device):
    g = torch.Generator().manual_seed(seed)
    o = torch.rand(n_rays, 3, generator=g) * 0.2 - 0.1
    d = torch.nn.functional.normalize(torch.randn(n_rays, 3, generator=g), dim=1)
    rays = torch.cat([o, d, torch.full((n_rays, 1), 0.05), torch.full((n_rays, 1), 1.0)], 1)
    idx = torch.randint(0, 10, (n_rays,), generator=g)
    rgbs = torch.rand(n_rays, 3, generator=g)
    return rays.to(device), idx.to(device), rgbs.to(device)


def _host_cores():
    """Cores this process may run on (len(os.sched_getaffinity(0)))."""
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


def cpu_baseline(seconds_budget=30.0):
    """The CPU oracle's training step (fwd + bwd, fp32) on one routing segment; returns rays/s on the host cores."""
    import numpy as np
    import synth
    from oracle import switchnerf_oracle as O
    n_rays, S, chunk = 512, 256, 131072
    sd = synth.make_weights(0, synth.BUILDING)
    rays, img, rgbs = synth.make_rays(1, n_rays)
    p = O.params_from_numpy(sd, requires_grad=True)
    t_best, reps, t_start = None, 0, time.time()
    while reps < 2 and (time.time() - t_start) < seconds_budget:
        for t in p.values():
            t.grad = None
        t0 = time.time()
        st = O.training_step(p, torch.from_numpy(rays), torch.from_numpy(img), torch.from_numpy(rgbs), synth.BUILDING, S, chunk)
        st["loss"].backward()
        dt = time.time() - t0
        t_best = dt if t_best is None else min(t_best, dt)
        reps += 1
    return dict(value=n_rays / t_best, unit="rays/s", cores=_host_cores(), threads=torch.get_num_threads(), kind="port",
                sample=f"oracle fwd+bwd fp32 on 1 of 16 segments (512 rays x 256 samples = 131072 points), best of {reps}; "
                       f"{t_best:.2f} s/segment => {16 * t_best:.1f} s per 8192-ray step")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--rays", type=int, default=8192)
    ap.add_argument("--samples", type=int, default=256)
    ap.add_argument("--chunk", type=int, default=131072)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16", "fp32"],
                    help="compute dtype: bf16 (headline, the reference's amp_use_bfloat16), fp16 (IEEE half + loss scaling: BASELINE configs[4]), fp32")
    ap.add_argument("--gate-scale", type=float, default=1.0, help="scale of the router weight (small => balanced routing)")
    ap.add_argument("--parallelism", default="dp", choices=["dp", "ep"],
                    help="dp (default): experts replicated, one gradient all-reduce per step; ep: experts sharded over the ranks, "
                         "dispatched rows exchanged with RCCL all-to-all (BASELINE.json configs[2]; needs gpus | 8)")
    ap.add_argument("--ep-padded", default="off", choices=["off", "on", "auto"],
                    help="expert parallel: off = kept rows only, unequal splits (one host read per forward pass, eager launches); on = the "
                         "reference's capacity-padded equal splits (nothing read on the host: the step can be replayed from a hipGraph with "
                         "--graph on); auto = padded when a segment's payload is at most 64 MiB")
    ap.add_argument("--ep-owner-tail", action="store_true", help="expert parallel: the dense tail on the EXPERT's rank (ep_owner.py) - both fused "
                    "launches stay, 0.53 x the bytes; eager launches")
    ap.add_argument("--fine", type=int, default=0, help="hierarchical sampling: fine samples per ray on top of --samples (other recipes; "
                                                        "the headline metric is --fine 0)")
    ap.add_argument("--mip", action="store_true", help="mip recipe: --samples edges per level (frustums = edges - 1), two levels")
    ap.add_argument("--model-dim", type=int, default=256, help="layer width (512 = mission_bay.yaml, other recipes)")
    ap.add_argument("--experts", type=int, default=8)
    ap.add_argument("--bg", action="store_true", help="with the dense background model behind an ellipsoidal foreground bound "
                    "(the Mega-NeRF scenes' default recipe, rendering.py:32-159), other recipes")
    ap.add_argument("--hash", action="store_true", help="BASELINE configs[4] input: multiresolution hash-grid encoding (16 levels, "
                    "2^19 entries x 2 features) instead of the frequency encoding, other recipes")
    ap.add_argument("--capacity-factor", type=float, default=1.0)
    ap.add_argument("--eval", action="store_true", help="inference only (render path: forward without activation saves, no "
                    "perturbation / noise / backward / Adam), other recipes")
    ap.add_argument("--dense", action="store_true", help="BASELINE configs[0]: the dense NeRF (--no-use_moe), other recipes")
    ap.add_argument("--scaling", default=None, choices=["strong", "weak"],
                    help="N > 1: strong (default) = the --rays batch is split over the ranks like the reference (runner.py:573-575); "
                         "weak = --rays per rank")
    ap.add_argument("--graph", default="auto", choices=["auto", "on", "off"],
                    help="replay the step's forward + backward launch sequence from a hipGraph (switch_nerf_amd/graph.py).  auto = on for "
                         "the headline recipe in data parallel: the FIRST process on a fresh box issues its ~150 launches per step 10 % "
                         "slower than the GPU runs them (18.5-19.7 ms/step against 16.7 for every later process; 200 warm-up steps do not "
                         "cure it), replay takes the host out of the step; per-kernel HIP events cannot be recorded inside a graph, they "
                         "come from eager steps that follow the timed region.  off: eager launches, events inside the timed region")
    ap.add_argument("--no-split-backward", action="store_true", help="N > 1: one backward graph and one all-reduce behind it (default: two "
                    "graphs, the expert block's all-reduce on the side stream under the second one)")
    ap.add_argument("--no-balanced", action="store_true", help="skip the extra balanced-routing measurement")
    ap.add_argument("--routing", choices=["router", "balanced"], default="router",
                    help="balanced: the MAIN measurement runs with point i -> expert i mod E (used by the counter passes: bytes per kept row)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-events", action="store_true", help="do not record per-kernel HIP events in the timed region")
    ap.add_argument("--loopback", action="store_true", help="--gpus 1 only: run the MULTI-GPU step on one GPU - a world-1 RCCL process "
                    "group, every collective of the N > 1 path issued for real (gradient all-reduce in two parts around the second "
                    "backward graph, the expert-parallel all-to-alls with their split sizes on the side stream); not the headline")
    ap.add_argument("--no-ep-probe", action="store_true",
                    help="N > 1, data parallel: skip the expert-parallel measurement that otherwise follows the headline in the same line")
    ap.add_argument("--ep-probe-limit", type=float, default=180.0,
                    help="seconds the expert-parallel probe may take before the line is printed without it (a collective that never returns)")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(a.gpus)                  # does not return
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("SWN_FORCE_DEVICE", os.environ.get("LOCAL_RANK", 0)))     # (override: several ranks on one GPU, for testing)
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}: launch {a.gpus} ranks (or drop WORLD_SIZE and let bench.py launch them)")
    if local >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: rank {rank} wants device {local} but only {torch.cuda.device_count()} are visible")
    scaling = a.scaling or "strong"      # (the metric splits ONE global batch over the ranks, runner.py:575; N = 1: the same thing either way)
    if a.loopback and world != 1:
        raise SystemExit("bench.py: --loopback is the one-GPU rehearsal of the multi-GPU step (--gpus 1)")
    multi = world > 1 or a.loopback      # the step issues collectives
    if scaling == "strong" and a.rays % world:
        raise SystemExit(f"bench.py: --rays {a.rays} is not divisible by {world} ranks")
    n_rays = a.rays // world if scaling == "strong" else a.rays          # rays of THIS rank per step
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if multi:
        import torch.distributed as dist
        from switch_nerf_amd import parallel
        parallel.init_from_env(os.environ.get("SWN_DIST_BACKEND", "nccl"), dev, loopback=a.loopback)       # "nccl" = RCCL; gloo only for single-GPU tests of this path

    from switch_nerf_amd import _lib
    from switch_nerf_amd.model import SwitchNeRF, BUILDING
    dtype = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[a.dtype]
    cfg = dict(BUILDING, model_dim=a.model_dim, gate_hidden=a.model_dim, num_experts=a.experts)
    if a.hash:
        cfg["hash"] = dict(n_levels=16, log2_table=19, base_res=16, per_level_scale=1.3819, aabb_lo=(-1.2, -1.2, -1.2), aabb_hi=(1.2, 1.2, 1.2))
    other = a.dtype != "bf16" or a.fine or a.mip or a.model_dim != 256 or a.experts != 8 or a.dense or a.bg or a.hash or a.capacity_factor != 1.0 or a.eval
    if a.dense:
        from switch_nerf_amd.dense import DenseNeRF
        model = DenseNeRF(dtype=dtype, device=dev, seed=0)
    else:
        model = SwitchNeRF(cfg, dtype=dtype, device=dev, seed=0, capacity_factor=a.capacity_factor)
    if a.hash:      # emulate a trained encoding (features O(1)): the standard U(-1e-4, 1e-4) initialisation gives every point the
        model.p["hash.table"].mul_(1e4)      # same first-layer output and the random-init router sends everything to one expert
    if a.gate_scale != 1.0 and not a.dense:
        model.p["wg"].mul_(a.gate_scale)
    if scaling == "strong":      # this rank's contiguous slice of the global batch (runner.py:575)
        rays, idx, rgbs = (t[rank * n_rays:(rank + 1) * n_rays].contiguous() for t in synth_batch(a.rays, 1000, dev))
    else:
        rays, idx, rgbs = synth_batch(n_rays, 1000 + rank, dev)
    P = n_rays * a.samples
    a.chunk = min(a.chunk, P)

    if multi:
        allreduce = parallel.make_grad_allreduce(loopback=a.loopback)     # RCCL all-reduce over xGMI, one 16 MB bucket
    if a.parallelism == "ep":
        from switch_nerf_amd.parallel import ExpertParallel
        model.set_expert_parallel(ExpertParallel(rank, world, model.E, padded={"off": False, "on": True, "auto": "auto"}[a.ep_padded],
                                                 loopback=a.loopback, owner_tail=a.ep_owner_tail))

    radii = torch.full((n_rays, 1), 1e-3, device=dev)
    scene = None
    if a.bg:
        from switch_nerf_amd.background import BackgroundScene
        from switch_nerf_amd.dense import DenseNeRF, DENSE
        bg = DenseNeRF(dict(DENSE, xyz_dim=4), dtype=dtype, device=dev, seed=1)
        scene = BackgroundScene(model, bg, [0.02, -0.03, 0.01], [0.6, 0.8, 0.7])
        rays[:, 7] = torch.rand(n_rays, device=dev) * 1.2 + 0.3          # about half of the rays leave the bound

    eval_counts = [None, None]
    if a.eval:                    # (the kept-token fraction of the JSON line: one eager forward's routing)
        with torch.no_grad():
            c0_ = model.forward_rays(rays, idx, a.samples, a.chunk, 0.0, None, None, training=False)
        eval_counts = [c0_["counts"].clone(), c0_["cap"]]
    route_override = [None]       # [P] int32 expert of every point (the balanced-routing measurement) or None = the router's choice
    plain = not (a.eval or a.mip or a.bg or a.fine or a.dense)
    # (expert parallel: only the padded mode is capturable, and only on request - RCCL collectives inside a captured graph have not
    #  run on hardware yet)
    use_graph = plain and ((a.parallelism == "dp" and a.graph in ("on", "auto")) or
                           (a.parallelism == "ep" and a.ep_padded != "off" and a.graph == "on"))
    graphed = [None]
    if use_graph:
        from switch_nerf_amd.graph import GraphedTrainStep
    split_bwd = False if a.no_split_backward else (True if a.loopback else None)      # (None: two backward graphs when N > 1)

    def step():
        if a.eval:       # render_rays in eval mode (runner.py:2835-2885 render_image's inner call): forward only
            if a.graph != "off":      # replayed from a hipGraph (graph.GraphedRender = rendering.render_rays with nerf.graph_eval)
                if graphed[0] is None:
                    from switch_nerf_amd.graph import GraphedRender
                    graphed[0] = GraphedRender(model, rays, idx, a.samples, a.chunk)
                o_ = graphed[0](rays, idx)
                return dict(ctx=dict(counts=eval_counts[0], cap=eval_counts[1]), loss=o_["rgb"].sum() * 0)
            with torch.no_grad():
                c = model.forward_rays(rays, idx, a.samples, a.chunk, 0.0, None, None, training=False)
            return dict(ctx=c, loss=c["rgb"].sum() * 0)
        pr = torch.rand(n_rays, a.samples, device=dev)              # rendering.py:582 rand_like
        ar = allreduce if multi else None
        if a.mip:      # rendering_mip recipe: a.samples edges -> a.samples - 1 frustums per level, coarse + fine level
            nf = n_rays * (a.samples - 1)
            return model.train_step_mip(rgbs, rays, radii, idx, a.samples, a.samples, a.chunk, perturb=1.0, perturb_rand=pr,
                                        sigma_noise=torch.randn(nf, device=dev), sigma_noise_fine=torch.randn(nf, device=dev),
                                        grad_allreduce=ar)
        if graphed[0] is not None:   # forward + loss + backward replayed from the hipGraph (jitter / noise drawn inside it), then
            return graphed[0](grad_allreduce=ar)                    # all-reduce + Adam + compute-copy refresh
        noise = torch.randn(P, device=dev)                          # rendering.py:366, sigma_noise_std = 1
        if scene is not None:
            return scene.train_step(rgbs, rays, idx, a.samples, a.chunk, perturb=1.0, perturb_rand=pr, sigma_noise=noise,
                                    sigma_noise_bg="randn", sigma_noise_bg_fine="randn", noise_std=1.0, grad_allreduce=ar,
                                    fine_samples=a.fine,
                                    sigma_noise_fine=torch.randn(n_rays * a.fine, device=dev) if a.fine else None)
        kw = {}
        if a.fine > 0:
            kw = dict(fine_samples=a.fine, sigma_noise_fine=torch.randn(n_rays * a.fine, device=dev))
        return model.train_step(rgbs, rays, idx, a.samples, a.chunk, perturb=1.0, perturb_rand=pr, sigma_noise=noise,
                                grad_allreduce=ar, routing_override=route_override[0], **kw)

    models = [model] + ([scene.bg] if scene is not None else [])
    snap = [m.flat.clone() for m in models]

    def reset_state(seed):
        """Parameters, Adam state and RNG back to the start: what follows is the same computation in every run."""
        for m, f0 in zip(models, snap):
            m.flat.copy_(f0); m.m.zero_(); m.v.zero_(); m.step_count = 0
            m.refresh_compute_copies()
        torch.manual_seed(seed + rank)
        torch.cuda.manual_seed(seed + rank)

    kept_acc = [None]             # mean kept rows per step of the last timed() call with events on

    def timed(steps, events):
        model.profile = events
        model.events = {}
        if dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        st = None
        kept_acc[0] = None
        for _ in range(steps):
            st = step()
            if events and not a.dense:       # work follows the kept rows and routing moves while training: average them like the events
                c_ = st["ctx"]["c"] if a.bg else st["ctx"]
                k_ = c_["counts"].clamp(max=c_["cap"]).sum()
                kept_acc[0] = k_ if kept_acc[0] is None else kept_acc[0] + k_
        if kept_acc[0] is not None:
            kept_acc[0] = float(kept_acc[0].item()) / steps
        if dist:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if dist:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = t.item()
        model.profile = False
        return dt, st

    fused_tail = [False, False]      # the step's expert forward / backward launches carry the dense tail (SwitchNeRF._tail_fused)

    def kernel_times(reps=5):
        """Per-kernel durations that do not depend on how fast the host enqueues a step: ONE eager step with the relaunch hooks on
        (SwitchNeRF.profile), then every expert kernel `reps` times back to back on that step's live buffers between two HIP events
        on the launch stream.  Returns ({name: ms}, kept rows of that step)."""
        model.profile = True
        st_ = step()
        model.profile = False
        c_ = st_["ctx"]["c"] if a.bg else st_["ctx"]
        hooks = c_.get("_relaunch", {})
        out_ = {}
        for name in ("expert_fwd", "expert_bwd", "expert_wgrad", "expert_fwd_nosave", "expert_gemm_nosave"):
            fn = hooks.get(name)
            if fn is None:
                continue
            fn()
            torch.cuda.synchronize()
            best = None
            for _batch in range(3):      # the fastest of three batches of `reps` launches (a batch right behind an eager step can sit
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)      # in a clock ramp: seen once as +20 %)
                e0.record()
                for _ in range(reps):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                t_ = e0.elapsed_time(e1) / reps
                best = t_ if best is None else min(best, t_)
            out_[name] = best
        kept_ = int(torch.minimum(c_["counts"], torch.tensor(c_["cap"], device=dev)).sum().item())
        fused_tail[0] = bool(c_.get("tail_fused"))
        fused_tail[1] = fused_tail[0] and bool(model.sw["fused_tail_bwd"])
        return out_, kept_

    if a.routing == "balanced":
        route_override[0] = (torch.arange(P, device=dev, dtype=torch.int32) % a.experts).contiguous()
    reset_state(4321)
    if use_graph:
        graphed[0] = GraphedTrainStep(model, rgbs, rays, idx, a.samples, a.chunk, perturb=1.0, noise_std=1.0,
                                      routing_override=route_override[0], split_backward=split_bwd)
    for _ in range(a.warmup):
        st = step()
    reset_state(1234)
    dt, st = timed(a.steps, False)
    ms = dt / a.steps * 1e3
    ev_steps = max(3, min(10, a.steps))
    eager_ms = None
    ktimes, kkept = {}, None
    if use_graph:
        graphed[0] = None
    if plain and not a.no_events:
        edt, st = timed(ev_steps, False)     # the same steps launched eagerly (reported next to the replayed number)
        eager_ms = edt / ev_steps * 1e3
        ktimes, kkept = kernel_times()       # back-to-back relaunches of the expert kernels on the next step's live buffers
    value = n_rays * world * a.steps / dt

    # ---- per-kernel accounting from the live HIP events
    L, M, E = model.L, model.M, model.E
    esz = 4 if dtype == torch.float32 else 2
    traffic_tab = {}
    tp = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tp):
        try:
            traffic_tab = json.load(open(tp))
        except Exception:
            traffic_tab = {}
    csrc_sha = _lib.source_hash()
    traffic_ok, traffic_note = False, "profiles/traffic.json missing"
    if traffic_tab:
        if traffic_tab.get("csrc_sha256") != csrc_sha:
            traffic_note = ("profiles/traffic.json was collected on other kernel sources (csrc_sha256 "
                            f"{str(traffic_tab.get('csrc_sha256'))[:12]} != {csrc_sha[:12]}): traffic not reported")
        elif traffic_tab.get("points") != P:
            traffic_note = f"profiles/traffic.json was collected at {traffic_tab.get('points')} points per launch, this run has {P}"
        else:
            traffic_ok, traffic_note = True, traffic_tab.get("_how")
    names = {"expert_fwd": "chainq_kernel<Bf16,1,true> (expert forward: 7 fused layers, persistent 256-row workgroups on a tile queue, row groups half a layer apart)",
             "expert_bwd": "chainq_kernel<Bf16,2,true> (expert backward-data: 7 fused layers, persistent 256-row workgroups on a tile queue, row groups half a layer apart)",
             "expert_wgrad": "wgrad_stream_kernel<bf16,1> (expert weight gradients, 7 layers in one balanced launch)",
             "expert_fwd_nosave": "chainq_kernel<Bf16,1,true> without activation saves (the inference / --eval expert chain on the same rows: "
                                  "the grouped GEMM alone)"}

    def kept_of(st_):
        c_ = st_["ctx"]["c"] if a.bg else st_["ctx"]
        return P if a.dense else int(torch.minimum(c_["counts"], torch.tensor(c_["cap"], device=dev)).sum().item())

    def flops_of(kept_):
        return 2.0 * L * M * M * kept_

    def library_yardstick(kept_):
        """The reference's own way of running the expert MLP (tutel_moe_layer_nobatch.py:887-924: seven torch.baddbmm + ReLU over the
        dispatched [E, rows, M] tensor, under bf16 autocast) on THIS GPU with torch's library GEMM (hipBLASLt / rocBLAS), on as many rows
        per expert as the step keeps on average: what a non-fused grouped GEMM reaches here (DVFS, real data) next to the fused chain."""
        rows = max(256, int(kept_ // E) // 256 * 256)
        x = torch.randn(E, rows, M, device=dev).to(torch.bfloat16)
        ws = [(torch.randn(E, M, M, device=dev) / 16).to(torch.bfloat16) for _ in range(L)]
        bs = [torch.zeros(E, 1, M, device=dev, dtype=torch.bfloat16) for _ in range(L)]

        def run():
            h = x
            for l in range(L):
                h = torch.baddbmm(bs[l], h, ws[l])
                if l < L - 1:
                    h = torch.relu(h)
            return h
        run(); run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms_ = e0.elapsed_time(e1) / 5
        tf = 2.0 * L * M * M * E * rows / (ms_ * 1e-3) / 1e12
        return dict(what="7 x (torch.baddbmm + ReLU) on [E, rows, 256] bf16 = the reference's ExpertMLP.forward on this GPU (library GEMMs, "
                         "activations through HBM between the layers, no saves)", rows_per_expert=rows, ms=round(ms_, 4), tflops=round(tf, 1),
                    mfma_frac=round(tf / MFMA_BF16_PEAK_TFLOPS, 4))

    def gemm_yardsticks(kept_):
        """What the matrix pipe and the library deliver on THIS box (clocks under load are 1.3-1.9 GHz, the 2.5 PFLOP/s peak is quoted at
        2.4 GHz): (a) one large square bf16 GEMM - the practical ceiling of any MFMA kernel here; (b) ONE layer of the expert MLP as a
        plain [rows, 256] x [256, 256] GEMM over all kept rows - what a layer-by-layer (unfused) grouped GEMM is bounded by: it reads
        and writes every activation through HBM."""
        def timed(fn, n=5):
            fn(); fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / n
        n = 8192
        a_, b_ = torch.randn(n, n, device=dev).to(torch.bfloat16), torch.randn(n, n, device=dev).to(torch.bfloat16)
        ms_sq = timed(lambda: torch.matmul(a_, b_))
        tf_sq = 2.0 * n * n * n / (ms_sq * 1e-3) / 1e12
        del a_, b_
        rows = int(kept_) // 256 * 256
        x_, w_ = torch.randn(rows, M, device=dev).to(torch.bfloat16), (torch.randn(M, M, device=dev) / 16).to(torch.bfloat16)
        ms_l = timed(lambda: torch.matmul(x_, w_))
        tf_l = 2.0 * rows * M * M / (ms_l * 1e-3) / 1e12
        return dict(square_gemm_8192=dict(ms=round(ms_sq, 4), tflops=round(tf_sq, 1), mfma_frac=round(tf_sq / MFMA_BF16_PEAK_TFLOPS, 4)),
                    one_layer_gemm=dict(rows=rows, ms=round(ms_l, 4), tflops=round(tf_l, 1), mfma_frac=round(tf_l / MFMA_BF16_PEAK_TFLOPS, 4),
                                        gbs=round(rows * M * 2 * 2 / (ms_l * 1e-3) / 1e9, 1)))

    def alg_of(kept_):
        """(algorithmic HBM bytes, algorithmic flops) per launch name for `kept_` kept rows."""
        flops_e = 2.0 * L * M * M * kept_
        alg = {"expert_fwd": kept_ * M * esz * (1 + (L - 1) + 1), "expert_bwd": kept_ * M * esz * (1 + (L - 1) + 1 + 1),
               "expert_wgrad": kept_ * M * esz * 2 * L, "expert_fwd_nosave": kept_ * M * esz * 2, "expert_gemm_nosave": kept_ * M * esz * 2}
        fl = {k: flops_e for k in alg}
        if fused_tail[0]:
            # the forward launch also runs the dense tail on EVERY point (kept or not): Linear "1" (M x M) and Linear "2" (M x H2); it
            # reads the kept rows and the per-ray bias (fp32, one row per point), writes the L - 1 expert activations (kept rows), y / h1
            # (M) and h2 (H2) per point for the backward and raw (16 B per point) - the save-free form writes raw only
            H2_ = model.H2
            tail_fl = 2.0 * (M * M + M * H2_) * P
            fl["expert_fwd"] += tail_fl
            fl["expert_fwd_nosave"] += tail_fl
            alg["expert_fwd"] = kept_ * M * esz * (1 + (L - 1)) + P * (2 * M * esz + H2_ * esz + H2_ * 4 + 16)
            alg["expert_fwd_nosave"] = kept_ * M * esz + P * (H2_ * 4 + 16)
        if fused_tail[1]:
            # ... and the backward launch runs the tail's two backward layers and the combine backward in front of the experts': reads dh2
            # (H2) and y (M) per point and the skip layer's dZ, writes dh1 (M) per point, L dZ (the last expert layer's included) + dx per kept row
            fl["expert_bwd"] += 2.0 * (M * M + M * model.H2) * P
            alg["expert_bwd"] = kept_ * M * esz * (L + 1 + 1) + P * (model.H2 * esz + 2 * M * esz + 12)
        return alg, fl

    def traffic_of(name, kept_):
        """HBM bytes of launch `name` from the counter passes (profiles/traffic.json, scripts/make_traffic.py): measured on the timed
        workload at traffic_tab['kept_rows'] kept rows, scaled to this run's kept rows by the ratio of the algorithmic bytes (a few
        percent: routing moves while training).  None when the table was collected on other kernel sources, another kernel set or
        another problem size."""
        if not traffic_ok:
            return None
        t_ = traffic_tab.get("launches", {}).get(name)
        if not t_:
            return None
        ks_t, ks_now = traffic_tab.get("kernel_set") or {}, model.kernel_set()
        if any(ks_t.get(k) != ks_now.get(k) for k in ("geom", "front_geom", "tail_fused", "fused_backward", "comb_dwsig") if k in ks_t):
            return None
        a_now, _ = alg_of(kept_)
        a_pmc, _ = alg_of(traffic_tab["kept_rows"])
        return t_["hbm_bytes"] * a_now[name] / a_pmc[name]

    def account(events, kept_):
        """Expert kernels against both rooflines.  flops: 2 L M^2 per KEPT row for each of forward, backward-data and weight
        gradients.  Algorithmic HBM bytes per launch: fwd reads x, writes L-1 activations +
        the output; bwd reads dout and the skip layer's dZ, writes L-1 dZ + dx; the weight gradients read L layer inputs and L dZ.
        hbm_measured_*: bytes from the FETCH_SIZE / WRITE_SIZE counters of a separate profiling pass (profiles/traffic.json,
        stored per kept row), i.e. what actually crossed the memory-side fabric."""
        detail_ = {}
        alg, fl = alg_of(kept_)
        for name in ("expert_fwd", "expert_bwd", "expert_wgrad", "expert_fwd_nosave", "expert_gemm_nosave"):
            ms_ = events.get(name)
            if not ms_ or ms_ <= 0:
                continue
            flops = fl[name]
            tf = flops / (ms_ * 1e-3) / 1e12
            gbs = alg[name] / (ms_ * 1e-3) / 1e9
            d_ = dict(ms=round(ms_, 4), tflops=round(tf, 1), mfma_frac=round(tf / MFMA_BF16_PEAK_TFLOPS, 4), alg_bytes=int(alg[name]),
                      alg_flops=float(flops), alg_gbs=round(gbs, 1), hbm_frac_alg=round(gbs / HBM_PEAK_GBS, 4))
            tb = traffic_of(name, kept_)
            if tb:
                d_.update(hbm_measured_bytes=int(tb), hbm_measured_gbs=round(tb / (ms_ * 1e-3) / 1e9, 1),
                          hbm_frac_measured=round(tb / (ms_ * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), hbm_measured_over_alg=round(tb / alg[name], 4))
            detail_[name] = d_
        return detail_

    if fused_tail[0]:
        names["expert_fwd"] = ("chainq_kernel<Bf16,7,true> (expert forward, 7 fused layers, AND the dense tail behind it - gate scaling, Linear 1, "
                               "Linear 2 + per-ray bias, sigma / colour heads - on every point: persistent 256-row workgroups on a tile queue)")
        names["expert_fwd_nosave"] = "chainq_kernel<Bf16,7,true> without activation saves (the inference / --eval launch: experts + tail, raw is all it writes)"
    if fused_tail[1]:
        names["expert_bwd"] = ("chainq_kernel<Bf16,8,true> (the tail's two backward layers + the combine backward on every point, then the expert "
                               "backward-data chain, 7 fused layers: persistent 256-row workgroups on a tile queue; timed with the three small launches "
                               "around it that finish the sigma head's weight gradient - workspace fill, run sums, ordered reduce: ~0.04 ms)")
    kept = kept_of(st)
    kept_mean = kkept if kkept is not None else kept                   # kept rows of the step whose buffers the kernels were timed on
    detail = {} if other else account(ktimes, kept_mean)               # (other recipes: headline number only)
    roof = None
    if detail:
        # the expert grouped GEMM is priced against the bf16 MFMA peak; the dominant kernel = the slowest of its three
        # training launches.  (Their HBM side - a training chain must save every activation for the weight gradients - is in `kernels`;
        # `expert_fwd_nosave` is the same grouped GEMM without the saves.)
        dom = max((k for k in detail if k in ("expert_fwd", "expert_bwd", "expert_wgrad")), key=lambda k: detail[k]["ms"])
        d = detail[dom]
        # which roofline bounds it: arithmetic intensity (algorithmic flops / algorithmic bytes) against the ridge 2.5 PFLOP/s / 8 TB/s =
        # 312 flop/B.  The weight-gradient launch (reads every operand once: 128 flop/B) is HBM-bound; the chains that save every
        # activation sit at ~220 flop/B (HBM side as well), the save-free forward at 1790 flop/B (MFMA side).  Both fractions are reported.
        intensity = d["alg_flops"] / d["alg_bytes"]
        attainable = min(MFMA_BF16_PEAK_TFLOPS, intensity * HBM_PEAK_GBS * 1e9 / 1e12)      # TFLOP/s: both rooflines in one number
        if intensity < MFMA_BF16_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBS * 1e9):
            roof = dict(kernel=names[dom], bound="hbm", achieved=d["alg_gbs"], peak=HBM_PEAK_GBS, unit="GB/s", frac=d["hbm_frac_alg"],
                        traffic=d.get("hbm_measured_bytes"), flop_per_byte=round(intensity, 1), mfma_frac=d["mfma_frac"], tflops=d["tflops"],
                        hbm_frac_measured=d.get("hbm_frac_measured"))
        else:
            roof = dict(kernel=names[dom], bound="mfma", achieved=d["tflops"], peak=MFMA_BF16_PEAK_TFLOPS, unit="TFLOP/s",
                        frac=d["mfma_frac"], traffic=d.get("hbm_measured_bytes"), flop_per_byte=round(intensity, 1), alg_gbs=d["alg_gbs"],
                        hbm_frac_measured=d.get("hbm_frac_measured"))
        roof["traffic_source"] = traffic_note
        roof["attainable"] = dict(tflops=round(attainable, 1), frac_of_attainable=round(d["tflops"] / attainable, 4),
                                  what="min(MFMA peak, flop_per_byte x HBM peak) for this launch's algorithmic flops and bytes")
        # the three training launches side by side (the dominant one changes with the box and the router's fill: they are within 15 %)
        roof["launches"] = {k: dict(ms=detail[k]["ms"], hbm_frac_alg=detail[k]["hbm_frac_alg"], hbm_frac_measured=detail[k].get("hbm_frac_measured"),
                                    mfma_frac=detail[k]["mfma_frac"]) for k in ("expert_fwd", "expert_bwd", "expert_wgrad") if k in detail}
        if "expert_gemm_nosave" in detail:     # north_star: the grouped GEMM against the MFMA peak = the expert layers alone, without saves
            roof["grouped_gemm_mfma_frac_nosave"] = detail["expert_gemm_nosave"]["mfma_frac"]
        elif "expert_fwd_nosave" in detail:
            roof["grouped_gemm_mfma_frac_nosave"] = detail["expert_fwd_nosave"]["mfma_frac"]
    if detail and world == 1 and a.dtype == "bf16":
        try:
            detail["reference_style_library_gemms"] = library_yardstick(kept_mean)
        except Exception as e:      # informational only
            detail["reference_style_library_gemms"] = dict(error=str(e))
        try:
            detail["library_gemm_yardsticks"] = gemm_yardsticks(kept_mean)
        except Exception as e:
            detail["library_gemm_yardsticks"] = dict(error=str(e))
    loss_main = float(st["loss"].item())

    # ---- the same measurement with perfectly balanced routing (GEMM work follows the kept tokens; the random-init
    #      router fills some experts to capacity and starves others).  Point i goes to expert i mod E - every (segment, expert) group
    #      is exactly full, nothing is dropped; gate values, ranking, dispatch and every kernel run as usual.
    balanced = None
    if not other and not a.no_balanced and not a.fine and a.routing == "router":
        route_override[0] = (torch.arange(P, device=dev, dtype=torch.int32) % E).contiguous()
        reset_state(4321)
        if use_graph:
            graphed[0] = GraphedTrainStep(model, rgbs, rays, idx, a.samples, a.chunk, perturb=1.0, noise_std=1.0,
                                          routing_override=route_override[0])
        for _ in range(2):
            step()
        reset_state(1234)
        bsteps = max(2, min(10, a.steps))
        bdt, bst = timed(bsteps, False)
        if use_graph:
            graphed[0] = None
        bk, bkept = kernel_times() if not a.no_events else ({}, kept_of(bst))
        balanced = dict(routing="expert = point index mod E (every group full)", steps=bsteps, ms_per_step=round(bdt / bsteps * 1e3, 3),
                        value=round(n_rays * world * bsteps / bdt, 1), kept_token_fraction=round(bkept / P, 4),
                        kernels=account(bk, bkept))
        route_override[0] = None

    # ---- the loop a maintainer of the reference would run (runner.py:604-693): rendering.render_rays under autograd, loss.backward(),
    #      torch.optim.Adam on the flat parameter, ExponentialLR - forward / backward replayed from captured graphs (nerf.graph_train)
    runner_ms = None
    if plain and not other and world == 1 and not a.no_events:        # (the headline recipe only)
        from argparse import Namespace
        from switch_nerf_amd import rendering
        reset_state(1234)
        hp = Namespace(coarse_samples=a.samples, fine_samples=0, model_chunk_size=a.chunk, perturb=1.0, use_sigma_noise=True,
                       sigma_noise_std=1.0, use_cascade=False)
        model.graph_train = True
        model.train()
        opt = torch.optim.Adam(model.trainable_parameters(), lr=5e-4)
        sch = torch.optim.lr_scheduler.ExponentialLR(opt, gamma=0.1 ** (1 / 500000))

        def runner_step():
            res, _ = rendering.render_rays(model, None, rays, idx, hp, None, None, True, True, False)
            loss = torch.nn.functional.mse_loss(res["rgb_coarse"], rgbs) + model.wt * res["gate_loss_coarse"].mean()
            opt.zero_grad(set_to_none=True)
            loss.backward()
            opt.step()
            sch.step()
            return loss
        for _ in range(3):
            runner_step()
        torch.cuda.synchronize()
        rsteps = max(3, min(20, a.steps))
        t0 = time.perf_counter()
        for _ in range(rsteps):
            runner_step()
        torch.cuda.synchronize()
        runner_ms = (time.perf_counter() - t0) / rsteps * 1e3
        model.graph_train = False

    # ---- expert parallel: what travelled, and how much of the exchange was hidden behind the expert kernels
    def ep_report(psteps=3):
        """A few profiled expert-parallel steps -> what left this GPU, the collectives' time on the side stream, the time the launch
        stream waited for them, and the rows every rank's experts received (the GPU-level load imbalance of one expert per GPU)."""
        ep_ = model.ep
        ep_.profile, ep_.bytes_sent = True, 0
        ep_.overlap_report()
        st_ = None
        for _ in range(psteps):
            st_ = step()
        torch.cuda.synchronize()
        rep = ep_.overlap_report()
        ep_.profile = False
        c_ = st_["ctx"]
        kept_rows = int(c_["counts"].clamp(max=c_["cap"]).sum().item())
        mine = torch.tensor([int(c_["ep_counts"].sum().item())], device=dev, dtype=torch.int64)      # rows this rank's experts ran
        per_rank = [torch.zeros_like(mine) for _ in range(world)]
        if dist:
            dist.all_gather(per_rank, mine)
        else:
            per_rank = [mine]
        per_rank = [int(t.item()) for t in per_rank]
        mean_rows = sum(per_rank) / max(1, len(per_rank))
        owner_ = c_.get("ep_owner") is not None
        return dict(exchange=("tail on the expert's rank: kept rows + 16 B per token out, raw + 16 B back; d_raw out, dx + gate gradient back "
                              "(7 unequal-split all_to_all_single per STEP over all segments, blocking)" if owner_ else
                              ("capacity-padded equal-split" if c_.get("ep_padded") else "kept rows only, unequal-split") +
                              " all_to_all_single per routing segment on a side HIP stream; 4 exchanges per segment and step (dispatch / "
                              "return, forward / backward)"), padded=bool(c_.get("ep_padded")), owner_tail=owner_,
                    segments=int(c_["n_seg"]), kept_rows_per_step=kept_rows,
                    bytes_leaving_this_gpu_per_step=int(ep_.bytes_sent // psteps),
                    bytes_per_segment_exchange=int(ep_.bytes_sent // psteps // max(1, 4 * int(c_["n_seg"]))),
                    capacity_padded_bytes_per_step=int(4 * c_["n_seg"] * model.E * c_["cap"] * model.M * esz * (world - 1) // world),
                    collectives_per_step=rep["collectives"] // psteps, collective_ms_per_step=round(rep["collective_ms"] / psteps, 3),
                    wait_ms_per_step=round(rep["wait_ms"] / psteps, 3),
                    hidden_fraction=None if rep["hidden_fraction"] is None else round(rep["hidden_fraction"], 4),
                    expert_rows_per_rank=per_rank,
                    load_imbalance_max_over_mean=None if mean_rows <= 0 else round(max(per_rank) / mean_rows, 4),
                    tail=("inside the owner's fused launches (chain tags 7 / 8 on the received token space)" if owner_ else
                          "64-row tail launches (the fused tail rides the expert launch of LOCAL experts only: INTEGRATION.md section 5)"))

    ep_info = None
    if a.parallelism == "ep" and model.ep is not None:
        ep_info = ep_report()

    # ---- data parallel: the gradient all-reduce - its time and how much of it ran under the second backward graph
    ar_info = None
    if multi and a.parallelism == "dp" and hasattr(allreduce, "report"):
        if use_graph and graphed[0] is None:
            graphed[0] = GraphedTrainStep(model, rgbs, rays, idx, a.samples, a.chunk, perturb=1.0, noise_std=1.0, split_backward=split_bwd)
        allreduce.profile = True
        allreduce.report()
        psteps = 5
        for _ in range(psteps):
            step()
        torch.cuda.synchronize()
        rep = allreduce.report()
        allreduce.profile = False
        ar_info = dict(what="flat fp32 gradient, RCCL all-reduce; the expert block (behind n_dense) is issued on the side stream between the "
                            "two backward graphs, the dense prefix behind the second one" if (graphed[0] is not None and graphed[0].split)
                       else "flat fp32 gradient, one RCCL all-reduce behind the backward",
                       bytes=int(model.grad.numel() * 4), expert_block_bytes=int((model.grad.numel() - model.n_dense) * 4),
                       collectives_per_step=rep["collectives"] // psteps, allreduce_ms=round(rep["allreduce_ms"] / psteps, 4),
                       wait_ms=round(rep["wait_ms"] / psteps, 4),
                       allreduce_hidden_fraction=None if rep["hidden_fraction"] is None else round(rep["hidden_fraction"], 4))
        graphed[0] = None

    gb = a.rays if scaling == "strong" else a.rays * world
    out = {
        "metric": "train rays/sec (8192-ray batch, 256 samples, 8 experts)", "value": round(value, 1), "unit": "rays/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True,
        "scaling": scaling, "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
        "config": {"workload": ("other recipe (informational): " if other else "configs[1]: ")
                               + ("dense NeRF 8 x 256 (configs[0] network, --no-use_moe)" if a.dense else f"{a.experts}-expert top-1 expertmlp, capacity_factor=1.0, BPR")
                               + f", {gb} rays x {a.samples} samples per step over {world} GPU(s) ({n_rays} rays per GPU)"
                               f", {P // a.chunk} segments of {a.chunk} points per GPU, building.yaml shapes, random-init weights,"
                               f" gate_scale={a.gate_scale}" + (", ROUTING OVERRIDDEN: point i -> expert i mod E" if a.routing == "balanced" else "")
                               + (f", + {a.fine} fine samples (hierarchical)" if a.fine else "")
                               + (", mip recipe (two levels)" if a.mip else "")
                               + ((", INFERENCE ONLY (forward without saves; no backward / Adam" + ("; replayed from a hipGraph)" if a.graph != "off" else "; eager launches)")) if a.eval else "")
                               + (", hash-grid input encoding (16 levels x 2^19 x 2, table scaled to U(-1,1))" if a.hash else "")
                               + (f", capacity_factor {a.capacity_factor}" if a.capacity_factor != 1.0 else "")
                               + (f", + dense background model on {st['ctx']['Nb']} of {n_rays} rays x {a.samples // 2} samples" if a.bg else "")
                               + (f", model_dim {a.model_dim}, {a.experts} experts" if (a.model_dim != 256 or a.experts != 8) else ""),
                   "global_batch_rays": gb, "rays_per_gpu": n_rays, "samples": a.samples, "segment_points": a.chunk,
                   "kernel_set": None if a.dense else model.kernel_set(), "csrc_sha256": csrc_sha,
                   "parallelism": f"{a.parallelism}{world}" + ("-loopback" if a.loopback else ""), "balanced_value": None if balanced is None else balanced["value"],
                   "kept_token_fraction": round(kept / P, 4), "kept_token_fraction_mean": round(kept_mean / P, 4), "loss": round(loss_main, 6), "eager_ms_per_step": None if eager_ms is None else round(eager_ms, 3),
                   "runner_loop_ms_per_step": None if runner_ms is None else round(runner_ms, 3),
                   "timed_region": ((("forward + backward replayed from a hipGraph, all-reduce + Adam eager" if use_graph else "eager launches")
                                     + f"; eager_ms_per_step = the same step launched eagerly ({ev_steps} steps); runner_loop_ms_per_step = the "
                                     "reference's loop (render_rays under autograd + loss.backward() + torch.optim.Adam + ExponentialLR) with the "
                                     "forward and backward graphs of nerf.graph_train; `kernels`: every expert kernel relaunched 5 x back to back "
                                     "on one step's live buffers between two HIP events on the launch stream (independent of the host's launch rate)")
                                    if not a.no_events else "no per-kernel timing")},
        "roofline": roof, "kernels": detail, "balanced": balanced,
    }
    if ep_info is not None:
        out["config"]["expert_parallel"] = ep_info
    if ar_info is not None:
        out["config"]["allreduce"] = ar_info
        out["allreduce_ms"], out["allreduce_hidden_fraction"] = ar_info["allreduce_ms"], ar_info["allreduce_hidden_fraction"]

    # ---- N > 1, data parallel (the driver's scaling run): the expert-parallel step of the SAME batch right behind the headline, in the
    #      same line (north_star: RCCL all-to-all over xGMI in place of Tutel's; BASELINE configs[2]) - experts sharded E / N per GPU, kept
    #      rows exchanged per routing segment on a side stream, eager launches (the kept-rows mode reads split sizes on the host).
    #      A watchdog prints the line without it if a collective never returns: the headline must not depend on the probe.
    if multi and a.parallelism == "dp" and plain and not other and not a.no_ep_probe and model.E % world == 0:
        import threading

        probe_done = threading.Event()

        def give_up():
            if probe_done.is_set():
                return
            if rank == 0:
                for key in ("expert_parallel", "expert_parallel_owner_tail"):      # (a probe that finished keeps its record)
                    out["config"].setdefault(key, dict(error=f"the expert-parallel probes did not finish within {a.ep_probe_limit:.0f} s"))
                out["cpu_baseline"] = None
                print(json.dumps(out), flush=True)
            os._exit(0)
        dog = threading.Timer(a.ep_probe_limit + (0.0 if rank == 0 else 5.0), give_up)
        dog.daemon = True
        dog.start()
        def ep_probe(key, **ep_kw):
            try:
                from switch_nerf_amd.parallel import ExpertParallel
                graphed[0] = None
                model.set_expert_parallel(ExpertParallel(rank, world, model.E, padded=False, loopback=a.loopback, **ep_kw))
                reset_state(4321)
                for _ in range(3):
                    step()
                reset_state(1234)
                esteps = max(3, min(10, a.steps))
                edt_, est_ = timed(esteps, False)
                x = ep_report()
                x.update(steps=esteps, ms_per_step=round(edt_ / esteps * 1e3, 3), value=round(n_rays * world * esteps / edt_, 1), unit="rays/s",
                         launches="eager (compare with config.eager_ms_per_step of the data-parallel step)", loss=round(float(est_["loss"].item()), 6))
                out["config"][key] = x
            except Exception as e:      # (the same exception on every rank: a one-sided failure ends in the watchdog)
                out["config"][key] = dict(error=f"{type(e).__name__}: {e}"[:400])
        ep_probe("expert_parallel")                                      # kept rows exchanged per segment, side-stream overlap (the reference's placement)
        ep_probe("expert_parallel_owner_tail", owner_tail=True)          # the dense tail on the expert's rank: 0.53 x the bytes, fused launches (ep_owner.py)
        probe_done.set()
        dog.cancel()
    if rank == 0:
        if world == 1 and not a.no_cpu_baseline and not other and not a.loopback:
            try:
                out["cpu_baseline"] = cpu_baseline()
            except Exception as e:      # the baseline must never take the bench line down
                out["cpu_baseline"] = dict(value=None, unit="rays/s", cores=_host_cores(), threads=torch.get_num_threads(), kind="port",
                                           sample=f"failed: {e}")
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if dist:
        # (the ONE line is out: whatever the communicator's teardown prints - RCCL writes its version banner to stdout - goes to stderr)
        sys.stdout.flush()
        os.dup2(2, 1)
        graphed[0] = None
        from switch_nerf_amd import parallel as _par
        _par.shutdown()


def self_launch(n: int):
    """`python bench.py --gpus N` without torchrun's environment: become the launcher of N ranks on this node (one per GPU, RCCL)."""
    import socket
    have = torch.cuda.device_count()
    if have < n and "SWN_FORCE_DEVICE" not in os.environ:
        raise SystemExit(f"bench.py: --gpus {n} but only {have} GPU(s) are visible on this node")
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    os.execv(sys.executable, cmd)


if __name__ == "__main__":
    main()
