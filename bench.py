#!/usr/bin/env python
"""bench.py - train rays/s of the Switch-NeRF hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One step = one full training step of BASELINE.json configs[1] on every rank: 8192 synthetic rays x 256 samples
(16 routing segments of 131072 points), 8 experts top-1, capacity_factor 1.0, batch-prioritised routing, bf16 compute,
stratified perturbation + sigma noise drawn inside the timed region, forward + loss + backward + gradient
all-reduce (N > 1) + Adam.  Rays are independent, so ranks are data-parallel replicas with per-rank work fixed
(scaling = "weak"); the only collective is the RCCL all-reduce of the flat fp32 gradient buffer.

Prints ONE JSON line (rank 0).  `roofline` describes the dominant kernel (timed live with HIP events on the launch
stream); `cpu_baseline` is the CPU oracle (a port of the reference's CPU path) timed on this box's host cores on a
bounded sample (one 131072-point segment = 512 rays).
"""
import argparse
import json
import os
import sys
import time

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
    return dict(value=n_rays / t_best, unit="rays/s", cores=torch.get_num_threads(), kind="port",
                sample=f"oracle fwd+bwd fp32 on 1 of 16 segments (512 rays x 256 samples = 131072 points), best of {reps}; "
                       f"{t_best:.2f} s/segment => {16 * t_best:.1f} s per 8192-ray step")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rays", type=int, default=8192)
    ap.add_argument("--samples", type=int, default=256)
    ap.add_argument("--chunk", type=int, default=131072)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--gate-scale", type=float, default=1.0, help="scale of the router weight (small => balanced routing)")
    ap.add_argument("--parallelism", default="dp", choices=["dp", "ep"],
                    help="dp (default): experts replicated, one gradient all-reduce per step; ep: experts sharded over the ranks, "
                         "dispatched rows exchanged with RCCL all-to-all (BASELINE.json configs[2]; needs gpus | 8)")
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
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-events", action="store_true", help="do not record per-kernel HIP events in the timed region")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("SWN_FORCE_DEVICE", os.environ.get("LOCAL_RANK", 0)))     # (override: several ranks on one GPU, for testing)
    assert world == a.gpus or world == 1, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        from switch_nerf_amd import parallel
        parallel.init_from_env(os.environ.get("SWN_DIST_BACKEND", "nccl"), dev)       # "nccl" = RCCL; gloo only for single-GPU tests of this path

    from switch_nerf_amd.model import SwitchNeRF, BUILDING
    dtype = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    cfg = dict(BUILDING, model_dim=a.model_dim, gate_hidden=a.model_dim, num_experts=a.experts)
    if a.hash:
        cfg["hash"] = dict(n_levels=16, log2_table=19, base_res=16, per_level_scale=1.3819, aabb_lo=(-1.2, -1.2, -1.2), aabb_hi=(1.2, 1.2, 1.2))
    other = a.fine or a.mip or a.model_dim != 256 or a.experts != 8 or a.dense or a.bg or a.hash or a.capacity_factor != 1.0 or a.eval
    if a.dense:
        from switch_nerf_amd.dense import DenseNeRF
        model = DenseNeRF(dtype=dtype, device=dev, seed=0)
    else:
        model = SwitchNeRF(cfg, dtype=dtype, device=dev, seed=0, capacity_factor=a.capacity_factor)
    if a.hash:      # emulate a trained encoding (features O(1)): the standard U(-1e-4, 1e-4) initialisation gives every point the
        model.p["hash.table"].mul_(1e4)      # same first-layer output and the random-init router sends everything to one expert
    if a.gate_scale != 1.0 and not a.dense:
        model.p["wg"].mul_(a.gate_scale)
    rays, idx, rgbs = synth_batch(a.rays, 1000 + rank, dev)
    P = a.rays * a.samples

    if world > 1:
        allreduce = parallel.make_grad_allreduce()     # RCCL all-reduce over xGMI, one 16 MB bucket
    if a.parallelism == "ep":
        from switch_nerf_amd.parallel import ExpertParallel
        model.set_expert_parallel(ExpertParallel(rank, world, model.E))

    radii = torch.full((a.rays, 1), 1e-3, device=dev)
    scene = None
    if a.bg:
        from switch_nerf_amd.background import BackgroundScene
        from switch_nerf_amd.dense import DenseNeRF, DENSE
        bg = DenseNeRF(dict(DENSE, xyz_dim=4), dtype=dtype, device=dev, seed=1)
        scene = BackgroundScene(model, bg, [0.02, -0.03, 0.01], [0.6, 0.8, 0.7])
        rays[:, 7] = torch.rand(a.rays, device=dev) * 1.2 + 0.3          # about half of the rays leave the bound

    def step():
        if a.eval:       # render_rays in eval mode (runner.py:2835-2885 render_image's inner call): forward only
            with torch.no_grad():
                c = model.forward_rays(rays, idx, a.samples, a.chunk, 0.0, None, None, training=False)
            return dict(ctx=c, loss=c["rgb"].sum() * 0)
        pr = torch.rand(a.rays, a.samples, device=dev)              # rendering.py:582 rand_like
        ar = allreduce if world > 1 else None
        if a.mip:      # rendering_mip recipe: a.samples edges -> a.samples - 1 frustums per level, coarse + fine level
            nf = a.rays * (a.samples - 1)
            return model.train_step_mip(rgbs, rays, radii, idx, a.samples, a.samples, a.chunk, perturb=1.0, perturb_rand=pr,
                                        sigma_noise=torch.randn(nf, device=dev), sigma_noise_fine=torch.randn(nf, device=dev),
                                        grad_allreduce=ar)
        noise = torch.randn(P, device=dev)                          # rendering.py:366, sigma_noise_std = 1
        if scene is not None:
            return scene.train_step(rgbs, rays, idx, a.samples, a.chunk, perturb=1.0, perturb_rand=pr, sigma_noise=noise,
                                    sigma_noise_bg="randn", sigma_noise_bg_fine="randn", noise_std=1.0, grad_allreduce=ar,
                                    fine_samples=a.fine,
                                    sigma_noise_fine=torch.randn(a.rays * a.fine, device=dev) if a.fine else None)
        kw = {}
        if a.fine > 0:
            kw = dict(fine_samples=a.fine, sigma_noise_fine=torch.randn(a.rays * a.fine, device=dev))
        return model.train_step(rgbs, rays, idx, a.samples, a.chunk, perturb=1.0, perturb_rand=pr, sigma_noise=noise,
                                grad_allreduce=ar, **kw)

    for _ in range(a.warmup):
        st = step()
    model.profile = not a.no_events
    model.events = {}
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        st = step()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    ms = dt / a.steps * 1e3
    value = a.rays * world * a.steps / dt

    # ---- per-kernel accounting from the live HIP events
    c = st["ctx"]["c"] if a.bg else st["ctx"]
    kept = P if a.dense else int(torch.minimum(c["counts"], torch.tensor(c["cap"], device=dev)).sum().item())
    L, M, E = model.L, model.M, model.E
    esz = 2 if dtype == torch.bfloat16 else 4
    kern = {}
    for name, evs in ({} if other else model.events).items():       # (other recipes: headline number only)
        kern[name] = sum(x.elapsed_time(y) for x, y in evs) / len(evs)          # ms per step
    flops_chain = 2.0 * L * M * M * kept                                       # expert fwd == bwd-data == wgrad flops
    # algorithmic HBM bytes per launch: fwd reads x, writes L-1 activations + output; bwd reads dout and the
    # skip layer's dZ, writes L-1 dZ + dx; the weight gradients read L layer inputs and L dZ (ReLU masks: 1/16 of a tensor each)
    bytes_fwd = kept * M * esz * (1 + (L - 1) + 1)
    bytes_bwd = kept * M * esz * (1 + (L - 1) + 1 + 1)
    bytes_wgrad = kept * M * esz * 2 * L
    roof = None
    detail = {}
    for name, fl, by in (("expert_fwd", flops_chain, bytes_fwd), ("expert_bwd", flops_chain, bytes_bwd),
                         ("expert_wgrad", flops_chain, bytes_wgrad)):
        if name in kern and kern[name] > 0:
            tf = fl / (kern[name] * 1e-3) / 1e12
            gbs = by / (kern[name] * 1e-3) / 1e9
            detail[name] = dict(ms=round(kern[name], 4), tflops=round(tf, 1), mfma_frac=round(tf / MFMA_BF16_PEAK_TFLOPS, 4),
                                alg_gbs=round(gbs, 1), hbm_frac=round(gbs / HBM_PEAK_GBS, 4))
    if detail:
        dom = max(detail, key=lambda k: detail[k]["ms"])
        d = detail[dom]
        traffic = None
        tp = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tp):
            try:
                traffic = json.load(open(tp)).get(dom)
            except Exception:
                traffic = None
        names = {"expert_fwd": "chain_kernel<bf16,1> (expert forward, 7 fused layers)",
                 "expert_bwd": "chain_kernel<bf16,2> (expert backward-data, 7 fused layers)",
                 "expert_wgrad": "wgrad_kernel<bf16,1> (expert weight gradients, 7 layers in one launch)"}
        # the binding roofline is the one the kernel is closest to: in training the expert
        # chains must save every activation for the weight-gradient GEMM, which makes them HBM-leaning
        if d["hbm_frac"] >= d["mfma_frac"]:
            roof = dict(kernel=names[dom], bound="hbm", achieved=d["alg_gbs"], peak=HBM_PEAK_GBS, unit="GB/s",
                        frac=d["hbm_frac"], traffic=traffic, mfma_tflops=d["tflops"], mfma_frac=d["mfma_frac"])
        else:
            roof = dict(kernel=names[dom], bound="mfma", achieved=d["tflops"], peak=MFMA_BF16_PEAK_TFLOPS, unit="TFLOP/s",
                        frac=d["mfma_frac"], traffic=traffic, alg_gbs=d["alg_gbs"], hbm_frac=d["hbm_frac"])

    out = {
        "metric": "train rays/sec (8192-ray batch, 256 samples, 8 experts)", "value": round(value, 1), "unit": "rays/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
        "config": {"workload": ("other recipe (informational): " if other else "configs[1]: ")
                               + ("dense NeRF 8 x 256 (configs[0] network, --no-use_moe)" if a.dense else f"{a.experts}-expert top-1 expertmlp, capacity_factor=1.0, BPR")
                               + f", {a.rays} rays x {a.samples} samples"
                               f" per GPU, {P // a.chunk} segments of {a.chunk} points, building.yaml shapes, random-init weights,"
                               f" gate_scale={a.gate_scale}" + (f", + {a.fine} fine samples (hierarchical)" if a.fine else "")
                               + (", mip recipe (two levels)" if a.mip else "")
                               + (", INFERENCE ONLY (forward without saves; no backward / Adam)" if a.eval else "")
                               + (", hash-grid input encoding (16 levels x 2^19 x 2, table scaled to U(-1,1))" if a.hash else "")
                               + (f", capacity_factor {a.capacity_factor}" if a.capacity_factor != 1.0 else "")
                               + (f", + dense background model on {st['ctx']['Nb']} of {a.rays} rays x {a.samples // 2} samples" if a.bg else "")
                               + (f", model_dim {a.model_dim}, {a.experts} experts" if (a.model_dim != 256 or a.experts != 8) else ""),
                   "rays_per_gpu": a.rays, "samples": a.samples, "segment_points": a.chunk, "parallelism": f"{a.parallelism}{world}",
                   "kept_token_fraction": round(kept / P, 4), "loss": round(float(st["loss"].item()), 6)},
        "roofline": roof, "kernels": detail,
    }
    if rank == 0:
        if world == 1 and not a.no_cpu_baseline and not other:
            try:
                out["cpu_baseline"] = cpu_baseline()
            except Exception as e:      # the baseline must never take the bench line down
                out["cpu_baseline"] = dict(value=None, unit="rays/s", cores=torch.get_num_threads(), kind="port", sample=f"failed: {e}")
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
