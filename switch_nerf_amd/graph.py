"""hipGraph capture of the training step's launch sequence.

A train step of the hot path is ~150 kernel launches.  At the reference's multi-GPU batch (8192 rays split over 8 ranks = 1024 rays
per GPU, runner.py:573-575) they hold ~2.5 ms of GPU work while the Python host needs ~3.7 ms to enqueue them: the step is
launch-bound.  Nothing in forward + loss + backward depends on host state that changes between steps (shapes, capacities and
buffer addresses are static; the random draws are device-side), so the sequence is captured once into a hipGraph and replayed:
one host call per step, kernels back to back.  The gradient all-reduce (RCCL) and Adam (its bias correction takes the step count
as a launch argument) stay outside the graph; the refresh of the compute copies of the weights is a second small graph.
"""
from __future__ import annotations

import torch

MAX_CACHED_GRAPHS = 8


def cached_graph(cache: dict, key, make):
    """cache[key], created with make() on a miss; the cache holds at most MAX_CACHED_GRAPHS captured graphs (least recently used one
    dropped: a graph owns a private memory pool with every activation of its batch shape).  Dropping a graph is safe - the buffers a
    capture bakes in besides its own pool (the model's cached buffers, the library workspaces of ops.py) are never freed."""
    g = cache.pop(key, None)
    if g is None:
        while len(cache) >= MAX_CACHED_GRAPHS:
            cache.pop(next(iter(cache)))
        g = make()
    cache[key] = g              # (re-inserted: dict order = recency)
    return g


class GraphedTrainStep:
    """step = GraphedTrainStep(model, rgbs, rays, image_indices, n_samples, seg_tokens, ...); res = step(rgbs, rays, image_indices)

    Same result dict as SwitchNeRF.train_step (tensors live in static graph memory: read them before the next call).  The inputs
    are copied into static buffers; stratified jitter (perturb > 0) and the sigma noise (noise_std > 0) are drawn inside the
    graph from the device generator, like rendering.py:582 / :366.  Only the plain (non-hierarchical, non-mip) step is graphed.

    split_backward (default: on when torch.distributed runs more than one rank): the step is captured as TWO graphs cut behind the
    expert weight gradients (SwitchNeRF.backward_net_a / _b).  Between the replays the all-reduce of the expert block of the flat
    gradient - final at that point, 14.7 of 15.8 MB - is issued on the model's side stream and travels while the second graph (router
    backward, front backward chain, dense weight gradients: a quarter of the step) runs; the dense prefix follows behind it.  This is
    the overlap DDP's buckets give the reference (runner.py:203-207); with one all-reduce behind the whole backward nothing hides it
    at 1024 rays per GPU.  Results are bit-identical to the single-graph step (same sums over the same ranks)."""

    def __init__(self, model, rgbs, rays, image_indices, n_samples: int, seg_tokens: int, perturb: float = 1.0, noise_std: float = 1.0,
                 routing_override=None, warmup: int = 2, split_backward=None):
        N, S = rays.shape[0], int(n_samples)
        if model.ep is not None:
            seg_payload = model.E * int(model.cf * ((min(int(seg_tokens), N * S) + model.E - 1) // model.E)) * model.M * (4 if model.dtype == torch.float32 else 2)
            if not (model.ep.capturable and model.ep.use_padded(seg_payload)):
                raise ValueError("GraphedTrainStep: the expert-parallel step with unequal splits reads their sizes on the host and cannot be "
                                 "captured; use SwitchNeRF.train_step, or ExpertParallel(..., padded=True) (equal, capacity-padded splits)")
        self.model = model
        dev = model.dev
        self.rgbs, self.rays, self.idx = rgbs.clone(), rays.clone(), image_indices.clone()
        P = N * S
        if split_backward is None:
            import torch.distributed as dist
            split_backward = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        self.split = bool(split_backward) and model.ep is None
        if model.loss_scaler is not None:
            model._loss_scale_tensor()                     # exists before the capture (it is read, not created, inside the graph)
        ro = None if routing_override is None else routing_override.to(dev).int().contiguous()

        def run():
            pr = torch.rand(N, S, device=dev) if perturb > 0 else None
            noise = torch.randn(P, device=dev) * noise_std if noise_std > 0 else None
            return model.grad_step(self.rgbs, self.rays, self.idx, S, min(int(seg_tokens), P), perturb=perturb, perturb_rand=pr,
                                   sigma_noise=noise, routing_override=ro, split=self.split)

        was_profile, model.profile = model.profile, False
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                      # warm-up on a side stream: allocates every cached buffer / workspace
            for _ in range(max(1, warmup)):
                r_ = run()
                if self.split:
                    model.backward_net_b(r_["bwd_b"])
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, stream=side):
            self.res = run()
        self.graph_b = None
        if self.split:
            self.graph_b = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph_b, stream=side, pool=self.graph.pool()):
                model.backward_net_b(self.res["bwd_b"])
        self.copies = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.copies, stream=side):
            model.refresh_compute_copies()
        model.profile = was_profile

    def __call__(self, rgbs=None, rays=None, image_indices=None, grad_allreduce=None, optimizer_step=True):
        m = self.model
        if rgbs is not None:
            self.rgbs.copy_(rgbs)
        if rays is not None:
            self.rays.copy_(rays)
        if image_indices is not None:
            self.idx.copy_(image_indices)
        self.graph.replay()
        ar = grad_allreduce
        if self.graph_b is not None:
            if ar is not None and hasattr(ar, "begin"):
                ar.begin(m.grad[m.n_dense:], m.side)       # the expert block is final: it travels under the second backward graph
                ar = lambda _view, f=grad_allreduce: f.finish(m.grad[: m.n_dense])
            self.graph_b.replay()
        # all-reduce, loss-scale handling (fp16: inf check, skipped step, scale update - the captured step reads the scale from a
        # device scalar), Adam and the refresh of the compute copies (a second small graph): SwitchNeRF.apply_step
        m.apply_step(ar, optimizer_step, refresh=self.copies.replay)
        return self.res


class GraphedRender:
    """The inference forward of ONE ray-batch shape (render_rays in eval mode: no jitter, no noise, no activation saves), captured
    once and replayed: Runner.render_image (runner.py:2835-2885) calls render_rays for every image_pixel_batch_size rays of every
    image with the same shapes - ~60 launches per call that the host otherwise enqueues slower than the GPU runs them.

        g = GraphedRender(model, rays, image_indices, n_samples, seg_tokens, fine_samples=F, no_batch=model.moe_no_batch)
        out = g(rays, image_indices)     # dict: rgb, depth, depth_variance, l_aux_coarse, idx_coarse, sigma_coarse (+ *_fine)

    The returned tensors live in the graph's static memory: consume (or clone) them before the next call.  Expert-parallel
    evaluation reads split sizes on the host (list_all_to_all) and cannot be captured."""

    def __init__(self, model, rays, image_indices, n_samples: int, seg_tokens: int, fine_samples: int = 0, no_batch: bool = False,
                 warmup: int = 2):
        if model.ep is not None and not model.ep.local:
            raise ValueError("GraphedRender: expert-parallel evaluation exchanges host-sized splits and cannot be captured")
        self.model = model
        dev = model.dev
        self.rays, self.idx = rays.clone().contiguous(), image_indices.clone().contiguous()
        N, S, F = rays.shape[0], int(n_samples), int(fine_samples)
        self.key = (N, S, F, int(seg_tokens), bool(no_batch))

        def run():
            with torch.no_grad():
                chunk = min(int(seg_tokens), N * S)
                if F > 0:
                    c, cf, o = model.forward_hier(self.rays, self.idx, S, F, chunk, 0.0, None, None, None, None, no_batch=no_batch,
                                                  training=False)
                    return dict(rgb=o["rgb"], depth=o["depth"], depth_variance=o["depth_variance"], l_aux_coarse=c["l_aux"],
                                l_aux_fine=cf["l_aux"], idx_coarse=c["idx"], idx_fine=cf["idx"], sigma_coarse=c["raw"][:, 3],
                                sigma_fine=cf["raw"][:, 3])
                c = model.forward_rays(self.rays, self.idx, S, chunk, 0.0, None, None, training=False, no_batch=no_batch)
                return dict(rgb=c["rgb"], depth=c["depth"], depth_variance=c["depth_variance"], l_aux_coarse=c["l_aux"],
                            idx_coarse=c["idx"], sigma_coarse=c["raw"][:, 3])

        was_profile, model.profile = model.profile, False
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                      # warm-up on the capture stream: allocates every cached buffer / workspace
            for _ in range(max(1, warmup)):
                run()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, stream=side):
            self.out = run()
        model.profile = was_profile

    def __call__(self, rays=None, image_indices=None):
        if rays is not None:
            self.rays.copy_(rays)
        if image_indices is not None:
            self.idx.copy_(image_indices)
        self.graph.replay()
        return self.out


class GraphedRenderTrain:
    """The TRAINING render of one ray-batch shape as two captured graphs - forward and backward - behind autograd.RenderRaysFunction,
    so that the reference's own loop (runner.py:604-693: `render_rays` under autograd, `scaler.scale(loss).backward()`, torch.optim.Adam,
    ExponentialLR) runs the hot path with two host calls per step instead of ~150 launches:

        forward graph : [jitter / sigma noise drawn on the device] -> forward_rays / forward_hier (training=True, activations saved
                        into static buffers) -> rgb, gate losses, depth, depth variance
        backward graph: static d_rgb, d_gate_loss -> grad.zero_() + compositing backward + backward_net -> the flat gradient

    Enabled per model with `nerf.graph_train = True` (rendering.render_rays then takes this path in training mode; one instance is
    cached per batch shape).  The loss itself, the optimizer and the scheduler stay ordinary torch code between the two replays.
    Not for expert-parallel models (their step issues collectives between its kernels)."""

    def __init__(self, model, rays, image_indices, n_samples: int, fine_samples: int, seg_tokens: int, perturb: float, noise_std: float,
                 warmup: int = 2):
        if model.ep is not None and not model.ep.local:
            raise ValueError("GraphedRenderTrain: expert-parallel training is not captured")
        from . import ops
        self.model = model
        dev = model.dev
        N, S, F = rays.shape[0], int(n_samples), int(fine_samples)
        self.rays, self.idx = rays.clone().contiguous(), image_indices.clone().contiguous()
        self.d_rgb = torch.zeros(N, 3, dtype=torch.float32, device=dev)
        chunk = min(int(seg_tokens), N * S)

        def fwd():
            pr = torch.rand(N, S, device=dev) if perturb > 0 else None                                   # rendering.py:582
            noise = torch.randn(N * S, device=dev) * noise_std if noise_std > 0 else None                # rendering.py:366
            if F > 0:
                noise_f = torch.randn(N * F, device=dev) * noise_std if noise_std > 0 else None
                c, cf, out = model.forward_hier(self.rays, self.idx, S, F, chunk, perturb, pr, None, noise, noise_f,
                                                no_batch=model.moe_no_batch, training=True)
                return (c, cf, out), (out["rgb"], c["l_aux"], cf["l_aux"], out["depth"], out["depth_variance"])
            c = model.forward_rays(self.rays, self.idx, S, chunk, perturb, pr, noise, training=True, no_batch=model.moe_no_batch)
            return (c,), (c["rgb"], c["l_aux"], torch.zeros(0, device=dev), c["depth"], c["depth_variance"])

        def bwd(state):
            model.grad.zero_()
            if F > 0:
                c, cf, out = state
                d_raw_m = ops.composite_bwd(out["raw"], out["z"], self.d_rgb)
                d_raw_f, d_raw_c = ops.unmerge_grad(d_raw_m, out["order"], F, S)
                model.backward_net(cf, d_raw_f, self.d_laux_f)
                model.backward_net(c, d_raw_c, self.d_laux_c)
            else:
                model.backward(state[0], self.d_rgb, self.d_laux_c)

        was_profile, model.profile = model.profile, False
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                      # warm-up on the capture stream: allocates every cached buffer / workspace
            for _ in range(max(1, warmup)):
                st, outs = fwd()
                self.d_laux_c = torch.zeros_like(outs[1])
                self.d_laux_f = torch.zeros_like(outs[2])
                bwd(st)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.fwd_graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.fwd_graph, stream=side):
            self.state, self.outs = fwd()
        self.bwd_graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.bwd_graph, stream=side, pool=self.fwd_graph.pool()):
            bwd(self.state)
        model.profile = was_profile

    generation = 0           # forward replays so far: the static activation buffers belong to the LAST one (autograd.py checks it)

    def forward(self, rays, image_indices):
        self.rays.copy_(rays)
        self.idx.copy_(image_indices)
        self.fwd_graph.replay()
        self.generation += 1
        return self.state, self.outs

    def backward(self, d_rgb, d_laux_c, d_laux_f):
        self.d_rgb.copy_(d_rgb)
        self.d_laux_c.copy_(d_laux_c)
        if self.d_laux_f.numel():
            self.d_laux_f.copy_(d_laux_f)
        self.bwd_graph.replay()
        return self.model.grad
