"""torch.autograd bridge: lets the reference's training loop drive the HIP path unchanged.

Runner.train does, per step (runner.py:604-693):
    metrics = self._training_step(rgbs, rays, image_indices)          # calls render_rays(nerf, ...)
    scaler.scale(metrics['loss']).backward()
    scaler.step(optimizer); scaler.update(); scheduler.step()          # torch.optim.Adam over nerf.parameters(), ExponentialLR
so render_rays must return tensors with a grad_fn and `nerf.parameters()` must be real leaf Parameters that receive `.grad`.
Here the model's whole parameter set is ONE leaf Parameter (`nerf.flat_param`, sharing storage with the flat fp32 master buffer
the kernels read - Adam is elementwise, one flat tensor is equivalent to the reference's per-module tensors), and the rendering of a
ray batch is one autograd node whose backward runs the HIP backward kernels and hands the flat gradient to autograd (which
accumulates it into `.grad`: gradient accumulation over micro-batches, loss scaling and torch's finite checks all work as usual).
"""
from __future__ import annotations

import torch

from . import ops


class RenderRaysFunction(torch.autograd.Function):
    """(flat_param, nerf, rays, image_indices, S, F, chunk, perturb, perturb_rand, sigma_noise, sigma_noise_fine)
    -> (rgb [N,3], gate_loss_coarse [n_seg], gate_loss_fine [n_seg_f] or empty, depth [N], depth_variance [N]).
    perturb_rand = "graph" (with sigma_noise = the noise std): the forward and the backward are replayed from the captured graphs of
    graph.GraphedRenderTrain (`nerf.graph_train = True`), which draw the jitter and the noise on the device themselves."""

    @staticmethod
    def forward(ctx, flat_param, nerf, rays, image_indices, S, F, chunk, perturb, perturb_rand, sigma_noise, sigma_noise_fine):
        nerf._sync_compute_copies()
        ctx.nerf, ctx.fine = nerf, F > 0
        ctx.graph = None
        if isinstance(perturb_rand, str) and perturb_rand == "graph":
            from .graph import GraphedRenderTrain, cached_graph
            cache = nerf.__dict__.setdefault("_train_graphs", {})
            key = (rays.shape[0], S, F, int(chunk), float(perturb), float(sigma_noise or 0.0), bool(nerf.moe_no_batch), nerf.dtype)
            g = cached_graph(cache, key, lambda: GraphedRenderTrain(nerf, rays, image_indices, S, F, chunk, float(perturb),
                                                                    float(sigma_noise or 0.0)))
            state, outs = g.forward(rays, image_indices)
            ctx.graph, ctx.state, ctx.generation = g, state, g.generation
            res = (outs[0].clone(), outs[1].clone(), outs[2].clone(), outs[3].clone(), outs[4].clone())
        elif F > 0:
            c, cf, out = nerf.forward_hier(rays, image_indices, S, F, chunk, perturb, perturb_rand, None, sigma_noise, sigma_noise_fine,
                                           no_batch=nerf.moe_no_batch, training=True)
            ctx.state = (c, cf, out, S, F)
            res = (out["rgb"], c["l_aux"], cf["l_aux"], out["depth"], out["depth_variance"])
        else:
            c = nerf.forward_rays(rays, image_indices, S, chunk, perturb, perturb_rand, sigma_noise, training=True, no_batch=nerf.moe_no_batch)
            ctx.state = (c,)
            res = (c["rgb"], c["l_aux"], torch.zeros(0, device=rays.device), c["depth"], c["depth_variance"])
        ctx.mark_non_differentiable(res[3], res[4])
        nerf._last_ctx = ctx.state
        return res

    @staticmethod
    def backward(ctx, d_rgb, d_laux_c, d_laux_f, _d_depth, _d_var):
        nerf = ctx.nerf
        if ctx.graph is not None:
            # every replay of the forward graph returns the SAME state object (static buffers): the replay counter tells whether the
            # buffers still hold this node's activations
            if ctx.graph.generation != ctx.generation or nerf._last_ctx is not ctx.state:
                raise RuntimeError("graph_train: backward() must follow the forward it belongs to (the captured graphs share one set of "
                                   "static activation buffers per batch shape)")
            g = ctx.graph.backward(d_rgb.to(torch.float32), d_laux_c.to(torch.float32), d_laux_f.to(torch.float32))
            ctx.state = None
            return (g.clone(),) + (None,) * 10
        nerf.grad.zero_()
        d_rgb = d_rgb.to(torch.float32).contiguous()
        if ctx.fine:
            c, cf, out, S, F = ctx.state
            d_raw_m = ops.composite_bwd(out["raw"], out["z"], d_rgb)
            d_raw_f, d_raw_c = ops.unmerge_grad(d_raw_m, out["order"], F, S)
            nerf.backward_net(cf, d_raw_f, d_laux_f.to(torch.float32).contiguous())
            nerf.backward_net(c, d_raw_c, d_laux_c.to(torch.float32).contiguous())
        else:
            (c,) = ctx.state
            nerf.backward(c, d_rgb, d_laux_c.to(torch.float32).contiguous())
        ctx.state = None
        return (nerf.grad.clone(),) + (None,) * 10
