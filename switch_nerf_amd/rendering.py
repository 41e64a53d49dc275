"""Mirror of the reference's rendering.render_rays (/root/reference/switch_nerf/rendering.py:15-196): SwitchNeRF (or
DenseNeRF) model, optional dense background model behind the foreground bound (background.py), no cascade;
fine_samples = 0 (coarse pass composited) or fine_samples > 0 (hierarchical: coarse weights -> importance samples -> fine
pass -> merged compositing).

    results, bg_nerf_rays_present = render_rays(nerf, None, rays, image_indices, hparams, None, None,
                                                get_depth, get_depth_variance, get_bg_fg_rgb)

Result keys follow the reference: rgb_coarse, depth_coarse, depth_variance_coarse, gate_loss_coarse
([chunks * moe layers], rendering.py:388-390), moe_gates_coarse [N, S, 1, 1], sigma_coarse (hparams.return_sigma);
with fine_samples > 0: rgb_fine, depth_fine, depth_variance_fine, gate_loss_fine, gate_loss_coarse (and no rgb_coarse,
exactly like the reference's composite_rgb=False coarse pass, rendering.py:227).
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch


def render_rays(nerf, bg_nerf, rays: torch.Tensor, image_indices: Optional[torch.Tensor], hparams, sphere_center=None,
                sphere_radius=None, get_depth: bool = True, get_depth_variance: bool = True,
                get_bg_fg_rgb: bool = False) -> Tuple[Dict[str, torch.Tensor], bool]:
    if getattr(hparams, "use_cascade", False):
        raise NotImplementedError("use_cascade is outside the hot path")
    F = int(getattr(hparams, "fine_samples", 0))
    N = rays.shape[0]
    S = hparams.coarse_samples
    if bg_nerf is not None:
        return _render_rays_bg(nerf, bg_nerf, rays, image_indices, hparams, sphere_center, sphere_radius, get_depth,
                               get_depth_variance, get_bg_fg_rgb)
    P = N * S
    chunk = min(hparams.model_chunk_size, P)
    # (a ragged last model chunk - P % chunk != 0 - is routed on its own like the reference's loop, rendering.py:354-383, in
    #  evaluation and in training)
    perturb = hparams.perturb if nerf.training else 0
    use_noise = bool(getattr(hparams, "use_sigma_noise", False) and hparams.sigma_noise_std > 0 and nerf.training)
    under_autograd = nerf.training and torch.is_grad_enabled() and hasattr(nerf, "flat_param") and getattr(nerf, "hash", None) is None
    if image_indices is None:
        image_indices = torch.zeros(N, dtype=torch.long, device=rays.device)
    if under_autograd and getattr(nerf, "graph_train", False) and getattr(nerf, "ep", None) is None:
        # forward / backward replayed from captured graphs (graph.GraphedRenderTrain); jitter and sigma noise are drawn inside them
        pr, noise, noise_f = "graph", (float(hparams.sigma_noise_std) if use_noise else 0.0), None
    else:
        pr = torch.rand(N, S, device=rays.device) if perturb > 0 else None
        noise = torch.randn(P, device=rays.device) * hparams.sigma_noise_std if use_noise else None      # rendering.py:366
        noise_f = torch.randn(N * F, device=rays.device) * hparams.sigma_noise_std if (F > 0 and noise is not None) else None
    if under_autograd:
        # training under autograd (the reference's Runner loop: loss.backward() + torch optimizer, runner.py:679-693): one autograd
        # node over the HIP forward; its backward runs the HIP backward and fills nerf.flat_param.grad (autograd.py)
        from .autograd import RenderRaysFunction
        rgb, gl_c, gl_f, depth, dvar = RenderRaysFunction.apply(nerf.flat_param, nerf, rays.contiguous(), image_indices, S, F, chunk,
                                                                float(perturb), pr, noise, noise_f)
        typ = "fine" if F > 0 else "coarse"
        res = {f"rgb_{typ}": rgb, "gate_loss_coarse": gl_c}
        if F > 0:
            res["gate_loss_fine"] = gl_f
        if get_depth:
            res[f"depth_{typ}"] = depth
        if get_depth_variance:
            res[f"depth_variance_{typ}"] = dvar
        st = nerf._last_ctx
        if getattr(hparams, "moe_return_gates", False):
            res["moe_gates_coarse"] = st[0]["idx"].long().view(N, S, 1, 1)
            if F > 0:
                res["moe_gates_fine"] = st[1]["idx"].long().view(N, F, 1, 1)
        if getattr(hparams, "return_sigma", False):
            res["sigma_coarse"] = st[0]["raw"][:, 3].view(N, S)
            if F > 0:
                res["sigma_fine"] = st[1]["raw"][:, 3].view(N, F)
        return res, False
    if not nerf.training and getattr(nerf, "graph_eval", False) and getattr(nerf, "ep", None) is None:
        return _render_rays_graphed(nerf, rays, image_indices, hparams, N, S, F, chunk, get_depth, get_depth_variance), False
    if F > 0:
        c, cf, out = nerf.forward_hier(rays.contiguous(), image_indices, S, F, chunk, float(perturb), pr, None, noise, noise_f,
                                       no_batch=nerf.moe_no_batch, training=nerf.training)
        res = {"rgb_fine": out["rgb"], "gate_loss_coarse": c["l_aux"], "gate_loss_fine": cf["l_aux"]}
        if get_depth:
            res["depth_fine"] = out["depth"]
        if get_depth_variance:
            res["depth_variance_fine"] = out["depth_variance"]
        if getattr(hparams, "moe_return_gates", False):
            res["moe_gates_coarse"] = c["idx"].long().view(N, S, 1, 1)
            res["moe_gates_fine"] = cf["idx"].long().view(N, F, 1, 1)
        if getattr(hparams, "return_sigma", False):
            res["sigma_coarse"] = c["raw"][:, 3].view(N, S)
            res["sigma_fine"] = cf["raw"][:, 3].view(N, F)
        return res, False
    c = nerf.forward_rays(rays.contiguous(), image_indices, S, chunk, float(perturb), pr, noise, training=nerf.training,
                          no_batch=nerf.moe_no_batch)
    res = {"rgb_coarse": c["rgb"], "gate_loss_coarse": c["l_aux"]}
    if get_depth:
        res["depth_coarse"] = c["depth"]
    if get_depth_variance:
        res["depth_variance_coarse"] = c["depth_variance"]
    if getattr(hparams, "moe_return_gates", False):
        res["moe_gates_coarse"] = c["idx"].long().view(N, S, 1, 1)
    if getattr(hparams, "return_sigma", False):
        res["sigma_coarse"] = c["raw"][:, 3].view(N, S)
    return res, False


def _render_rays_graphed(nerf, rays, image_indices, hparams, N, S, F, chunk, get_depth, get_depth_variance):
    """Evaluation with `nerf.graph_eval = True`: the forward of this batch shape is captured once (graph.GraphedRender, cached on the
    model per (rays, samples, fine samples, chunk, no_batch)) and replayed - Runner.render_image's pixel-batch loop
    (runner.py:2835-2885) is ~60 launches per call otherwise.  Results are cloned out of the graph's static memory."""
    from .graph import GraphedRender, cached_graph
    cache = nerf.__dict__.setdefault("_render_graphs", {})
    key = (N, S, F, int(chunk), bool(nerf.moe_no_batch), nerf.dtype)
    nerf._sync_compute_copies()
    g = cached_graph(cache, key, lambda: GraphedRender(nerf, rays.contiguous(), image_indices, S, chunk, F, nerf.moe_no_batch))
    o = g(rays, image_indices)
    typ = "fine" if F > 0 else "coarse"
    res = {f"rgb_{typ}": o["rgb"].clone(), "gate_loss_coarse": o["l_aux_coarse"].clone()}
    if F > 0:
        res["gate_loss_fine"] = o["l_aux_fine"].clone()
    if get_depth:
        res[f"depth_{typ}"] = o["depth"].clone()
    if get_depth_variance:
        res[f"depth_variance_{typ}"] = o["depth_variance"].clone()
    if getattr(hparams, "moe_return_gates", False):
        res["moe_gates_coarse"] = o["idx_coarse"].long().view(N, S, 1, 1)
        if F > 0:
            res["moe_gates_fine"] = o["idx_fine"].long().view(N, F, 1, 1)
    if getattr(hparams, "return_sigma", False):
        res["sigma_coarse"] = o["sigma_coarse"].reshape(N, S).clone()
        if F > 0:
            res["sigma_fine"] = o["sigma_fine"].reshape(N, F).clone()
    return res


def _render_rays_bg(nerf, bg_nerf, rays, image_indices, hparams, sphere_center, sphere_radius, get_depth, get_depth_variance,
                    get_bg_fg_rgb):
    """The bg_nerf branch (rendering.py:32-159) through background.BackgroundScene."""
    from .background import BackgroundScene
    N, S, F = rays.shape[0], hparams.coarse_samples, int(getattr(hparams, "fine_samples", 0))
    perturb = hparams.perturb if nerf.training else 0
    use_noise = getattr(hparams, "use_sigma_noise", False) and hparams.sigma_noise_std > 0 and nerf.training
    std = hparams.sigma_noise_std if use_noise else 0.0
    if image_indices is None:
        image_indices = torch.zeros(N, dtype=torch.long, device=rays.device)
    scene = getattr(nerf, "_bg_scene", None)
    if scene is None or scene.bg is not bg_nerf:
        scene = nerf._bg_scene = BackgroundScene(nerf, bg_nerf, sphere_center, sphere_radius)
    scene.center, scene.radius = sphere_center, sphere_radius
    kw = {}
    if use_noise:        # rendering.py:366: randn per evaluated chunk, for either model
        kw = dict(sigma_noise=torch.randn(N * S, device=rays.device) * std, sigma_noise_bg="randn", sigma_noise_bg_fine="randn",
                  sigma_noise_fine=torch.randn(N * F, device=rays.device) * std if F else None)
    ctx = scene.forward(rays.contiguous(), image_indices, S, min(hparams.model_chunk_size, N * S), float(perturb), fine_samples=F,
                        no_batch=nerf.moe_no_batch, noise_std=std, training=nerf.training, **kw)
    typ = "fine" if F > 0 else "coarse"
    res = {f"rgb_{typ}": ctx["rgb"], "gate_loss_coarse": ctx["c"]["l_aux"]}
    if F > 0:
        res["gate_loss_fine"] = ctx["cf"]["l_aux"]
    if get_depth:
        res[f"depth_{typ}"] = ctx["depth"]
    if get_depth_variance:
        res[f"depth_variance_{typ}"] = ctx["depth_variance"]
    if get_bg_fg_rgb:                                                    # :111-112, :124-125
        res[f"fg_rgb_{typ}"] = ctx["fg_rgb"]
        res[f"bg_rgb_{typ}"] = ctx["rgb"] - ctx["fg_rgb"]
        res[f"fg_depth_{typ}"] = ctx["fg_depth"]
        res[f"bg_depth_{typ}"] = ctx["depth"] - ctx["fg_depth"]
    if getattr(hparams, "moe_return_gates", False):
        res["moe_gates_coarse"] = ctx["c"]["idx"].long().view(N, S, 1, 1)
        if F > 0:
            res["moe_gates_fine"] = ctx["cf"]["idx"].long().view(N, F, 1, 1)
    return res, ctx["Nb"] > 0


def render_rays_mip(nerf, rays: torch.Tensor, radii: torch.Tensor, image_indices: Optional[torch.Tensor], hparams,
                    get_depth: bool = True, get_depth_variance: bool = True) -> Tuple[Dict[str, torch.Tensor], bool]:
    """Mirror of rendering_mip.render_rays (/root/reference/switch_nerf/rendering_mip.py:133-172) for MipNeRFMoE-style models:
    results carry rgb_coarse, gate_loss_coarse and, with fine_samples > 0, rgb_fine, depth_fine, depth_variance_fine,
    gate_loss_fine (the coarse level reports depth only when it is the last one, :207-208)."""
    N = rays.shape[0]
    S, F = hparams.coarse_samples, int(getattr(hparams, "fine_samples", 0))
    perturb = hparams.perturb if nerf.training else 0
    pr = torch.rand(N, S, device=rays.device) if perturb > 0 else None
    chunk = hparams.model_chunk_size
    noise = noise_f = None
    if getattr(hparams, "use_sigma_noise", False) and hparams.sigma_noise_std > 0 and nerf.training:
        noise = torch.randn(N * (S - 1), device=rays.device) * hparams.sigma_noise_std
        noise_f = torch.randn(N * max(F - 1, 0), device=rays.device) * hparams.sigma_noise_std if F > 0 else None
    if image_indices is None:
        image_indices = torch.zeros(N, dtype=torch.long, device=rays.device)
    c, cf = nerf.forward_mip(rays.contiguous(), radii, image_indices, S, F, chunk, float(perturb), pr, None, noise, noise_f,
                             no_batch=nerf.moe_no_batch, rgb_padding=float(getattr(hparams, "rgb_padding", 0.001) or 0.0),
                             resample_padding=float(getattr(hparams, "weights_resample_padding", 0.01)), training=nerf.training,
                             fine_randomized=bool(hparams.perturb))      # rendering_mip.py:227: randomized=hparams.perturb, also in eval
    res = {"rgb_coarse": c["rgb"], "gate_loss_coarse": c["l_aux"]}
    top, typ = (c, "coarse") if cf is None else (cf, "fine")
    if cf is not None:
        res["rgb_fine"], res["gate_loss_fine"] = cf["rgb"], cf["l_aux"]
    if get_depth:
        res[f"depth_{typ}"] = top["depth"]
    if get_depth_variance:
        res[f"depth_variance_{typ}"] = top["depth_variance"]
    if getattr(hparams, "moe_return_gates", False):
        res["moe_gates_coarse"] = c["idx"].long().view(N, S - 1, 1, 1)
        if cf is not None:
            res["moe_gates_fine"] = cf["idx"].long().view(N, F - 1, 1, 1)
    return res, False


def render_image_rays(nerf, bg_nerf, rays: torch.Tensor, image_index, hparams, sphere_center=None, sphere_radius=None,
                      radii: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
    """The pixel-batch loop of Runner.render_image / render_image_blocknerf (/root/reference/switch_nerf/runner.py:2835-2885,
    :2887-2950) over rays that are already built ([R, 8], R = H * W; ray construction from camera metadata is dataset code):
    batches of hparams.image_pixel_batch_size rays through render_rays (or render_rays_mip when radii are given) with
    get_depth=True, get_depth_variance=False, get_bg_fg_rgb=True, results concatenated on the host like the reference.
    The last batch of an image - and the last model chunk inside a batch - may be ragged."""
    R = rays.shape[0]
    rays = rays.reshape(-1, 8)
    idx = None
    if getattr(hparams, "appearance_dim", 1) > 0:
        idx = image_index if torch.is_tensor(image_index) and image_index.numel() == R else \
            torch.full((R,), int(image_index), dtype=torch.long, device=rays.device)
    results: Dict[str, list] = {}
    step = int(hparams.image_pixel_batch_size)
    for i in range(0, R, step):
        r = rays[i:i + step].contiguous()
        ii = None if idx is None else idx[i:i + step].contiguous()
        if radii is not None:
            batch, _ = render_rays_mip(nerf, r, radii.reshape(-1, 1)[i:i + step].contiguous(), ii, hparams, True, False)
        else:
            batch, _ = render_rays(nerf, bg_nerf, r, ii, hparams, sphere_center, sphere_radius, True, False, True)
        for k, v in batch.items():
            results.setdefault(k, []).append(v.cpu())
    return {k: torch.cat(v) for k, v in results.items()}
