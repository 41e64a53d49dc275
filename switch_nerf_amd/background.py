"""Foreground model + background model rendering: render_rays' bg_nerf branch
(/root/reference/switch_nerf/rendering.py:32-159; the default of the Mega-NeRF scenes, opts.py:89, with the ellipsoidal
bound the runner derives from the camera positions, runner.py:221-243).

  * every ray is clipped at the foreground bound (swn_fg_bounds); the foreground network (SwitchNeRF) samples
    [near, min(far, fg_far)] and its last sample's delta is the distance left to the bound (:42, :216-217);
  * rays with far > fg_far continue into the background model - the reference's dense NeRF over the NeRF++ inverted-sphere
    parametrisation (4-D points, DenseNeRF with xyz_dim 4) - on coarse_samples // 2 inverse-distance samples evaluated
    in descending order (swn_bg_sample_pe, flip);
  * rgb / depth of both are blended with the foreground's leftover transmittance bg_lambda (:104-131), whose gradient
    flows back into the foreground densities (swn_composite_bounded_bwd).

With fine_samples > 0 both models run the hierarchical pass (fine_samples // 2 for the background, :241).  One reference
quirk is mirrored deliberately: the background's importance sampling pairs the ASCENDING bin mid points (:238 uses
_get_results' own un-flipped z_vals) with the coarse weights in FLIPPED order (computed by _inference on its flipped
copy, :302-304), and its depth map reads the un-flipped depth_real (:483-484); tests/golden/bg_train_*.npz pin both.
"""
from __future__ import annotations

import torch

from . import ops


class BackgroundScene:
    """nerf: SwitchNeRF (foreground), bg_nerf: DenseNeRF with cfg xyz_dim = 4; sphere_center / sphere_radius: 3 floats each
    (tensors or sequences) or both None for the unit sphere."""

    def __init__(self, nerf, bg_nerf, sphere_center=None, sphere_radius=None):
        assert getattr(bg_nerf, "xyz_dim", 3) == 4, "the background model takes the 4-D inverted-sphere points (get_bg_nerf: xyz_dim 4)"
        self.nerf, self.bg = nerf, bg_nerf
        self.center, self.radius = sphere_center, sphere_radius
        bg_nerf._grow_bufs = True           # the number of background rays changes every batch
        self.dev = nerf.dev
        # ONE loss scaler over both models (the reference's single GradScaler over both optimizers, runner.py:483, 679-690): when either
        # model computes in fp16, both hold the SAME LossScaler object - one scale multiplies the loss gradient, both unscale by it, one
        # growth tracker is checkpointed whichever model's state_dict is asked (a 16-bit-float foreground beside an fp16 background included)
        # This MUTATES both models (documented contract of a scene): a model that had no scaler gains one, the background's own
        # scaler state is replaced by the foreground's.  Build the scene BEFORE capturing graphs on its models (a graph captured on a
        # scaler-less model does not read the scale); detach() hands the models back their own scalers.
        self._own_scalers = (nerf.loss_scaler, bg_nerf.loss_scaler)
        self.loss_scaler = nerf.loss_scaler if nerf.loss_scaler is not None else bg_nerf.loss_scaler
        if self.loss_scaler is not None:
            for m in (nerf, bg_nerf):
                if m.loss_scaler is not self.loss_scaler and getattr(m, "_train_graphs", None):
                    raise RuntimeError("BackgroundScene: a model whose training graphs were captured before the scene was built would keep "
                                       "running without the shared loss scale; build the scene first")
            nerf.loss_scaler = bg_nerf.loss_scaler = self.loss_scaler

    def detach(self):
        """Undo the scaler sharing: each model gets back the LossScaler (or None) it had before the scene was built; a model that owned
        one continues from the shared scale."""
        for m, own in zip((self.nerf, self.bg), self._own_scalers):
            if own is not None and self.loss_scaler is not None and own is not self.loss_scaler:
                own.load_state_dict(self.loss_scaler.state_dict())
            m.loss_scaler = own
            if own is None:
                m._ls_dev = None

    # ------------------------------------------------------------------------------------------ forward
    def forward(self, rays, image_indices, n_samples, seg_tokens, perturb=0.0, perturb_rand=None, perturb_rand_bg=None,
                sigma_noise=None, sigma_noise_bg=None, fine_samples=0, fine_u=None, fine_u_bg=None, sigma_noise_fine=None,
                sigma_noise_bg_fine=None, no_batch=False, noise_std=0.0, training=True):
        """sigma_noise_bg / sigma_noise_bg_fine: [Nb * samples] tensors, or the string "randn" to draw noise_std * N(0,1)
        here (the number of background rays is only known inside)."""
        o, nerf, bg = ops, self.nerf, self.bg
        N, S, Fn = rays.shape[0], n_samples, int(fine_samples)
        rays_fg, fg_far, last0, has_bg, n_out = o.fg_bounds(rays, self.center, self.radius)
        idx_bg = has_bg.nonzero().view(-1)                      # rays_with_bg, rendering.py:36 (host sync, like the reference)
        if int(n_out.item()) > 0:
            raise Exception("Not all your cameras are bounded by the unit sphere; please make sure the cameras are normalized properly!")
        Nb = idx_bg.numel()
        ctx = dict(N=N, S=S, F=Fn, idx_bg=idx_bg, Nb=Nb, fg_far=fg_far, has_bg=has_bg)
        det_u = lambda n, k: torch.linspace(0, 1, k).expand(n, k).contiguous().to(self.dev) if perturb == 0 else torch.rand(n, k, device=self.dev)
        # ---- background first (the reference's order, and so the order of its random draws)
        bg._saving = bool(training)          # (DenseNeRF._net_forward is called directly below; reset after the block)
        if Nb > 0:
            Sb = S // 2
            rays_b = rays.index_select(0, idx_bg).contiguous()
            img_b = image_indices.index_select(0, idx_bg).contiguous()
            if perturb > 0 and perturb_rand_bg is None:
                perturb_rand_bg = torch.rand(Nb, Sb, device=self.dev)
            z_b, dreal_b, pe_b = o.bg_sample_pe(rays_b, self.center, self.radius, Sb, bg.cfg["pos_xyz_dim"], bg.dtype, bg.KP,
                                                perturb_rand_bg, float(perturb), pe_out=bg._buf("c:pe", (Nb * Sb, bg.KP), bg.dtype))
            pe_dir_b = bg._dir_pe(rays_b)
            if isinstance(sigma_noise_bg, str):
                sigma_noise_bg = torch.randn(Nb * Sb, device=self.dev) * noise_std
            cb = bg._net_forward(pe_b, pe_dir_b, img_b, Nb, Sb, Nb * Sb, sigma_noise_bg, None, False, "c")
            cb["z"], cb["depth_real"] = z_b, dreal_b
            rgb_b, depth_b, _, w_b, _ = o.composite_bounded_fwd(cb["raw"], z_b, None, True, dreal_b, want_weights=Fn > 0)
            b = dict(c=cb, raw=cb["raw"], z=z_b)
            if Fn > 0:
                Fb = Fn // 2
                if fine_u_bg is None:
                    fine_u_bg = det_u(Nb, Fb)
                if isinstance(sigma_noise_bg_fine, str):
                    sigma_noise_bg_fine = torch.randn(Nb * Fb, device=self.dev) * noise_std
                z_f = o.sample_pdf(z_b.flip(-1).contiguous(), w_b, fine_u_bg, Fb)        # ascending bins x flipped weights (see module doc)
                _, dreal_f, pe_f = o.bg_sample_pe(rays_b, self.center, self.radius, Fb, bg.cfg["pos_xyz_dim"], bg.dtype, bg.KP,
                                                  z_in=z_f, pe_out=bg._buf("f:pe", (Nb * Fb, bg.KP), bg.dtype))
                cfb = bg._net_forward(pe_f, pe_dir_b, img_b, Nb, Fb, Nb * Fb, sigma_noise_bg_fine, None, False, "f")
                # descending sort of the union (:421 descending=flip) = ascending sort of the negated depths
                zneg, order, raw_m = o.merge_samples((-z_f).contiguous(), (-z_b).contiguous(), cfb["raw"], cb["raw"])
                z_m = -zneg
                dreal_m = torch.gather(torch.cat([dreal_f, dreal_b], 1), 1, order.long())    # :432-433
                rgb_b, depth_b, _, _, _ = o.composite_bounded_fwd(raw_m, z_m, None, True, dreal_m)
                b.update(cf=cfb, raw=raw_m, z=z_m, order=order, z_fine=z_f)
            b.update(rgb=rgb_b, depth=depth_b)
            ctx["bg"] = b
        # ---- foreground on the clipped rays
        bg._saving = True
        c = nerf.forward_rays(rays_fg, image_indices, S, seg_tokens, perturb, perturb_rand, sigma_noise, training, None,
                              no_batch=no_batch, want_weights=Fn > 0, composite=Fn > 0)
        ctx["c"] = c
        if Fn == 0:
            raw, z, z_last = c["raw"], c["z"], c["z"][:, -1]              # stratified depths ascend: the last one is the maximum (:217)
        else:
            if fine_u is None:
                fine_u = det_u(N, Fn)
            z_fine = o.sample_pdf(c["z"], c["weights"], fine_u, Fn)
            cf = nerf.forward_rays(rays_fg, image_indices, Fn, min(seg_tokens, N * Fn), 0.0, None, sigma_noise_fine, training, None,
                                   no_batch=no_batch, z_in=z_fine, pe_dir=c["pe_dir"], tag="f", composite=False)
            z, order, raw = o.merge_samples(z_fine, c["z"], cf["raw"], c["raw"])
            z_last = z_fine.max(dim=-1)[0]                               # the FINE depths' maximum (:249-250)
            ctx.update(cf=cf, order=order, z_fine=z_fine)
        last_delta = torch.where(has_bg > 0, fg_far - z_last, last0).contiguous()          # :216-217 / :249-250
        rgb, depth, dvar, _, lam = o.composite_bounded_fwd(raw, z, last_delta, False, None, want_bg_lambda=True)
        ctx.update(raw=raw, z=z, last_delta=last_delta, bg_lambda=lam, fg_rgb=rgb, fg_depth=depth, depth_variance=dvar)
        if Nb > 0:                                                        # :104-131
            lam_b = lam.index_select(0, idx_bg)
            rgb = rgb.index_add(0, idx_bg, ctx["bg"]["rgb"] * lam_b[:, None])
            depth = depth.index_add(0, idx_bg, ctx["bg"]["depth"] * lam_b)
        ctx.update(rgb=rgb, depth=depth)
        return ctx

    # ------------------------------------------------------------------------------------------ backward
    def backward(self, ctx, d_rgb, wt_coarse, wt_fine=0.0):
        """Accumulates both models' parameter gradients given dL/d rgb [N,3]; wt_* = the weight of each level's mean gate loss."""
        o, nerf, bg = ops, self.nerf, self.bg
        S, Fn, Nb, idx_bg = ctx["S"], ctx["F"], ctx["Nb"], ctx["idx_bg"]
        d_lam = None
        if Nb > 0:
            b = ctx["bg"]
            d_rgb_sel = d_rgb.index_select(0, idx_bg)
            d_lam = torch.zeros(ctx["N"], dtype=torch.float32, device=self.dev)
            d_lam.index_copy_(0, idx_bg, (d_rgb_sel * b["rgb"]).sum(-1))
            d_rgb_b = (d_rgb_sel * ctx["bg_lambda"].index_select(0, idx_bg)[:, None]).contiguous()
            d_raw_b = o.composite_bounded_bwd(b["raw"], b["z"], d_rgb_b, None, True, None)
            if Fn > 0:
                d_f, d_c = o.unmerge_grad(d_raw_b, b["order"], Fn // 2, S // 2)
                bg.backward_net(b["cf"], d_f)
                bg.backward_net(b["c"], d_c)
            else:
                bg.backward_net(b["c"], d_raw_b)
        d_raw = o.composite_bounded_bwd(ctx["raw"], ctx["z"], d_rgb, ctx["last_delta"], False, d_lam)
        c = ctx["c"]
        full = lambda cc, w: torch.full((cc["n_seg"],), w / cc["n_seg"], dtype=torch.float32, device=self.dev)
        if Fn > 0:
            d_f, d_c = o.unmerge_grad(d_raw, ctx["order"], Fn, S)
            nerf.backward_net(ctx["cf"], d_f, full(ctx["cf"], wt_fine))
            nerf.backward_net(c, d_c, full(c, wt_coarse))
        else:
            nerf.backward_net(c, d_raw, full(c, wt_coarse))

    # ------------------------------------------------------------------------------------------ training step
    def train_step(self, rgbs, rays, image_indices, n_samples, seg_tokens, perturb=1.0, optimizer_step=True, grad_allreduce=None,
                   fine_samples=0, **kw):
        """Runner._training_step with a background model (runner.py:1077-1123): loss = mse(rgb) + wt * gate loss of the
        foreground MoE (the dense background model has none); both models take an Adam step (runner.py:305-318 builds one
        optimizer over the parameters of both)."""
        nerf, bg = self.nerf, self.bg
        nerf.grad.zero_()
        bg.grad.zero_()
        ctx = self.forward(rays, image_indices, n_samples, seg_tokens, perturb, fine_samples=fine_samples, **kw)
        fine = fine_samples > 0
        gate_loss = ctx["c"]["l_aux"].mean()
        if fine:
            gate_loss = (ctx["cf"]["l_aux"].mean() + gate_loss) / 2.0
        diff = ctx["rgb"] - rgbs
        photo = (diff * diff).mean()
        loss = photo + nerf.wt * gate_loss
        # fp16 (the reference's single GradScaler over both optimizers, runner.py:483, 679-690): one loss scale - the foreground model's
        # scaler - multiplies the loss gradient; both models unscale by it, and a non-finite gradient in either skips that model's step
        ls = 1.0
        if self.loss_scaler is not None:
            ls = float(self.loss_scaler.scale)
            for m in (nerf, bg):
                m._loss_scale_tensor()                    # (records the scale this backward uses: _found_inf divides by it)
        d_rgb = (diff * (2.0 * ls / diff.numel())).contiguous()
        self.backward(ctx, d_rgb, ls * nerf.wt * (0.5 if fine else 1.0), ls * nerf.wt * 0.5)
        # the reference skips the background optimizer on batches without background rays (runner.py:683: `if key == 'bg_nerf'
        # and not bg_nerf_rays_present: continue`): no Adam step, no moment decay, no step-counter increment - in a SINGLE-process
        # run.  In a distributed run ('RANK' in os.environ) render_rays makes a dummy background forward on a rank without background
        # rays and reports bg_nerf_rays_present = True (rendering.py:162-194), so there the background Adam steps on EVERY iteration
        # (with a zero gradient when no rank had background rays: moments decay, step count grows).  Same rule here.
        import torch.distributed as dist
        distributed = grad_allreduce is not None or (dist.is_available() and dist.is_initialized())
        bg_present = True if distributed else ctx["Nb"] > 0
        models = [nerf] + ([bg] if bg_present else [])
        scale = {id(m): (grad_allreduce(m._allreduce_view()) if grad_allreduce is not None else 1.0) for m in (nerf, bg)}
        # fp16: the reference holds ONE GradScaler over both optimizers (runner.py:483, 686-690).  GradScaler.step skips an optimizer on
        # ITS OWN non-finite gradient; GradScaler.update backs the shared scale off when EITHER optimizer found one (and only then
        # resets the growth tracker).  The foreground model's LossScaler is that shared scaler; the background's mirrors its scale.
        found = {id(m): False for m in (nerf, bg)}
        if optimizer_step and self.loss_scaler is not None:
            for m in models:
                found[id(m)] = m._found_inf()
            self.loss_scaler.update(any(found.values()))
            for m in (nerf, bg):
                m._loss_scale_tensor()
        for m in models:
            if optimizer_step and not found[id(m)]:
                sc = scale[id(m)]
                if self.loss_scaler is not None:         # fp16: the gradient carries the loss scale (backward above)
                    sc /= m._applied_loss_scale
                m.step_count += 1
                ops.adam_step(m.flat, m.grad, m.m, m.v, None, m.step_count, m.lr, grad_scale=sc)
                m.refresh_compute_copies()
        return dict(loss=loss, photo_loss=photo, gate_loss=gate_loss, psnr=-10.0 * torch.log10(photo), rgb=ctx["rgb"],
                    depth=ctx["depth"], depth_variance=ctx["depth_variance"].mean(), ctx=ctx, bg_nerf_rays_present=bg_present)
