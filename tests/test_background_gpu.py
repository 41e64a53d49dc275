"""Background model + foreground bound on the HIP path (render_rays' bg_nerf branch, rendering.py:32-159): kernels against
the golden vectors of the reference's _intersect_sphere / _depth2pts_outside and against the CPU oracle, the whole
training step (both models' gradients) against the reference's own run (oracle/gen_golden.py gen_bg)."""
import os
from argparse import Namespace

import numpy as np
import pytest
import torch

import synth
from oracle import switchnerf_oracle as O

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
CENTER, RADIUS = synth.SPHERE_CENTER, synth.SPHERE_RADIUS


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _models(dtype, seed, seed_bg, gate_scale=0.02):
    from switch_nerf_amd.model import SwitchNeRF
    from switch_nerf_amd.dense import DenseNeRF
    m = SwitchNeRF(synth.BUILDING, dtype=dtype)
    m.load_state_dict(synth.make_weights(seed, synth.BUILDING, gate_scale=gate_scale))
    b = DenseNeRF(synth.DENSE_BG, dtype=dtype)
    b.load_state_dict(synth.make_dense_weights(seed_bg, synth.DENSE_BG))
    return m, b


def test_fg_bounds_and_background_points_vs_reference_golden():
    from switch_nerf_amd import ops
    g = np.load(os.path.join(G, "bg_points.npz"))
    rays, _, _ = synth.make_bg_rays(84, 40)
    r = _dev(rays)
    rays_fg, fg_far, last_delta, has_bg, n_out = ops.fg_bounds(r, CENTER, RADIUS)
    near, far = rays[:, 6], rays[:, 7]
    ref_far = np.maximum(g["fg_far"], near)
    np.testing.assert_allclose(fg_far.cpu().numpy(), ref_far, rtol=1e-6)
    hb = far > ref_far
    np.testing.assert_array_equal(has_bg.cpu().numpy() != 0, hb)
    np.testing.assert_allclose(last_delta.cpu().numpy(), np.where(hb, ref_far, 1e10).astype(np.float32), rtol=1e-6)
    np.testing.assert_array_equal(rays_fg[:, :7].cpu().numpy(), rays[:, :7])
    np.testing.assert_allclose(rays_fg[:, 7].cpu().numpy(), np.minimum(far, ref_far), rtol=1e-6)
    assert int(n_out.item()) == 0
    # cameras outside the bound: counted (the host mirror raises like the reference)
    far_rays = rays.copy()
    far_rays[:5, :3] += 5.0
    assert int(ops.fg_bounds(_dev(far_rays), CENTER, RADIUS)[4].item()) == 5
    # points / metric depths for supplied inverse distances; the first 4 PE columns are the point itself (fp32)
    depth = g["depth"]
    _, dreal, pe = ops.bg_sample_pe(r, CENTER, RADIUS, depth.shape[1], 12, torch.float32, 128, z_in=_dev(depth))
    pe = pe.cpu().numpy().reshape(40, depth.shape[1], 128)
    np.testing.assert_allclose(pe[..., :4], g["pts"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(dreal.cpu().numpy(), g["depth_real"], rtol=2e-5, atol=1e-5)
    enc = O.positional_encoding(torch.from_numpy(g["pts"]).reshape(-1, 4), 12).numpy().reshape(40, -1, 100)
    # sin/cos(2^k x) amplify the 1-ulp differences of x by 2^k: tolerance 2^k * 2.4e-7 * |x| + 1e-6
    tol = np.concatenate([np.full(4, 2e-6)] + [np.full(8, 2.0 ** k * 5e-7 + 2e-6) for k in range(12)])
    assert (np.abs(pe[..., :100] - enc) <= tol).all(), np.abs(pe[..., :100] - enc).max(axis=(0, 1))
    assert (pe[..., 100:] == 0).all()


@pytest.mark.parametrize("perturb", [0.0, 1.0])
def test_background_sampling_flipped_order(perturb):
    """Coarse background samples: z descending (flip), depth_real ascending (the reference never flips it), bf16 rows."""
    from switch_nerf_amd import ops
    N, S = 77, 32
    rays, _, _ = synth.make_bg_rays(91, N)
    pr = np.random.default_rng(92).uniform(0, 1, (N, S)).astype(np.float32)
    z, dreal, pe = ops.bg_sample_pe(_dev(rays), CENTER, RADIUS, S, 12, torch.float32, 128, _dev(pr) if perturb else None, perturb)
    z_ref = O.sample_z(torch.zeros(N, 1), torch.ones(N, 1), S, perturb, torch.from_numpy(pr) if perturb else None)
    np.testing.assert_array_equal(z.cpu().numpy(), torch.flip(z_ref, dims=[-1]).numpy())
    r = torch.from_numpy(rays)
    pts, dr = O.depth2pts_outside(r[:, None, :3], r[:, None, 3:6], z_ref, torch.from_numpy(CENTER), torch.from_numpy(RADIUS))
    np.testing.assert_allclose(dreal.cpu().numpy(), dr.numpy(), rtol=3e-5, atol=1e-5)
    got = pe.cpu().numpy().reshape(N, S, 128)[..., :4]
    np.testing.assert_allclose(got, torch.flip(pts, dims=[-2]).numpy(), rtol=0, atol=2e-6)
    _, _, pe16 = ops.bg_sample_pe(_dev(rays), CENTER, RADIUS, S, 12, torch.bfloat16, 128, _dev(pr) if perturb else None, perturb)
    ref = pe.reshape(-1, 128)
    assert (pe16.float() - ref).abs().max().item() < 1.2e-2          # bf16 rounding + angle doubling


@pytest.mark.parametrize("flip", [False, True])
def test_composite_bounded_vs_oracle(flip):
    from switch_nerf_amd import ops
    rng = np.random.default_rng(95)
    N, S = 130, 96
    z = np.sort(rng.uniform(0.05, 1.0, (N, S)).astype(np.float32), 1)
    if flip:
        z = z[:, ::-1].copy()
    raw = np.concatenate([rng.uniform(0, 1, (N, S, 3)), np.abs(rng.standard_normal((N, S, 1))) * 8], -1).astype(np.float32)
    ld = np.where(rng.uniform(size=N) < 0.5, rng.uniform(0.01, 0.3, N), 1e10).astype(np.float32)
    dreal = rng.uniform(1, 50, (N, S)).astype(np.float32)
    d_rgb = rng.standard_normal((N, 3)).astype(np.float32)
    d_lam = rng.standard_normal(N).astype(np.float32)
    rt = torch.from_numpy(raw).requires_grad_(True)
    c = O.composite(rt[..., :3], rt[..., 3], torch.from_numpy(z), torch.from_numpy(ld)[:, None], flip=flip, depth_real=torch.from_numpy(dreal))
    ((c["rgb"] * torch.from_numpy(d_rgb)).sum() + (c["bg_lambda"] * torch.from_numpy(d_lam)).sum()).backward()
    rgb, depth, dvar, w, lam = ops.composite_bounded_fwd(_dev(raw.reshape(-1, 4)), _dev(z), _dev(ld), flip, _dev(dreal), True, True)
    np.testing.assert_allclose(rgb.cpu().numpy(), c["rgb"].detach().numpy(), rtol=0, atol=2e-6)
    np.testing.assert_allclose(lam.cpu().numpy(), c["bg_lambda"].detach().numpy(), rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(w.cpu().numpy(), c["weights"].detach().numpy(), rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(depth.cpu().numpy(), c["depth"].numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(dvar.cpu().numpy(), c["depth_variance"].numpy(), rtol=1e-4, atol=1e-5)
    d_raw = ops.composite_bounded_bwd(_dev(raw.reshape(-1, 4)), _dev(z), _dev(d_rgb), _dev(ld), flip, _dev(d_lam))
    ref = rt.grad.numpy().reshape(-1, 4)
    np.testing.assert_allclose(d_raw.cpu().numpy(), ref, rtol=2e-4, atol=1e-6 + 1e-5 * np.abs(ref).max())


def _check_grads(g, pre, model, tol_rel=2e-4):
    worst = 0.0
    for k, t in model.grad_dict().items():
        got = t.cpu().numpy()
        ref_sum = g["gsum__" + pre + k]
        scale = max(1e-12, float(ref_sum[1]))
        # the density head's gradient is a sum of per-point terms of both signs that nearly cancel (|sum| ~ 1e-4 of the
        # terms' magnitude): its fp32 summation noise is larger relative to the result than for the other tensors
        rel = 5e-3 if "sigma" in k else 1e-3
        assert abs(synth.checksum(got)[0] - ref_sum[0]) <= rel * scale + 1e-9, pre + k
        assert abs(synth.checksum(got)[1] - ref_sum[1]) <= rel * scale + 1e-9, pre + k
        sl = got.reshape(-1)[:: max(1, got.size // 499)][:499]
        ref = g["gslice__" + pre + k]
        worst = max(worst, float(np.abs(sl - ref).max() / (np.abs(ref).max() + 1e-12)))
        np.testing.assert_allclose(sl, ref, rtol=2e-3, atol=1e-7 + (25 if "sigma" in k else 1) * tol_rel * np.abs(ref).max(), err_msg=pre + k)
    return worst


@pytest.mark.parametrize("tag", ["coarse_det", "coarse", "fine"])
def test_background_train_step_vs_reference_golden_fp32(tag):
    from switch_nerf_amd.background import BackgroundScene
    g = np.load(os.path.join(G, f"bg_train_{tag}.npz"))
    N, S, Fn, chunk = int(g["N"]), int(g["S"]), int(g["F"]), int(g["chunk"])
    m, b = _models(torch.float32, int(g["seed"]), int(g["seed_bg"]), float(g["gate_scale"]))
    scene = BackgroundScene(m, b, CENTER, RADIUS)
    rays, img, rgbs = synth.make_bg_rays(83, N)
    kw = {}
    perturb = float(g["perturb"])
    if perturb > 0:
        kw = dict(perturb_rand=_dev(g["perturb_rand"]), perturb_rand_bg=_dev(g["perturb_rand_bg"]))
        if Fn:
            kw.update(fine_u=_dev(g["fine_u"]), fine_u_bg=_dev(g["fine_u_bg"]))
    st = scene.train_step(_dev(rgbs), _dev(rays), _dev(img), S, chunk, perturb=perturb, optimizer_step=False, fine_samples=Fn, **kw)
    ctx = st["ctx"]
    assert ctx["Nb"] == int(g["n_bg"])
    np.testing.assert_allclose(ctx["fg_far"].cpu().numpy(), g["fg_far"], rtol=1e-6)
    np.testing.assert_allclose(ctx["fg_rgb"].cpu().numpy(), g["fg_rgb"], rtol=0, atol=1e-4)
    np.testing.assert_allclose((ctx["rgb"] - ctx["fg_rgb"]).cpu().numpy(), g["bg_rgb"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(ctx["rgb"].cpu().numpy(), g["rgb"], rtol=0, atol=1e-4)               # north-star tolerance
    np.testing.assert_allclose(ctx["depth"].cpu().numpy(), g["depth"], rtol=2e-3, atol=1e-4)
    np.testing.assert_allclose(ctx["depth_variance"].cpu().numpy(), g["depth_variance"], rtol=2e-3, atol=1e-6)
    np.testing.assert_allclose(st["loss"].item(), float(g["loss"]), rtol=1e-5)
    w1 = _check_grads(g, "", m)
    w2 = _check_grads(g, "bg__", b)
    print(f"{tag}: worst relative gradient-slice error fg {w1:.2e} bg {w2:.2e}")


def test_no_background_rays_equals_plain_path_and_render_rays_mirror():
    """Rays that all end inside the bound: no background evaluation, the result is the plain foreground rendering; then
    the render_rays mirror with a background model (result keys of rendering.py:104-131)."""
    from switch_nerf_amd.background import BackgroundScene
    from switch_nerf_amd import rendering
    m, b = _models(torch.float32, 101, 102)
    N, S = 64, 64
    rays, img, rgbs = synth.make_rays(103, N, far=0.3)            # fg_far >= ~0.45 for these origins
    scene = BackgroundScene(m, b, CENTER, RADIUS)
    ctx = scene.forward(_dev(rays), _dev(img), S, 1024)
    assert ctx["Nb"] == 0
    c = m.forward_rays(_dev(rays), _dev(img), S, 1024)
    np.testing.assert_array_equal(ctx["rgb"].cpu().numpy(), c["rgb"].cpu().numpy())
    np.testing.assert_array_equal(ctx["depth"].cpu().numpy(), c["depth"].cpu().numpy())
    assert float(b.grad.abs().max()) == 0.0
    h = Namespace(coarse_samples=S, fine_samples=32, model_chunk_size=1024, perturb=1.0, use_sigma_noise=True, sigma_noise_std=1.0,
                  use_cascade=False, moe_return_gates=True)
    rays2, img2, _ = synth.make_bg_rays(104, N)
    m.train()
    torch.manual_seed(0)
    res, present = rendering.render_rays(m, b, _dev(rays2), _dev(img2), h, CENTER, RADIUS, True, True, True)
    assert present
    for k in ("rgb_fine", "depth_fine", "depth_variance_fine", "fg_rgb_fine", "bg_rgb_fine", "gate_loss_coarse", "gate_loss_fine",
              "moe_gates_coarse", "moe_gates_fine"):
        assert k in res, k
    assert res["rgb_fine"].shape == (N, 3) and torch.isfinite(res["rgb_fine"]).all()
    assert (res["bg_rgb_fine"].abs().sum(-1) > 0).sum().item() > 0
    m.eval()
    res2, _ = rendering.render_rays(m, b, _dev(rays2), _dev(img2), h, CENTER, RADIUS, True, False, False)
    res3, _ = rendering.render_rays(m, b, _dev(rays2), _dev(img2), h, CENTER, RADIUS, True, False, False)
    np.testing.assert_array_equal(res2["rgb_fine"].cpu().numpy(), res3["rgb_fine"].cpu().numpy())      # eval: deterministic
    with pytest.raises(Exception, match="bounded by the unit sphere"):
        bad = rays2.copy()
        bad[:, :3] += 3.0
        rendering.render_rays(m, b, _dev(bad), _dev(img2), h, CENTER, RADIUS, True, False, False)


def test_background_bf16_step_close_to_fp32_and_adam_updates_both():
    from switch_nerf_amd.background import BackgroundScene
    N, S = 256, 64
    rays, img, rgbs = synth.make_bg_rays(111, N)
    pr = np.random.default_rng(112).uniform(0, 1, (N, S)).astype(np.float32)
    out = {}
    for dt in (torch.float32, torch.bfloat16):
        m, b = _models(dt, 113, 114)
        scene = BackgroundScene(m, b, CENTER, RADIUS)
        Nb = int((torch.from_numpy(rays[:, 7]) > torch.maximum(O.intersect_sphere(torch.from_numpy(rays[:, :3]), torch.from_numpy(rays[:, 3:6]),
                  torch.from_numpy(CENTER), torch.from_numpy(RADIUS)), torch.from_numpy(rays[:, 6]))).sum())
        prb = np.random.default_rng(115).uniform(0, 1, (Nb, S // 2)).astype(np.float32)
        before = (m.flat.clone(), b.flat.clone())
        st = scene.train_step(_dev(rgbs), _dev(rays), _dev(img), S, 4096, perturb=1.0, perturb_rand=_dev(pr), perturb_rand_bg=_dev(prb))
        out[dt] = (st["rgb"].float().cpu().numpy(), st["loss"].item())
        assert (m.flat != before[0]).any() and (b.flat != before[1]).any()
        assert m.step_count == 1 and b.step_count == 1
    assert np.abs(out[torch.float32][0] - out[torch.bfloat16][0]).max() < 3e-2
    assert abs(out[torch.float32][1] - out[torch.bfloat16][1]) < 2e-2 * out[torch.float32][1]


def test_background_fp16_one_shared_grad_scaler():
    """The reference holds ONE GradScaler over both optimizers (runner.py:483, 686-690): GradScaler.step skips an optimizer on ITS OWN
    non-finite gradient, GradScaler.update backs the shared scale off when EITHER found one.  An overflow in the background gradient
    alone must therefore halve the shared scale, skip the background step only, and reset the growth tracker."""
    from switch_nerf_amd.background import BackgroundScene
    N, S = 256, 64
    rays, img, rgbs = synth.make_bg_rays(131, N)
    m, b = _models(torch.float16, 133, 134)
    scene = BackgroundScene(m, b, CENTER, RADIUS)
    assert m.loss_scaler is b.loss_scaler is scene.loss_scaler          # ONE object: one scale, one growth tracker to checkpoint
    m.loss_scaler.scale = 1024.0
    m.loss_scaler._good = 7
    st = scene.train_step(_dev(rgbs), _dev(rays), _dev(img), S, 4096, perturb=0.0)
    assert st["ctx"]["Nb"] > 0 and m.step_count == 1 and b.step_count == 1
    assert m.loss_scaler.scale == 1024.0 and b.loss_scaler.scale == 1024.0 and m.loss_scaler._good == 8
    # poison the BACKGROUND gradient only: its backward accumulates into bg.grad, so a non-finite value planted in a parameter the
    # foreground never reads (the background's own first-layer bias) makes bg.grad non-finite and leaves nerf.grad finite
    fg_before, bg_before = m.flat.clone(), b.flat.clone()
    orig_backward = scene.backward

    def poisoned(*a, **k):
        orig_backward(*a, **k)
        b.grad[0] = float("inf")
    scene.backward = poisoned
    scene.train_step(_dev(rgbs), _dev(rays), _dev(img), S, 4096, perturb=0.0)
    scene.backward = orig_backward
    assert m.step_count == 2 and b.step_count == 1                      # each optimizer skips on ITS OWN found_inf
    assert (m.flat != fg_before).any() and torch.equal(b.flat, bg_before)
    assert m.loss_scaler.scale == 512.0 and b.loss_scaler.scale == 512.0 and m.loss_scaler._good == 0      # ONE scaler: backed off for both
    assert b.loss_scaler.state_dict()["_growth_tracker"] == 0 and b.loss_scaler.skipped == 1               # ... whichever model is asked


def test_background_model_call_mirror_on_explicit_points():
    """NeRF.forward of the xyz_dim = 4 background model on explicit inverted-sphere points (nerf.py:143-190)."""
    from switch_nerf_amd.dense import DenseNeRF
    b = DenseNeRF(synth.DENSE_BG, dtype=torch.float32)
    sd = synth.make_dense_weights(121, synth.DENSE_BG)
    b.load_state_dict(sd)
    b.eval()
    rng = np.random.default_rng(122)
    P = 555
    p3 = rng.standard_normal((P, 3))
    p3 /= np.linalg.norm(p3, axis=1, keepdims=True)
    d = rng.standard_normal((P, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    x = np.concatenate([p3, rng.uniform(0.01, 1, (P, 1)), d, rng.integers(0, 10, (P, 1))], 1).astype(np.float32)
    got = b(_dev(x)).cpu().numpy()
    ref = O.nerf_dense_forward(O.params_from_numpy(sd), torch.from_numpy(x), synth.DENSE_BG).detach().numpy()
    np.testing.assert_allclose(got[:, :3], ref[:, :3], rtol=0, atol=1e-4)
    np.testing.assert_allclose(got[:, 3], ref[:, 3], rtol=1e-4, atol=1e-4)
    sig = b(_dev(x[:, :4]), sigma_only=True).cpu().numpy()
    np.testing.assert_allclose(sig[:, 0], ref[:, 3], rtol=1e-4, atol=1e-4)
    with pytest.raises(Exception, match="Unexpected input shape"):
        b(_dev(x[:, :7]))
