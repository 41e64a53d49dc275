"""The torch.autograd bridge (switch_nerf_amd/autograd.py): the reference's Runner loop - render_rays -> loss -> scaler.scale(loss)
.backward() -> torch.optim.Adam + ExponentialLR (runner.py:486-512, 604-693) - must drive the HIP path unchanged."""
from argparse import Namespace

import numpy as np
import pytest
import torch

import synth

pytestmark = pytest.mark.gpu


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _model(dtype, seed, gate_scale=0.05):
    from switch_nerf_amd.model import SwitchNeRF
    m = SwitchNeRF(synth.BUILDING, dtype=dtype)
    m.load_state_dict(synth.make_weights(seed, synth.BUILDING, gate_scale=gate_scale))
    return m


def _hp(S, F, chunk, perturb=0.0):
    return Namespace(coarse_samples=S, fine_samples=F, model_chunk_size=chunk, perturb=perturb, use_sigma_noise=False, sigma_noise_std=0.0,
                     use_cascade=False)


def _runner_loss(res, rgbs, wt):
    """Runner._training_step + the loss assembly of Runner.train (runner.py:1094-1111, 646-651)."""
    typ = "fine" if "rgb_fine" in res else "coarse"
    photo = torch.nn.functional.mse_loss(res[f"rgb_{typ}"], rgbs, reduction="mean")
    gate_loss = res["gate_loss_coarse"].mean()
    if typ == "fine":
        gate_loss = (res["gate_loss_fine"].mean() + gate_loss) / 2
    return photo + wt * gate_loss


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("fine", [0, 64])
def test_backward_through_render_rays_fills_the_same_gradients_as_train_step(dtype, fine):
    from switch_nerf_amd.rendering import render_rays
    N, S, chunk = 64, 64, 1024
    rays, img, rgbs = synth.make_rays(402, N)
    a, b = _model(dtype, 401), _model(dtype, 401)
    st = a.grad_step(_dev(rgbs), _dev(rays), _dev(img), S, chunk, perturb=0.0, fine_samples=fine)
    res, _ = render_rays(b, None, _dev(rays), _dev(img), _hp(S, fine, chunk), None, None, True, True, False)
    key = "rgb_fine" if fine else "rgb_coarse"
    assert res[key].requires_grad and res["gate_loss_coarse"].requires_grad and not res["depth_" + key[4:]].requires_grad
    loss = _runner_loss(res, _dev(rgbs), b.wt)
    assert abs(loss.item() - st["loss"].item()) <= 1e-6 * abs(st["loss"].item()) + 1e-9
    loss.backward()
    g_ref, g = a.grad, b.flat_param.grad
    assert g is not None and g.shape == g_ref.shape
    err = (g - g_ref).abs().max().item() / g_ref.abs().max().item()
    assert err <= (1e-5 if dtype == torch.float32 else 2e-3), err         # (bf16: atomically accumulated weight gradients)
    # a second micro-batch accumulates into .grad like any torch parameter
    res2, _ = render_rays(b, None, _dev(rays), _dev(img), _hp(S, fine, chunk), None, None, True, True, False)
    _runner_loss(res2, _dev(rgbs), b.wt).backward()
    err2 = (b.flat_param.grad - 2 * g_ref).abs().max().item() / g_ref.abs().max().item()
    assert err2 <= (2e-5 if dtype == torch.float32 else 4e-3), err2
    # evaluation / no_grad: plain tensors, nothing recorded
    with torch.no_grad():
        res3, _ = render_rays(b, None, _dev(rays), _dev(img), _hp(S, fine, chunk), None, None, True, True, False)
    assert not res3[key].requires_grad


def test_runner_style_loop_with_torch_adam_scheduler_and_gradscaler():
    """Three iterations of the reference's loop (GradScaler -> backward -> scaler.step(Adam) -> ExponentialLR.step) on the autograd
    bridge against SwitchNeRF.train_step with the same learning-rate schedule (set_iteration): same losses, same parameters."""
    from torch.optim.lr_scheduler import ExponentialLR
    from switch_nerf_amd.rendering import render_rays
    N, S, chunk = 64, 64, 1024
    rays, img, rgbs = synth.make_rays(412, N)
    a, b = _model(torch.float32, 411), _model(torch.float32, 411)
    lr, decay, total = 5e-4, 0.1, 50
    opt = torch.optim.Adam(b.trainable_parameters(), lr=lr)
    sch = ExponentialLR(opt, gamma=decay ** (1 / total), last_epoch=-1)
    scaler = torch.amp.GradScaler("cuda", enabled=True, init_scale=1024.0)
    b.train()
    for it in range(3):
        a.set_iteration(it, decay, total)
        sa = a.train_step(_dev(rgbs), _dev(rays), _dev(img), S, chunk, perturb=0.0)
        res, _ = render_rays(b, None, _dev(rays), _dev(img), _hp(S, 0, chunk), None, None, False, True, False)
        loss = _runner_loss(res, _dev(rgbs), b.wt)
        opt.zero_grad(set_to_none=True)
        scaler.scale(loss).backward()
        scaler.step(opt)
        scaler.update()
        sch.step()
        assert abs(loss.item() - sa["loss"].item()) <= 2e-5 * abs(sa["loss"].item()), (it, loss.item(), sa["loss"].item())
    d = (a.flat - b.flat).abs().max().item()
    print(f"runner-style loop: max parameter difference after 3 steps {d:.2e}")
    assert d < 2e-5


def test_eval_after_external_optimizer_step_uses_fresh_weights():
    """after torch.optim.Adam moved flat_param, an evaluation / no_grad render must read the NEW weights in every
    kernel (the packed compute copies are refreshed at the top of each forward, not only inside the autograd Function)."""
    from switch_nerf_amd.rendering import render_rays
    N, S, chunk = 64, 64, 1024
    rays, img, rgbs = synth.make_rays(422, N)
    m = _model(torch.bfloat16, 421)
    opt = torch.optim.Adam(m.trainable_parameters(), lr=1e-2)
    res, _ = render_rays(m, None, _dev(rays), _dev(img), _hp(S, 0, chunk), None, None, True, True, False)
    opt.zero_grad(set_to_none=True)
    _runner_loss(res, _dev(rgbs), m.wt).backward()
    opt.step()
    m.eval()
    with torch.no_grad():
        got, _ = render_rays(m, None, _dev(rays), _dev(img), _hp(S, 0, chunk), None, None, True, True, False)
    ref_model = _model(torch.bfloat16, 421)
    ref_model.flat.copy_(m.flat)
    ref_model.refresh_compute_copies()
    ref_model.eval()
    with torch.no_grad():
        ref, _ = render_rays(ref_model, None, _dev(rays), _dev(img), _hp(S, 0, chunk), None, None, True, True, False)
    assert torch.equal(got["rgb_coarse"], ref["rgb_coarse"])


@pytest.mark.parametrize("fine", [0, 32])
def test_graph_train_runner_loop_matches_eager_bridge(fine):
    """nerf.graph_train = True: the Runner-style loop (render_rays under autograd, GradScaler, torch Adam, ExponentialLR) with the
    forward and the backward replayed from captured graphs (graph.GraphedRenderTrain) against the same loop on the eager bridge:
    same losses and the same parameters after four optimizer steps on changing ray batches (deterministic sampling)."""
    from torch.optim.lr_scheduler import ExponentialLR
    from switch_nerf_amd.rendering import render_rays
    N, S, chunk = 128, 64, 2048
    batches = [synth.make_rays(430 + i, N) for i in range(4)]
    models = [_model(torch.float32, 431), _model(torch.float32, 431)]
    models[1].graph_train = True
    losses = [[], []]
    for k, m in enumerate(models):
        opt = torch.optim.Adam(m.trainable_parameters(), lr=5e-4)
        sch = ExponentialLR(opt, gamma=0.1 ** (1 / 50))
        scaler = torch.amp.GradScaler("cuda", enabled=True, init_scale=1024.0)
        m.train()
        for rays, img, rgbs in batches:
            res, _ = render_rays(m, None, _dev(rays), _dev(img), _hp(S, fine, chunk), None, None, True, True, False)
            loss = _runner_loss(res, _dev(rgbs), m.wt)
            opt.zero_grad(set_to_none=True)
            scaler.scale(loss).backward()
            scaler.step(opt)
            scaler.update()
            sch.step()
            losses[k].append(loss.item())
    assert len(models[1]._train_graphs) == 1
    for la, lb in zip(*losses):
        assert abs(la - lb) <= 2e-5 * abs(la), losses
    d = (models[0].flat - models[1].flat).abs().max().item()
    print(f"graph_train vs eager bridge (fine={fine}): max parameter difference after 4 steps {d:.2e}")
    assert d < 2e-5
