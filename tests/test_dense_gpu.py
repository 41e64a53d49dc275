"""The dense (non-MoE) NeRF of BASELINE.json configs[0] on the HIP path: parity against the golden vectors produced by the
imported reference (oracle/gen_golden.py gen_dense) and against the CPU oracle with perturbation + sigma noise."""
import os

import numpy as np
import pytest
import torch

import synth
from oracle import switchnerf_oracle as O

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _model(dtype, seed, cfg=synth.DENSE):
    from switch_nerf_amd.dense import DenseNeRF
    m = DenseNeRF(cfg, dtype=dtype)
    m.load_state_dict(synth.make_dense_weights(seed, cfg))
    return m


def test_dense_train_step_vs_reference_golden_fp32():
    g = np.load(os.path.join(G, "dense_nerf_train.npz"))
    N, S = int(g["N"]), int(g["S"])
    m = _model(torch.float32, int(g["seed"]))
    rays, img, rgbs = synth.make_rays(162, N)
    st = m.train_step(_dev(rgbs), _dev(rays), _dev(img), S, N * S, perturb=0.0, optimizer_step=False)
    c = st["ctx"]
    np.testing.assert_allclose(c["rgb"].cpu().numpy(), g["rgb"], rtol=0, atol=1e-4)          # north-star tolerance
    np.testing.assert_allclose(c["depth"].cpu().numpy(), g["depth"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(c["raw"][:, 3].cpu().numpy().reshape(N, S)[:64], g["sigma_head"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(st["loss"].item(), float(g["loss"]), rtol=1e-5)
    worst = 0.0
    for k, t in m.grad_dict().items():
        got = t.cpu().numpy()
        ref_sum = g["gsum__" + k]
        scale = max(1e-12, float(ref_sum[1]))
        assert abs(synth.checksum(got)[0] - ref_sum[0]) <= 1e-3 * scale + 1e-9, k
        assert abs(synth.checksum(got)[1] - ref_sum[1]) <= 1e-3 * scale + 1e-9, k
        sl = got.reshape(-1)[:: max(1, got.size // 499)][:499]
        ref = g["gslice__" + k]
        worst = max(worst, float(np.abs(sl - ref).max() / (np.abs(ref).max() + 1e-12)))
        np.testing.assert_allclose(sl, ref, rtol=2e-3, atol=1e-7 + 2e-4 * np.abs(ref).max(), err_msg=k)
    print(f"dense: worst relative gradient-slice error {worst:.2e}")


def _oracle(sd, rays, img, rgbs, S, pr, noise):
    p = O.params_from_numpy(sd, requires_grad=True)
    res = O.render_rays_dense(p, torch.from_numpy(rays), torch.from_numpy(img), synth.DENSE, S, perturb=1.0,
                              perturb_rand=torch.from_numpy(pr), sigma_noise=torch.from_numpy(noise))
    loss = torch.nn.functional.mse_loss(res["rgb_coarse"], torch.from_numpy(rgbs))
    loss.backward()
    return p, res, loss


def test_dense_train_step_vs_oracle_perturbed_fp32_and_adam():
    """300 rays x 96 samples (a ragged last tile), stratified perturbation and sigma noise supplied; then one Adam step."""
    N, S = 300, 96
    sd = synth.make_dense_weights(171)
    rays, img, rgbs = synth.make_rays(172, N)
    rng = np.random.default_rng(173)
    pr = rng.uniform(0, 1, (N, S)).astype(np.float32)
    noise = rng.standard_normal((N * S, 1)).astype(np.float32)
    p, res, loss = _oracle(sd, rays, img, rgbs, S, pr, noise)
    m = _model(torch.float32, 171)
    st = m.train_step(_dev(rgbs), _dev(rays), _dev(img), S, N * S, perturb=1.0, perturb_rand=_dev(pr),
                      sigma_noise=_dev(noise.reshape(-1)), optimizer_step=True)
    np.testing.assert_allclose(st["rgb"].cpu().numpy(), res["rgb_coarse"].detach().numpy(), rtol=0, atol=1e-4)
    np.testing.assert_allclose(st["loss"].item(), loss.item(), rtol=1e-5)
    gd = m.grad_dict()
    for k, t in p.items():
        ref = t.grad.numpy()
        got = gd[k].cpu().numpy()
        tol = 2e-4 * np.abs(ref).max() + 1e-9
        np.testing.assert_allclose(got, ref, rtol=2e-3, atol=tol, err_msg=k)
    # Adam (runner.py:305-310 torch.optim.Adam defaults, lr 5e-4): first step = -lr * sign-like update
    opt = torch.optim.Adam(list(p.values()), lr=5e-4)
    opt.step()
    new = m.state_dict()
    for k, t in p.items():
        a, b = new[k].cpu().numpy(), t.detach().numpy()
        small = np.abs(t.grad.numpy()) < 1e-7 * max(1e-12, np.abs(t.grad.numpy()).max())     # update direction ill-defined there
        np.testing.assert_allclose(np.where(small, b, a), b, rtol=0, atol=2e-5, err_msg=k)


def test_dense_bf16_close_to_fp32():
    N, S = 512, 128
    rays, img, rgbs = synth.make_rays(182, N)
    outs = {}
    for dt in (torch.float32, torch.bfloat16):
        m = _model(dt, 181)
        st = m.train_step(_dev(rgbs), _dev(rays), _dev(img), S, N * S, perturb=0.0, optimizer_step=False)
        outs[dt] = (st["rgb"].float().cpu().numpy(), st["loss"].item(), {k: v.cpu().numpy() for k, v in m.grad_dict().items()})
    a, b = outs[torch.float32], outs[torch.bfloat16]
    assert np.abs(a[0] - b[0]).max() < 3e-2            # bf16 activations: the 1e-4 bar applies to the fp32 path
    assert abs(a[1] - b[1]) < 2e-2 * abs(a[1])
    for k in a[2]:
        den = np.abs(a[2][k]).max() + 1e-12
        assert np.abs(a[2][k] - b[2][k]).max() / den < 0.15, k


def test_dense_call_mirror_and_state_dict_roundtrip():
    m = _model(torch.float32, 191)
    sd = synth.make_dense_weights(191)
    out = m.state_dict()
    assert set(out) == set(sd)
    for k in sd:
        np.testing.assert_array_equal(out[k].cpu().numpy(), sd[k])
    rng = np.random.default_rng(192)
    P = 777
    x = np.concatenate([rng.uniform(-1, 1, (P, 3)), rng.standard_normal((P, 3)), rng.integers(0, synth.DENSE["appearance_count"], (P, 1))],
                       1).astype(np.float32)
    x[:, 3:6] /= np.linalg.norm(x[:, 3:6], axis=1, keepdims=True)
    m.eval()
    got = m(_dev(x)).cpu().numpy()
    p = O.params_from_numpy(sd)
    ref = O.nerf_dense_forward(p, torch.from_numpy(x), synth.DENSE).detach().numpy()
    np.testing.assert_allclose(got[:, :3], ref[:, :3], rtol=0, atol=1e-4)
    np.testing.assert_allclose(got[:, 3], ref[:, 3], rtol=1e-4, atol=1e-4)
    with pytest.raises(Exception, match="Unexpected input shape"):
        m(_dev(x[:, :5]))
