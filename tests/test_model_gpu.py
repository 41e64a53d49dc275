"""End-to-end parity of the HIP train step (forward, loss, backward) against (a) golden vectors produced by the
imported reference and (b) the CPU oracle at a larger size.  fp32 mode, tolerances written at each check."""
import os

import numpy as np
import pytest
import torch

import synth
from oracle import switchnerf_oracle as O

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def _model(dtype, seed, gate_scale, **kw):
    from switch_nerf_amd.model import SwitchNeRF
    m = SwitchNeRF(synth.BUILDING, dtype=dtype, **kw)
    m.load_state_dict(synth.make_weights(seed, synth.BUILDING, gate_scale=gate_scale))
    return m


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("tag", ["unbalanced", "balanced", "cf125_nobpr", "cf125_bpr", "cf050_bpr"])
def test_train_step_vs_reference_golden_fp32(tag):
    g = np.load(os.path.join(G, f"render_train_{tag}.npz"))
    N, S, chunk = int(g["N"]), int(g["S"]), int(g["chunk"])
    kw = {}
    if "capacity_factor" in g:       # other capacity factors (token dropping at 1.25 / 0.5) and position-order ranking
        kw = dict(capacity_factor=float(g["capacity_factor"]), batch_prioritized=bool(int(g["bpr"])))
    m = _model(torch.float32, int(g["seed"]), float(g["gate_scale"]), **kw)
    rays, img, rgbs = synth.make_rays(52, N)
    st = m.train_step(_dev(rgbs), _dev(rays), _dev(img), S, chunk, perturb=0.0, optimizer_step=False)
    c = st["ctx"]
    idx = c["idx"].cpu().numpy().reshape(N, S)
    n_mis = int((idx != g["moe_gates"]).sum())
    print(f"{tag}: routing mismatches vs reference: {n_mis} of {idx.size}; dropped {(c['tok2row'] < 0).float().mean().item():.3f}")
    assert n_mis == 0, "top-1 expert indices must equal the reference's"
    np.testing.assert_allclose(c["rgb"].cpu().numpy(), g["rgb"], rtol=0, atol=1e-4)          # north-star tolerance
    np.testing.assert_allclose(c["raw"][:, 3].cpu().numpy().reshape(N, S), g["sigma"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(c["depth_variance"].cpu().numpy(), g["depth_variance"], rtol=1e-3, atol=1e-6)
    np.testing.assert_allclose(c["l_aux"].cpu().numpy(), g["gate_loss"], rtol=1e-5)
    np.testing.assert_allclose(st["loss"].item(), float(g["loss"]), rtol=1e-5)
    gd = m.grad_dict()
    worst = 0.0
    for k, t in gd.items():
        got = t.cpu().numpy()
        ref_sum = g["gsum__" + k]
        scale = max(1e-12, float(ref_sum[1]))
        assert abs(synth.checksum(got)[0] - ref_sum[0]) <= 1e-3 * scale + 1e-9, k
        assert abs(synth.checksum(got)[1] - ref_sum[1]) <= 1e-3 * scale + 1e-9, k
        sl = got.reshape(-1)[:: max(1, got.size // 499)][:499]
        ref = g["gslice__" + k]
        tol = 1e-7 + 2e-4 * np.abs(ref).max()
        worst = max(worst, float(np.abs(sl - ref).max() / (np.abs(ref).max() + 1e-12)))
        np.testing.assert_allclose(sl, ref, rtol=2e-3, atol=tol, err_msg=k)
    print(f"{tag}: worst relative gradient-slice error {worst:.2e}")


def _oracle_step(sd, rays, img, rgbs, S, chunk, routings=None, noise=None, pr=None, perturb=0.0):
    p = O.params_from_numpy(sd, requires_grad=True)
    st = O.training_step(p, torch.from_numpy(rays), torch.from_numpy(img), torch.from_numpy(rgbs), synth.BUILDING, S, chunk,
                         routings=routings, sigma_noise=noise, perturb=perturb, perturb_rand=pr)
    st["loss"].backward()
    return p, st


@pytest.mark.parametrize("gate_scale", [1.0, 0.02])
def test_train_step_vs_oracle_larger_fp32(gate_scale):
    """256 rays x 128 samples, 8 segments of 4096 points, stratified perturbation and sigma noise supplied."""
    N, S, chunk = 256, 128, 4096
    sd = synth.make_weights(77, synth.BUILDING, gate_scale=gate_scale)
    rays, img, rgbs = synth.make_rays(78, N)
    rng = np.random.default_rng(79)
    pr = rng.uniform(0, 1, (N, S)).astype(np.float32)
    noise = rng.standard_normal((N * S, 1)).astype(np.float32)
    m = _model(torch.float32, 77, gate_scale)
    st = m.train_step(_dev(rgbs), _dev(rays), _dev(img), S, chunk, perturb=1.0, perturb_rand=_dev(pr),
                      sigma_noise=_dev(noise.reshape(-1)), optimizer_step=False)
    c = st["ctx"]
    p, ost = _oracle_step(sd, rays, img, rgbs, S, chunk, noise=torch.from_numpy(noise), pr=torch.from_numpy(pr), perturb=1.0)
    res = ost["results"]
    ref_idx = np.concatenate([r["idx"] for r in res["routings"]])
    ref_loc = np.concatenate([r["loc"] for r in res["routings"]])
    mis = (c["idx"].cpu().numpy() != ref_idx) | (c["loc"].cpu().numpy() != ref_loc)
    print(f"gate_scale {gate_scale}: routing (idx, loc) mismatches vs oracle: {int(mis.sum())} of {mis.size}")
    if mis.any():
        # fp32 summation-order differences can flip a near-tie; then re-run both sides with the HIP routing injected
        routings = []
        for s_ in range(N * S // chunk):
            sl = slice(s_ * chunk, (s_ + 1) * chunk)
            routings.append(dict(idx=c["idx"].cpu().numpy()[sl], loc=c["loc"].cpu().numpy()[sl], capacity=c["cap"]))
        assert mis.mean() < 2e-3
        p, ost = _oracle_step(sd, rays, img, rgbs, S, chunk, routings=routings, noise=torch.from_numpy(noise),
                              pr=torch.from_numpy(pr), perturb=1.0)
        res = ost["results"]
    np.testing.assert_allclose(c["rgb"].cpu().numpy(), res["rgb_coarse"].detach().numpy(), rtol=0, atol=1e-4)
    np.testing.assert_allclose(c["raw"][:, 3].cpu().numpy(), res["sigma_coarse"].detach().numpy().reshape(-1), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(st["loss"].item(), ost["loss"].item(), rtol=2e-5)
    gd = m.grad_dict()
    for k, t in gd.items():
        ref = p[k].grad.numpy()
        got = t.cpu().numpy()
        scale = np.abs(ref).max() + 1e-12
        err = np.abs(got - ref).max() / scale
        assert err <= (5e-3 if "sigma" in k else 2e-3), (k, err)      # (the sigma head's gradient is a sum of cancelling per-point terms)


def test_train_step_with_gate_noise_vs_oracle_fp32():
    """SwitchNeRF(gate_noise=1.0): a TRAINING step adds gate_noise * noise / E to the router's logits (--gate_noise, opts.py:208;
    tutel_moe_layer_nobatch.py:119-122) - against the oracle (whose gate-noise branch is pinned on the reference layer's own run,
    tests/test_oracle_golden.py) fed the same draw: routing, rgb, loss, every gradient; an evaluation forward adds none."""
    N, S, chunk, gn = 128, 64, 4096, 1.0
    sd = synth.make_weights(91, synth.BUILDING, gate_scale=1.0)
    rays, img, rgbs = synth.make_rays(92, N)
    rng = np.random.default_rng(93)
    draw = rng.standard_normal((N * S, 8)).astype(np.float32)
    m = _model(torch.float32, 91, 1.0, gate_noise=gn)
    m.gate_noise_draw = _dev(draw)
    st = m.train_step(_dev(rgbs), _dev(rays), _dev(img), S, chunk, perturb=0.0, optimizer_step=False)
    c = st["ctx"]
    p = O.params_from_numpy(sd, requires_grad=True)
    kw = dict(gate_noise=gn, gate_noise_draw=torch.from_numpy(draw))
    ost = O.training_step(p, torch.from_numpy(rays), torch.from_numpy(img), torch.from_numpy(rgbs), synth.BUILDING, S, chunk, **kw)
    res = ost["results"]
    ref_idx = np.concatenate([r["idx"] for r in res["routings"]])
    mis = c["idx"].cpu().numpy() != ref_idx
    assert mis.mean() < 2e-3, int(mis.sum())
    if mis.any():      # a near-tie flipped by summation order: both sides on the HIP routing
        routings = [dict(idx=c["idx"].cpu().numpy()[i:i + chunk], loc=c["loc"].cpu().numpy()[i:i + chunk], capacity=c["cap"])
                    for i in range(0, N * S, chunk)]
        p = O.params_from_numpy(sd, requires_grad=True)
        ost = O.training_step(p, torch.from_numpy(rays), torch.from_numpy(img), torch.from_numpy(rgbs), synth.BUILDING, S, chunk,
                              routings=routings, **kw)
        res = ost["results"]
    ost["loss"].backward()
    # the noise-free routing of the same weights differs: the draw is what this test is about
    st0 = _model(torch.float32, 91, 1.0).train_step(_dev(rgbs), _dev(rays), _dev(img), S, chunk, perturb=0.0, optimizer_step=False)
    assert (st0["ctx"]["idx"] != c["idx"]).float().mean().item() > 0.01
    np.testing.assert_allclose(c["rgb"].cpu().numpy(), res["rgb_coarse"].detach().numpy(), rtol=0, atol=1e-4)
    np.testing.assert_allclose(st["loss"].item(), ost["loss"].item(), rtol=2e-5)
    for k, t in m.grad_dict().items():
        ref = p[k].grad.numpy()
        err = np.abs(t.cpu().numpy() - ref).max() / (np.abs(ref).max() + 1e-12)
        assert err <= (5e-3 if "sigma" in k else 2e-3), (k, err)
    # evaluation: no noise
    with torch.no_grad():
        ce = m.forward_rays(_dev(rays), _dev(img), S, chunk, 0.0, None, None, training=False)
    assert torch.equal(ce["idx"], st0["ctx"]["idx"])


def test_bf16_step_close_to_fp32_and_adam_moves_loss():
    N, S, chunk = 128, 128, 4096
    rays, img, rgbs = synth.make_rays(88, N)
    m32 = _model(torch.float32, 87, 0.02)
    m16 = _model(torch.bfloat16, 87, 0.02)
    a = m32.train_step(_dev(rgbs), _dev(rays), _dev(img), S, chunk, perturb=0.0, optimizer_step=False)
    b = m16.train_step(_dev(rgbs), _dev(rays), _dev(img), S, chunk, perturb=0.0, optimizer_step=False)
    assert (a["ctx"]["rgb"] - b["ctx"]["rgb"]).abs().max().item() < 3e-2
    assert abs(a["loss"].item() - b["loss"].item()) < 2e-2 * abs(a["loss"].item())
    l0 = None
    for it in range(8):
        st = m16.train_step(_dev(rgbs), _dev(rays), _dev(img), S, chunk, perturb=0.0, optimizer_step=True)
        l0 = st["loss"].item() if l0 is None else l0
    assert st["loss"].item() < l0, "eight Adam steps on a fixed batch must reduce the loss"
    sd = m16.state_dict()
    assert set(sd.keys()) == set(synth.make_weights(1).keys())


@pytest.mark.parametrize("tag,no_batch", [("unbalanced", False), ("balanced", False), ("nobatch", True)])
def test_model_call_mirror_vs_reference_golden(tag, no_batch):
    """SwitchNeRF.__call__ == NeRFMoE.forward on 4096 points: batched/capacity path and the eval (no-batch) path."""
    g = np.load(os.path.join(G, f"model_fwd_{tag}.npz"))
    m = _model(torch.float32, int(g["seed"]), float(g["gate_scale"]), batch_prioritized=not no_batch)
    m.eval()
    m.set_no_batch(no_batch)
    noise = _dev(g["sigma_noise"]) if "sigma_noise" in g.files else None
    r = m(_dev(g["x"]), sigma_noise=noise)
    idx = r["extras"]["moe_gates"][0].cpu().numpy().reshape(-1)
    assert (idx != g["moe_gates"].reshape(-1)).sum() == 0
    np.testing.assert_allclose(r["outputs"].cpu().numpy(), g["outputs"], rtol=0, atol=1e-4)
    if "moe_loss" in g.files:
        np.testing.assert_allclose(r["extras"]["moe_loss"].cpu().numpy(), g["moe_loss"], rtol=1e-5)


def test_render_rays_mirror_keys_and_eval():
    from argparse import Namespace
    from switch_nerf_amd.rendering import render_rays
    g = np.load(os.path.join(G, "render_train_balanced.npz"))
    m = _model(torch.float32, int(g["seed"]), float(g["gate_scale"]))
    rays, img, _ = synth.make_rays(52, 64)
    hp = Namespace(coarse_samples=64, fine_samples=0, model_chunk_size=1024, perturb=1.0, use_sigma_noise=True, sigma_noise_std=1.0,
                   moe_return_gates=True, return_sigma=True)
    m.eval()          # eval: perturb = 0 and no sigma noise, like the reference (rendering.py:32, :366)
    res, bg = render_rays(m, None, _dev(rays), _dev(img), hp, None, None, True, True, False)
    assert bg is False and set(res) == {"rgb_coarse", "gate_loss_coarse", "depth_coarse", "depth_variance_coarse", "moe_gates_coarse", "sigma_coarse"}
    np.testing.assert_allclose(res["rgb_coarse"].cpu().numpy(), g["rgb"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(res["depth_coarse"].cpu().numpy(), g["depth"], rtol=1e-4, atol=1e-6)
    assert res["gate_loss_coarse"].shape == (4,) and res["moe_gates_coarse"].shape == (64, 64, 1, 1)


@pytest.mark.parametrize("tag", ["det", "perturbed"])
def test_hierarchical_train_step_vs_reference_golden_fp32(tag):
    """fine_samples = 96 on top of 64 coarse samples: coarse weights -> swn_sample_pdf -> fine pass -> swn_merge_samples ->
    compositing of the 160 merged samples, and the backward through the merge into both passes."""
    g = np.load(os.path.join(G, f"render_train_fine_{tag}.npz"))
    N, S, Fn, chunk = int(g["N"]), int(g["S"]), int(g["F"]), int(g["chunk"])
    m = _model(torch.float32, int(g["seed"]), float(g["gate_scale"]))
    rays, img, rgbs = synth.make_rays(62, N)
    kw = dict(perturb=0.0)
    if float(g["perturb"]) > 0:
        kw = dict(perturb=float(g["perturb"]), perturb_rand=_dev(g["perturb_rand"]), fine_u=_dev(g["fine_u"]))
    st = m.train_step(_dev(rgbs), _dev(rays), _dev(img), S, chunk, optimizer_step=False, fine_samples=Fn, **kw)
    c, cf = st["ctx"], st["ctx_fine"]
    mis_c = int((c["idx"].cpu().numpy().reshape(N, S) != g["moe_gates_coarse"]).sum())
    mis_f = int((cf["idx"].cpu().numpy().reshape(N, Fn) != g["moe_gates_fine"]).sum())
    print(f"{tag}: routing mismatches vs reference: coarse {mis_c}, fine {mis_f}")
    assert mis_c == 0 and mis_f == 0
    np.testing.assert_allclose(c["raw"][:, 3].cpu().numpy().reshape(N, S), g["sigma_coarse"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(cf["raw"][:, 3].cpu().numpy().reshape(N, Fn), g["sigma_fine"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(st["rgb"].cpu().numpy(), g["rgb"], rtol=0, atol=1e-4)          # north-star tolerance
    np.testing.assert_allclose(st["depth"].cpu().numpy(), g["depth"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(c["l_aux"].cpu().numpy(), g["gate_loss_coarse"], rtol=1e-5)
    np.testing.assert_allclose(cf["l_aux"].cpu().numpy(), g["gate_loss_fine"], rtol=1e-5)
    np.testing.assert_allclose(st["loss"].item(), float(g["loss"]), rtol=1e-5)
    gd = m.grad_dict()
    worst = 0.0
    for k, t in gd.items():
        got = t.cpu().numpy()
        ref_sum = g["gsum__" + k]
        scale = max(1e-12, float(ref_sum[1]))
        assert abs(synth.checksum(got)[0] - ref_sum[0]) <= 1e-3 * scale + 1e-9, k
        assert abs(synth.checksum(got)[1] - ref_sum[1]) <= 1e-3 * scale + 1e-9, k
        sl = got.reshape(-1)[:: max(1, got.size // 499)][:499]
        ref = g["gslice__" + k]
        worst = max(worst, float(np.abs(sl - ref).max() / (np.abs(ref).max() + 1e-12)))
        np.testing.assert_allclose(sl, ref, rtol=2e-3, atol=1e-7 + 2e-4 * np.abs(ref).max(), err_msg=k)
    print(f"{tag}: worst relative gradient-slice error {worst:.2e}")


def test_render_rays_mirror_hierarchical():
    from argparse import Namespace
    from switch_nerf_amd.rendering import render_rays
    g = np.load(os.path.join(G, "render_train_fine_det.npz"))
    N, S, Fn, chunk = int(g["N"]), int(g["S"]), int(g["F"]), int(g["chunk"])
    m = _model(torch.float32, int(g["seed"]), float(g["gate_scale"]))
    rays, img, _ = synth.make_rays(62, N)
    h = Namespace(coarse_samples=S, fine_samples=Fn, model_chunk_size=chunk, perturb=0.0, use_sigma_noise=False,
                  sigma_noise_std=1.0, moe_return_gates=True, return_sigma=True, use_cascade=False)
    m.train()
    res, bg = render_rays(m, None, _dev(rays), _dev(img), h, None, None, True, True, False)
    assert bg is False and "rgb_coarse" not in res
    assert res["rgb_fine"].requires_grad             # training mode: the results carry a grad_fn like the reference's (autograd.py)
    res = {k: v.detach() for k, v in res.items()}
    np.testing.assert_allclose(res["rgb_fine"].cpu().numpy(), g["rgb"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(res["depth_variance_fine"].cpu().numpy(), g["depth_variance"], rtol=1e-3, atol=1e-6)
    np.testing.assert_allclose(res["gate_loss_fine"].cpu().numpy(), g["gate_loss_fine"], rtol=1e-5)
    assert np.array_equal(res["moe_gates_fine"].cpu().numpy().reshape(N, Fn), g["moe_gates_fine"])


def test_hierarchical_bf16_runs_at_odd_sizes():
    """bf16, 128 coarse + 192 fine samples (merged 320 -> padded to 512 in the sorter), stratified noise: loss falls."""
    N, S, Fn, chunk = 64, 128, 192, 4096
    rays, img, rgbs = synth.make_rays(98, N)
    m = _model(torch.bfloat16, 97, 0.02)
    first = last = None
    for it in range(6):
        st = m.train_step(_dev(rgbs), _dev(rays), _dev(img), S, chunk, perturb=1.0,
                          perturb_rand=torch.rand(N, S, device="cuda"), fine_samples=Fn)
        z = st["ctx_fine"]["z"]
        assert torch.isfinite(st["loss"]).item()
        first = st["loss"].item() if first is None else first
        last = st["loss"].item()
    assert last < first


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_expert_parallel_path_single_rank_equals_local_experts(dtype, monkeypatch):
    """The expert-parallel code path (send buffer in payload order, counts exchange, local-expert groups, return path,
    remapped combine index) with world = 1, where the exchange is the identity: must reproduce the default path - forward
    and loss bit for bit, every gradient up to the summation-order noise of the fp32 atomics that both paths share.
    (World > 1 plumbing: tests/test_parallel_cpu.py, gloo.)  The local path runs its tail as the 64-row launch the expert-parallel
    path uses (SWN_FUSED_TAIL=0: inside the expert launch - local experts only - layer "1" sums in another fp32 order)."""
    from switch_nerf_amd.parallel import ExpertParallel
    monkeypatch.setenv("SWN_FUSED_TAIL", "0")
    N, S, chunk = 128, 64, 2048
    rays, img, rgbs = synth.make_rays(118, N)
    outs = []
    for use_ep in (False, True):
        m = _model(dtype, 117, 1.0)
        if use_ep:
            m.set_expert_parallel(ExpertParallel(0, 1, m.E))
        st = m.train_step(_dev(rgbs), _dev(rays), _dev(img), S, chunk, perturb=0.0, optimizer_step=False)
        outs.append((st["rgb"].clone(), st["loss"].clone(), m.grad.clone()))
    assert torch.equal(outs[0][0], outs[1][0])
    assert torch.equal(outs[0][1], outs[1][1])
    g0, g1 = outs[0][2], outs[1][2]
    assert g0.abs().sum().item() > 0
    for name, (off, shape) in m.spec.items():
        n = int(np.prod(shape))
        a, b = g0[off:off + n], g1[off:off + n]
        assert (a - b).abs().max().item() <= 1e-5 * max(a.abs().max().item(), 1e-12), name


@pytest.mark.parametrize("cf,chunk", [(1.0, 2048), (0.5, 4096), (1.25, 2048)])
def test_expert_parallel_owner_tail_single_rank_equals_fused_local_path(cf, chunk):
    """ExpertParallel(owner_tail=True) (ep_owner.py: the dense tail on the expert's rank - the fused launches on a received token space,
    raw + the sign bits of h2 home, d_raw out, dx + the gate gradient home) with world = 1, where every exchange is a copy: the forward
    is BIT-identical to the local fused path (the same rows in the same tiles: raw, rgb, loss), the per-ray bias gradient is formed on the
    source from the returned sign bits, and every parameter gradient agrees to summation order - with dropped tokens (capacity factor
    0.5), spare capacity (1.25: ragged groups) and two optimizer steps."""
    from switch_nerf_amd.parallel import ExpertParallel
    N, S = 128, 64
    rays, img, rgbs = synth.make_rays(171, N)
    pr = torch.rand(N, S, generator=torch.Generator().manual_seed(9)).cuda()
    noise = torch.randn(N * S, generator=torch.Generator().manual_seed(10)).cuda()
    outs = []
    for use_ep in (False, True):
        m = _model(torch.bfloat16, 170, 1.0, capacity_factor=cf)
        if use_ep:
            m.set_expert_parallel(ExpertParallel(0, 1, m.E, owner_tail=True))
        st = m.train_step(_dev(rgbs), _dev(rays), _dev(img), S, chunk, perturb=1.0, perturb_rand=pr, sigma_noise=noise, optimizer_step=False)
        c = st["ctx"]
        assert (c.get("ep_owner") is not None) == use_ep and c["tail_fused"]
        g0 = m.grad.clone()
        st2 = m.train_step(_dev(rgbs), _dev(rays), _dev(img), S, chunk, perturb=1.0, perturb_rand=pr, sigma_noise=noise)      # with Adam
        st3 = m.train_step(_dev(rgbs), _dev(rays), _dev(img), S, chunk, perturb=1.0, perturb_rand=pr, sigma_noise=noise, optimizer_step=False)
        outs.append((c["raw"].clone(), st["rgb"].clone(), st["loss"].clone(), g0, st3["loss"].clone(), int((c["loc"] >= c["cap"]).sum())))
    a, b = outs
    assert (a[5] > 0) == (cf < 1.25 or a[5] > 0) and a[5] == b[5]
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    assert a[3].abs().sum().item() > 0
    for name, (off, shape) in m.spec.items():
        n = int(np.prod(shape))
        x, y = a[3][off:off + n], b[3][off:off + n]
        assert (x - y).abs().max().item() <= 2e-5 * max(x.abs().max().item(), 1e-12), name
    assert abs(a[4].item() - b[4].item()) <= 5e-5 * abs(a[4].item())      # after one Adam step on those gradients


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_expert_parallel_padded_mode_is_host_free_and_graph_capturable(dtype, monkeypatch):
    """ExpertParallel(padded=True): the reference's capacity-padded equal-split exchange (tutel_moe_layer_nobatch.py:157) - standard row
    spaces, nothing read on the host.  World = 1: (a) forward / loss bit-identical to the default path and gradients to summation order,
    unbalanced routing with dropped tokens; (b) the whole step captured into hipGraphs (graph.GraphedTrainStep: a host read of split
    sizes would abort the capture) replays bit-identically to the eager padded step over three optimizer steps."""
    from switch_nerf_amd.graph import GraphedTrainStep
    from switch_nerf_amd.parallel import ExpertParallel
    monkeypatch.setenv("SWN_FUSED_TAIL", "0")      # (the default path with its tail as the 64-row launch the expert-parallel path uses)
    N, S, chunk = 128, 64, 2048
    rays, img, rgbs = synth.make_rays(128, N)
    outs = []
    for mode in ("default", "padded"):
        m = _model(dtype, 127, 1.0)
        if mode == "padded":
            ep = ExpertParallel(0, 1, m.E, padded=True)
            assert ep.capturable and ep.use_padded(1 << 40)
            m.set_expert_parallel(ep)
        st = m.train_step(_dev(rgbs), _dev(rays), _dev(img), S, chunk, perturb=0.0, optimizer_step=False)
        assert st["ctx"].get("ep_padded", None) in (None, True)
        outs.append((st["rgb"].clone(), st["loss"].clone(), m.grad.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    for name, (off, shape) in m.spec.items():
        n = int(np.prod(shape))
        a, b = outs[0][2][off:off + n], outs[1][2][off:off + n]
        assert (a - b).abs().max().item() <= 1e-5 * max(a.abs().max().item(), 1e-12), name
    auto = ExpertParallel(0, 1, m.E, padded="auto")
    assert auto.use_padded(1 << 20) and not auto.use_padded(1 << 30)
    with pytest.raises(ValueError, match="cannot be captured"):
        mp = _model(dtype, 127, 1.0)
        mp.set_expert_parallel(ExpertParallel(0, 1, mp.E))          # unequal splits: host-sized
        GraphedTrainStep(mp, _dev(rgbs), _dev(rays), _dev(img), S, chunk, perturb=0.0, noise_std=0.0)
    if dtype == torch.float32:
        return
    ma, mb = _model(dtype, 129, 1.0), _model(dtype, 129, 1.0)
    for mm in (ma, mb):
        mm.set_expert_parallel(ExpertParallel(0, 1, mm.E, padded=True))
    step = GraphedTrainStep(ma, _dev(rgbs), _dev(rays), _dev(img), S, chunk, perturb=0.0, noise_std=0.0)
    for it in range(3):
        ra = step(_dev(rgbs), _dev(rays), _dev(img))
        rb = mb.train_step(_dev(rgbs), _dev(rays), _dev(img), S, chunk, perturb=0.0)
        assert torch.equal(ra["loss"], rb["loss"]), it
    assert torch.equal(ma.flat, mb.flat) and ma.step_count == mb.step_count == 3


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_expert_parallel_path_512_wide_equals_local_experts(dtype):
    """the expert-parallel path keeps the received rows PACKED (groups addressed through ep_begin); with 512-feature
    experts (mission_bay.yaml widths) the expert weight gradients must honour that packing too (ops.wgrad_multi cuts the 512-wide
    operands into 256-column GEMMs of the balanced launch, group_begin included).  World = 1 against the default path, unbalanced
    routing (gate_scale 1: the groups are far from full, so packed rows and capacity slots differ)."""
    from switch_nerf_amd.model import SwitchNeRF
    from switch_nerf_amd.parallel import ExpertParallel
    cfg = dict(synth.BUILDING, model_dim=512, gate_hidden=512, num_experts=16)
    N, S, chunk = 64, 64, 2048
    sd = synth.make_weights(161, cfg, gate_scale=1.0)
    rays, img, rgbs = synth.make_rays(162, N)
    outs = []
    for use_ep in (False, True):
        m = SwitchNeRF(cfg, dtype=dtype)
        m.load_state_dict(sd)
        if use_ep:
            m.set_expert_parallel(ExpertParallel(0, 1, m.E))
        st = m.train_step(_dev(rgbs), _dev(rays), _dev(img), S, chunk, perturb=0.0, optimizer_step=False)
        kept = int(st["ctx"]["counts"].clamp(max=st["ctx"]["cap"]).sum().item())
        assert 0 < kept < N * S                       # tokens are dropped and groups are ragged: the layouts really differ
        outs.append((st["rgb"].clone(), st["loss"].clone(), m.grad.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    g0, g1 = outs[0][2], outs[1][2]
    for name, (off, shape) in m.spec.items():
        n = int(np.prod(shape))
        a, b = g0[off:off + n], g1[off:off + n]
        assert (a - b).abs().max().item() <= 1e-5 * max(a.abs().max().item(), 1e-12), name
    off, shape = m.spec["exp3.w"]
    assert g0[off:off + int(np.prod(shape))].abs().sum().item() > 0


@pytest.mark.parametrize("tag", ["det", "perturbed"])
def test_mip_train_step_vs_reference_golden_fp32(tag):
    """Two-level mip training step (frustum casting + integrated encoding, weights blur + resampling, colour padding,
    loss = (fine + coarse) / 2) against the reference's MipNeRFMoE run: routing of both levels exact, rgb <= 1e-4, all grads."""
    g = np.load(os.path.join(G, f"mip_train_{tag}.npz"))
    N, S, Fn, chunk = int(g["N"]), int(g["S"]), int(g["F"]), int(g["chunk"])
    m = _model(torch.float32, int(g["seed"]), float(g["gate_scale"]))
    rays, img, rgbs = synth.make_rays(72, N)
    kw = dict(perturb=0.0)
    if float(g["perturb"]) > 0:
        kw = dict(perturb=float(g["perturb"]), perturb_rand=_dev(g["perturb_rand"]), fine_u=_dev(g["fine_u"]))
    st = m.train_step_mip(_dev(rgbs), _dev(rays), _dev(g["radii"]), _dev(img), S, Fn, chunk, optimizer_step=False, **kw)
    c, cf = st["ctx"], st["ctx_fine"]
    mis_c = int((c["idx"].cpu().numpy().reshape(N, S - 1) != g["moe_gates_coarse"]).sum())
    mis_f = int((cf["idx"].cpu().numpy().reshape(N, Fn - 1) != g["moe_gates_fine"]).sum())
    print(f"mip {tag}: routing mismatches vs reference: coarse {mis_c}, fine {mis_f}")
    assert mis_c == 0 and mis_f == 0
    np.testing.assert_allclose(c["rgb"].cpu().numpy(), g["rgb_coarse"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(cf["rgb"].cpu().numpy(), g["rgb_fine"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(cf["depth_variance"].cpu().numpy(), g["depth_variance"], rtol=1e-3, atol=1e-6)
    np.testing.assert_allclose(c["l_aux"].cpu().numpy(), g["gate_loss_coarse"], rtol=1e-5)
    np.testing.assert_allclose(cf["l_aux"].cpu().numpy(), g["gate_loss_fine"], rtol=1e-5)
    np.testing.assert_allclose(st["loss"].item(), float(g["loss"]), rtol=1e-5)
    gd = m.grad_dict()
    worst = 0.0
    for k, t in gd.items():
        got = t.cpu().numpy()
        ref_sum = g["gsum__" + k]
        scale = max(1e-12, float(ref_sum[1]))
        assert abs(synth.checksum(got)[0] - ref_sum[0]) <= 1e-3 * scale + 1e-9, k
        assert abs(synth.checksum(got)[1] - ref_sum[1]) <= 1e-3 * scale + 1e-9, k
        sl = got.reshape(-1)[:: max(1, got.size // 499)][:499]
        ref = g["gslice__" + k]
        worst = max(worst, float(np.abs(sl - ref).max() / (np.abs(ref).max() + 1e-12)))
        np.testing.assert_allclose(sl, ref, rtol=2e-3, atol=1e-7 + 2e-4 * np.abs(ref).max(), err_msg=k)
    print(f"mip {tag}: worst relative gradient-slice error {worst:.2e}")


def test_mip_bf16_step_runs_and_learns():
    N, S, Fn, chunk = 64, 129, 129, 4096
    rays, img, rgbs = synth.make_rays(128, N)
    radii = torch.full((N, 1), 1e-3, device="cuda")
    m = _model(torch.bfloat16, 127, 0.02)
    first = last = None
    for it in range(6):
        st = m.train_step_mip(_dev(rgbs), _dev(rays), radii, _dev(img), S, Fn, chunk, perturb=1.0,
                              perturb_rand=torch.rand(N, S, device="cuda"))
        assert torch.isfinite(st["loss"]).item()
        first = st["loss"].item() if first is None else first
        last = st["loss"].item()
    assert last < first


def test_checkpoint_layouts_round_trip_through_the_model():
    """load_state_dict takes the DDP-prefixed expertmlp layout and the seqexperts layout of the reference's evaluation;
    both give the same model (identical render), and state_dict(layout=...) writes them back exactly."""
    from switch_nerf_amd.model import SwitchNeRF
    sd = synth.make_weights(131, synth.BUILDING)
    rays, img, _ = synth.make_rays(132, 32)
    outs = []
    m0 = _model(torch.float32, 131, 1.0)
    variants = [sd, {"module." + k: v for k, v in sd.items()}, m0.state_dict(layout="seqexperts", prefix="module.")]
    for v in variants:
        m = SwitchNeRF(synth.BUILDING, dtype=torch.float32)
        m.load_state_dict(v)
        c = m.forward_rays(_dev(rays), _dev(img), 64, 2048)
        outs.append(c["rgb"].clone())
        back = m.state_dict()
        for k, ref in sd.items():
            assert torch.equal(back[k].cpu(), torch.from_numpy(ref)), k
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


def test_sixteen_experts_vs_oracle_fp32():
    """16 experts (the expert count of the Mission Bay recipe) at building widths: routing, render and gradients vs the oracle."""
    cfg = dict(synth.BUILDING, num_experts=16)
    N, S, chunk = 64, 64, 2048
    sd = synth.make_weights(141, cfg, gate_scale=0.05)
    rays, img, rgbs = synth.make_rays(142, N)
    from switch_nerf_amd.model import SwitchNeRF
    m = SwitchNeRF(cfg, dtype=torch.float32)
    m.load_state_dict(sd)
    st = m.train_step(_dev(rgbs), _dev(rays), _dev(img), S, chunk, perturb=0.0, optimizer_step=False)
    c = st["ctx"]
    p = O.params_from_numpy(sd, requires_grad=True)
    ost = O.training_step(p, torch.from_numpy(rays), torch.from_numpy(img), torch.from_numpy(rgbs), cfg, S, chunk)
    ost["loss"].backward()
    res = ost["results"]
    ref_idx = np.concatenate([r["idx"] for r in res["routings"]])
    mis = int((c["idx"].cpu().numpy() != ref_idx).sum())
    print(f"E=16: routing mismatches vs oracle {mis} of {ref_idx.size}; experts used {len(np.unique(ref_idx))}")
    assert mis == 0 and len(np.unique(ref_idx)) >= 12
    np.testing.assert_allclose(c["rgb"].cpu().numpy(), res["rgb_coarse"].detach().numpy(), rtol=0, atol=1e-4)
    np.testing.assert_allclose(st["loss"].item(), ost["loss"].item(), rtol=2e-5)
    for k, t in m.grad_dict().items():
        ref = p[k].grad.numpy()
        err = np.abs(t.cpu().numpy() - ref).max() / (np.abs(ref).max() + 1e-12)
        assert err <= (2e-3 if ref.size > 4 else 1e-2), (k, err)       # scalar biases: a cancelling sum over all points


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_mission_bay_widths_vs_oracle(dtype):
    """512-wide layers and 16 experts (mission_bay.yaml's model block) through the 512-feature chain / weight-gradient kernels:
    fp32 against the oracle (routing exact, rgb 1e-4, all gradients); bf16 against the fp32 run."""
    cfg = dict(synth.BUILDING, model_dim=512, gate_hidden=512, num_experts=16)
    N, S, chunk = 32, 64, 2048
    sd = synth.make_weights(151, cfg, gate_scale=0.05)
    rays, img, rgbs = synth.make_rays(152, N)
    from switch_nerf_amd.model import SwitchNeRF
    m = SwitchNeRF(cfg, dtype=dtype)
    m.load_state_dict(sd)
    st = m.train_step(_dev(rgbs), _dev(rays), _dev(img), S, chunk, perturb=0.0, optimizer_step=False)
    c = st["ctx"]
    p = O.params_from_numpy(sd, requires_grad=True)
    ost = O.training_step(p, torch.from_numpy(rays), torch.from_numpy(img), torch.from_numpy(rgbs), cfg, S, chunk)
    ost["loss"].backward()
    res = ost["results"]
    if dtype == torch.bfloat16:
        assert (c["rgb"].cpu() - res["rgb_coarse"].detach()).abs().max().item() < 4e-2
        assert abs(st["loss"].item() - ost["loss"].item()) < 3e-2 * abs(ost["loss"].item())
        return
    ref_idx = np.concatenate([r["idx"] for r in res["routings"]])
    mis = int((c["idx"].cpu().numpy() != ref_idx).sum())
    print(f"512-wide, E=16: routing mismatches vs oracle {mis} of {ref_idx.size}")
    assert mis == 0
    np.testing.assert_allclose(c["rgb"].cpu().numpy(), res["rgb_coarse"].detach().numpy(), rtol=0, atol=1e-4)
    np.testing.assert_allclose(st["loss"].item(), ost["loss"].item(), rtol=2e-5)
    for k, t in m.grad_dict().items():
        ref = p[k].grad.numpy()
        err = np.abs(t.cpu().numpy() - ref).max() / (np.abs(ref).max() + 1e-12)
        assert err <= (2e-3 if ref.size > 4 else 1e-2), (k, err)


def test_ragged_last_chunk_inference_vs_oracle_fp32():
    """Evaluation batches whose point count is not a multiple of model_chunk_size: the last chunk is routed on its own with
    its own capacity, like the reference's `range(0, B, model_chunk_size)` loop (rendering.py:354-383).  50 rays x 64
    samples = 3 chunks of 1024 + 128 points; both the capacity (token-dropping) mode and the no-batch mode."""
    from switch_nerf_amd import rendering
    from argparse import Namespace
    N, S, chunk = 50, 64, 1024
    sd = synth.make_weights(131, synth.BUILDING, gate_scale=0.02)
    rays, img, _ = synth.make_rays(132, N)
    m = _model(torch.float32, 131, 0.02)
    m.eval()
    p = O.params_from_numpy(sd)
    with torch.no_grad():
        ref = O.render_rays(p, torch.from_numpy(rays), torch.from_numpy(img), synth.BUILDING, S, chunk)
    h = Namespace(coarse_samples=S, fine_samples=0, model_chunk_size=chunk, perturb=1.0, use_sigma_noise=True, sigma_noise_std=1.0,
                  use_cascade=False, moe_return_gates=True, return_sigma=True)
    res, _ = rendering.render_rays(m, None, _dev(rays), _dev(img), h, None, None, True, True, False)
    idx_ref = np.concatenate([r["idx"] for r in ref["routings"]])
    np.testing.assert_array_equal(res["moe_gates_coarse"].cpu().numpy().reshape(-1), idx_ref)
    np.testing.assert_allclose(res["rgb_coarse"].cpu().numpy(), ref["rgb_coarse"].numpy(), rtol=0, atol=1e-4)
    np.testing.assert_allclose(res["sigma_coarse"].cpu().numpy(), ref["sigma_coarse"].numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(res["gate_loss_coarse"].cpu().numpy(), ref["gate_loss_coarse"].numpy(), rtol=1e-5)
    assert res["gate_loss_coarse"].numel() == 4


@pytest.mark.parametrize("fine", [0, 24])
def test_ragged_last_chunk_training_vs_oracle_fp32(fine):
    """Training batches whose point count is not a multiple of model_chunk_size (rendering.py:354-383 trains any batch size): 40 rays
    x 64 samples = 2.5 chunks of 1024 points - the half chunk is routed and back-propagated on its own with its own capacity (64 per
    expert instead of 128) and carries a third of the l_aux gradient; rays are cut by the chunk boundaries (per-ray bias gradient
    through the rows' ray index).  fine = 24: the fine pass (960 points) is one short chunk.  Routing exact, rgb 1e-4, loss, every
    gradient against the oracle; then the same step through rendering.render_rays under autograd."""
    N, S, chunk = 40, 64, 1024
    sd = synth.make_weights(141, synth.BUILDING, gate_scale=0.05)
    rays, img, rgbs = synth.make_rays(142, N)
    rng = np.random.default_rng(143)
    pr = rng.uniform(0, 1, (N, S)).astype(np.float32)
    noise = rng.standard_normal((N * S, 1)).astype(np.float32)
    kw_h, kw_o = {}, {}
    if fine:
        u = rng.uniform(0, 1, (N, fine)).astype(np.float32)
        noise_f = rng.standard_normal((N * fine, 1)).astype(np.float32)
        kw_h = dict(fine_samples=fine, fine_u=_dev(u), sigma_noise_fine=_dev(noise_f.reshape(-1)))
        kw_o = dict(fine_samples=fine, fine_u=torch.from_numpy(u), sigma_noise_fine=torch.from_numpy(noise_f))
    m = _model(torch.float32, 141, 0.05)
    st = m.train_step(_dev(rgbs), _dev(rays), _dev(img), S, chunk, perturb=1.0, perturb_rand=_dev(pr), sigma_noise=_dev(noise.reshape(-1)),
                      optimizer_step=False, **kw_h)
    c = st["ctx"]
    assert c["n_seg"] == 3 and c["parts"][1]["cap"] == 64 and c["parts"][0]["cap"] == 128
    p = O.params_from_numpy(sd, requires_grad=True)
    ost = O.training_step(p, torch.from_numpy(rays), torch.from_numpy(img), torch.from_numpy(rgbs), synth.BUILDING, S, chunk,
                          sigma_noise=torch.from_numpy(noise), perturb=1.0, perturb_rand=torch.from_numpy(pr), **kw_o)
    ost["loss"].backward()
    res = ost["results"]
    ref_idx = np.concatenate([r["idx"] for r in res["routings"]])
    assert int((c["idx"].cpu().numpy() != ref_idx).sum()) == 0
    key = "rgb_fine" if fine else "rgb_coarse"
    np.testing.assert_allclose(st["rgb"].cpu().numpy(), res[key].detach().numpy(), rtol=0, atol=1e-4)
    np.testing.assert_allclose(c["l_aux"].cpu().numpy(), res["gate_loss_coarse"].detach().numpy(), rtol=1e-5)
    np.testing.assert_allclose(st["loss"].item(), ost["loss"].item(), rtol=2e-5)
    for k, t in m.grad_dict().items():
        ref = p[k].grad.numpy()
        err = np.abs(t.cpu().numpy() - ref).max() / (np.abs(ref).max() + 1e-12)
        assert err <= 2e-3, (k, err)
    if not fine:        # the reference-style loop: render_rays under autograd on the same ragged batch gives the same gradient
        from switch_nerf_amd import rendering
        from argparse import Namespace
        g_ref = m.grad.clone()
        h = Namespace(coarse_samples=S, fine_samples=0, model_chunk_size=chunk, perturb=0.0, use_sigma_noise=False, sigma_noise_std=0.0,
                      use_cascade=False)
        m.train()
        st0 = m.train_step(_dev(rgbs), _dev(rays), _dev(img), S, chunk, perturb=0.0, optimizer_step=False)
        g_ref = m.grad.clone()
        out, _ = rendering.render_rays(m, None, _dev(rays), _dev(img), h, None, None, True, True, False)
        loss = ((out["rgb_coarse"] - _dev(rgbs)) ** 2).mean() + m.wt * out["gate_loss_coarse"].mean()
        m.flat_param.grad = None
        loss.backward()
        assert out["gate_loss_coarse"].numel() == 3
        np.testing.assert_allclose(m.flat_param.grad.cpu().numpy(), g_ref.cpu().numpy(), rtol=1e-5, atol=1e-9)


def test_inference_forward_skips_saves_and_matches_training_forward():
    """training=False: the chains write no activation saves / ReLU masks; the outputs are bit-identical to the training
    forward, and a backward through such a context is refused."""
    N, S, chunk = 64, 64, 1024
    rays, img, _ = synth.make_rays(141, N)
    for dt in (torch.float32, torch.bfloat16):
        m = _model(dt, 142, 0.02)
        a = m.forward_rays(_dev(rays), _dev(img), S, chunk, training=True)
        raw_a, rgb_a = a["raw"].clone(), a["rgb"].clone()
        b = m.forward_rays(_dev(rays), _dev(img), S, chunk, training=False)
        assert torch.equal(raw_a, b["raw"]) and torch.equal(rgb_a, b["rgb"]) and torch.equal(a["idx"], b["idx"])
        with pytest.raises(RuntimeError, match="inference forward"):
            m.backward(b, torch.zeros(N, 3, device="cuda"), torch.zeros(4, device="cuda"))
    from switch_nerf_amd.dense import DenseNeRF
    d = DenseNeRF(synth.DENSE, dtype=torch.float32)
    d.load_state_dict(synth.make_dense_weights(143))
    a = d.forward_rays(_dev(rays), _dev(img), S, N * S, training=True)
    raw_a = a["raw"].clone()
    b = d.forward_rays(_dev(rays), _dev(img), S, N * S, training=False)
    assert torch.equal(raw_a, b["raw"])


def test_module_surface_used_by_the_runner():
    """Runner.set_no_batch walks nerf.modules() for `moe_no_batch` (runner.py:946-951); count_parameters sums
    p.numel() over nerf.parameters() (3.947 M for building.yaml)."""
    m = _model(torch.float32, 151, 1.0)
    for net in m.modules():
        if hasattr(net, "moe_no_batch"):
            net.moe_no_batch = True
    assert m.moe_no_batch is True
    n = sum(p.numel() for p in m.parameters())
    sd = synth.make_weights(151, synth.BUILDING)
    assert n == sum(v.size for v in sd.values())
    assert {k for k, _ in m.named_parameters()} == set(sd)
    assert m.to("cuda") is m and m.train(False).training is False and m.eval().training is False


@pytest.mark.parametrize("case", ["all_to_one_expert", "tiny_batch", "single_ray"])
def test_edge_cases_vs_oracle_fp32(case):
    """Collisions: every token picks the same expert (7/8 of the capacity slots stay empty, 7/8 of the tokens are dropped);
    a batch smaller than a kernel tile (2 rays x 32 samples, capacity 8); one ray."""
    sd = synth.make_weights(161, synth.BUILDING, gate_scale=0.02)
    if case == "all_to_one_expert":
        N, S, chunk = 32, 64, 1024
        sd["layers.0.gates.0.wg.weight"] = np.zeros_like(sd["layers.0.gates.0.wg.weight"])
        # LayerNorm output = bias when its weight is 0: a constant gate input -> identical logits for every token, expert 5 largest
        sd["layers.gate_input_norm.weight"] = np.zeros_like(sd["layers.gate_input_norm.weight"])
        sd["layers.gate_input_norm.bias"] = np.ones_like(sd["layers.gate_input_norm.bias"])
        sd["layers.0.gates.0.wg.weight"][5] = 0.01
    elif case == "tiny_batch":
        N, S, chunk = 2, 32, 64
    else:
        N, S, chunk = 1, 64, 64
    rays, img, rgbs = synth.make_rays(162, N)
    p = O.params_from_numpy(sd, requires_grad=True)
    st = O.training_step(p, torch.from_numpy(rays), torch.from_numpy(img), torch.from_numpy(rgbs), synth.BUILDING, S, chunk)
    st["loss"].backward()
    from switch_nerf_amd.model import SwitchNeRF
    m = SwitchNeRF(synth.BUILDING, dtype=torch.float32)
    m.load_state_dict(sd)
    out = m.train_step(_dev(rgbs), _dev(rays), _dev(img), S, chunk, perturb=0.0, optimizer_step=False)
    c = out["ctx"]
    idx_ref = np.concatenate([r["idx"] for r in st["results"]["routings"]])
    if case == "all_to_one_expert":
        assert (idx_ref == 5).all()
        # all gate values tie: BPR ranks ties in token order (stable), so the first C tokens of each chunk are kept
        kept = (c["tok2row"] >= 0).cpu().numpy().reshape(-1, chunk)
        cap = chunk // 8
        assert (kept[:, :cap]).all() and not kept[:, cap:].any()
    np.testing.assert_array_equal(c["idx"].cpu().numpy(), idx_ref)
    np.testing.assert_allclose(c["rgb"].cpu().numpy(), st["results"]["rgb_coarse"].detach().numpy(), rtol=0, atol=1e-4)
    np.testing.assert_allclose(out["loss"].item(), st["loss"].item(), rtol=1e-5)
    gd = m.grad_dict()
    for k, t in p.items():
        ref = t.grad.numpy()
        got = gd[k].cpu().numpy()
        np.testing.assert_allclose(got, ref, rtol=2e-3, atol=(25 if "sigma" in k else 1) * 2e-4 * np.abs(ref).max() + 1e-9, err_msg=k)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_training_reduces_the_loss(dtype):
    """Many consecutive train steps (forward, backward, Adam, compute-copy refresh) on a learnable synthetic target: the
    colour of a ray is a smooth function of its direction.  The photometric loss must fall well below its starting value
    and every expert that receives tokens must keep finite parameters."""
    from switch_nerf_amd.model import SwitchNeRF
    N, S, chunk = 1024, 64, 8192
    rays, img, _ = synth.make_rays(171, N)
    d = rays[:, 3:6]
    rgbs = np.stack([0.5 + 0.4 * d[:, 0], 0.5 + 0.4 * d[:, 1] * d[:, 2], 0.5 - 0.4 * d[:, 2]], 1).astype(np.float32)
    m = SwitchNeRF(synth.BUILDING, dtype=dtype, lr=5e-4, seed=3)
    r, i, c = _dev(rays), _dev(img), _dev(rgbs)
    torch.manual_seed(0)
    losses = []
    for step in range(80):
        pr = torch.rand(N, S, device="cuda")
        out = m.train_step(c, r, i, S, chunk, perturb=1.0, perturb_rand=pr)
        losses.append(out["photo_loss"].item())
    first, last = np.mean(losses[:3]), np.mean(losses[-5:])
    print(f"{dtype}: photo loss {first:.4f} -> {last:.4f}")
    assert last < 0.35 * first, (first, last)
    assert torch.isfinite(m.flat).all() and m.step_count == 80


def test_checkpoint_file_roundtrip_with_adam_state(tmp_path):
    """Runner._save_checkpoint format (runner.py:2799-2818): parameters + torch.optim.Adam state in the reference's
    parameter order.  A fresh pair of models restored from the file continues bit-identically; the saved optimizer state
    loads into a real torch.optim.Adam over tensors in the reference order and its next step equals ours."""
    from switch_nerf_amd import checkpoint
    from switch_nerf_amd.model import SwitchNeRF
    from switch_nerf_amd.dense import DenseNeRF
    from switch_nerf_amd.background import BackgroundScene
    N, S, chunk = 128, 64, 2048
    rays, img, rgbs = synth.make_bg_rays(181, N)
    C_, R_ = synth.SPHERE_CENTER, synth.SPHERE_RADIUS

    def fresh(seed):
        return SwitchNeRF(synth.BUILDING, dtype=torch.float32, seed=seed), DenseNeRF(synth.DENSE_BG, dtype=torch.float32, seed=seed + 1)
    m, b = fresh(5)
    scene = BackgroundScene(m, b, C_, R_)
    for _ in range(3):
        scene.train_step(_dev(rgbs), _dev(rays), _dev(img), S, chunk, perturb=0.0)
    path = str(tmp_path / "3.pt")
    checkpoint.save_checkpoint(path, m, b, iteration=3)
    ck = torch.load(path, map_location="cpu")
    assert set(ck) >= {"model_state_dict", "bg_model_state_dict", "optimizers", "iteration"} and ck["iteration"] == 3
    assert all(k.startswith("module.") for k in ck["model_state_dict"])
    m2, b2 = fresh(77)
    assert checkpoint.load_checkpoint(path, m2, b2) == 3
    assert torch.equal(m2.flat, m.flat) and torch.equal(m2.m, m.m) and torch.equal(m2.v, m.v) and m2.step_count == 3
    assert torch.equal(b2.flat, b.flat) and torch.equal(b2.v, b.v) and b2.step_count == 3
    o1 = scene.train_step(_dev(rgbs), _dev(rays), _dev(img), S, chunk, perturb=0.0)
    o2 = BackgroundScene(m2, b2, C_, R_).train_step(_dev(rgbs), _dev(rays), _dev(img), S, chunk, perturb=0.0)
    assert torch.equal(o1["rgb"], o2["rgb"]) and o1["loss"].item() == o2["loss"].item()
    np.testing.assert_allclose(m2.flat.cpu().numpy(), m.flat.cpu().numpy(), rtol=0, atol=1e-7)   # (atomics order in the weight gradients)
    # the reference side: torch.optim.Adam over the parameters in module order, state loaded from the file
    m3, _ = fresh(5)
    checkpoint.load_checkpoint(path, m3)
    sd = {k: v.clone() for k, v in m3.state_dict().items()}
    order = checkpoint.param_order(sd.keys())
    params = [torch.nn.Parameter(sd[k].cpu()) for k in order]
    opt = torch.optim.Adam(params, lr=m3.lr)
    osd = opt.state_dict()
    osd.update(ck["optimizers"]["nerf"])
    opt.load_state_dict(osd)
    m3.train_step(_dev(rgbs), _dev(rays), _dev(img), S, chunk, perturb=0.0, optimizer_step=False)      # gradients only
    gd = m3.grad_dict()
    for k, p in zip(order, params):
        p.grad = gd[k].cpu().clone()
    opt.step()
    ops_adam = m3
    ops_adam.step_count += 1
    from switch_nerf_amd import ops as O_
    O_.adam_step(m3.flat, m3.grad, m3.m, m3.v, None, m3.step_count, m3.lr)
    new = m3.state_dict()
    for k, p in zip(order, params):
        np.testing.assert_allclose(new[k].cpu().numpy(), p.detach().numpy(), rtol=0, atol=2e-6, err_msg=k)


def test_render_image_rays_loop_with_ragged_batches():
    """Runner.render_image's pixel-batch loop: 1000 rays in batches of 384 (last batch 232 rays, its last model chunk ragged),
    eval mode, against the oracle evaluated batch by batch with the same chunking."""
    from argparse import Namespace
    from switch_nerf_amd import rendering
    R, S, chunk, pb = 1000, 64, 4096, 384
    sd = synth.make_weights(191, synth.BUILDING, gate_scale=0.02)
    rays, _, _ = synth.make_rays(192, R)
    m = _model(torch.float32, 191, 0.02)
    m.eval()
    h = Namespace(coarse_samples=S, fine_samples=0, model_chunk_size=chunk, perturb=1.0, use_sigma_noise=True, sigma_noise_std=1.0,
                  use_cascade=False, image_pixel_batch_size=pb, appearance_dim=48, moe_return_gates=False)
    res = rendering.render_image_rays(m, None, _dev(rays), 7, h)
    assert res["rgb_coarse"].shape == (R, 3) and res["depth_coarse"].shape == (R,) and not res["rgb_coarse"].is_cuda
    p = O.params_from_numpy(sd)
    img = torch.full((R,), 7, dtype=torch.long)
    with torch.no_grad():
        ref = torch.cat([O.render_rays(p, torch.from_numpy(rays[i:i + pb]), img[i:i + pb], synth.BUILDING, S, chunk)["rgb_coarse"]
                         for i in range(0, R, pb)])
    np.testing.assert_allclose(res["rgb_coarse"].numpy(), ref.numpy(), rtol=0, atol=1e-4)
    assert res["gate_loss_coarse"].numel() == 6 + 6 + 4        # chunks per batch: 6, 6, 3 full + 1 ragged


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_nobatch_inference_runs_on_packed_rows(dtype):
    """The evaluation forward without token dropping (Runner.set_no_batch): the rows of every (segment, expert) group are packed
    contiguously like the reference's no-batch dispatcher (expert_locations_begin) - P rows, not n_seg * E * seg_tokens - and the
    result is bit-identical to the capacity-padded layout (the same kernels on the same rows at other addresses; bf16 runs the
    256-row chain geometry through group_begin)."""
    N, S, chunk = 64, 128, 4096
    m = _model(dtype, 97, 1.0, batch_prioritized=False)
    rays, img, _ = synth.make_rays(98, N)
    c_pack = m.forward_rays(_dev(rays), _dev(img), S, chunk, training=False, no_batch=True)
    raw_pack = c_pack["raw"].clone()
    c_pad = m.forward_rays(_dev(rays), _dev(img), S, chunk, training=True, no_batch=True)
    P, n_seg, E = N * S, N * S // chunk, m.E
    assert "group_begin" in c_pack and c_pack["rows"] == P and c_pad["rows"] == n_seg * E * chunk
    assert int((c_pack["tok2row"] < 0).sum()) == 0 and torch.equal(c_pack["idx"], c_pad["idx"])
    begin = c_pack["group_begin"].cpu().numpy()
    counts = c_pack["counts"].cpu().numpy().reshape(-1)
    assert np.array_equal(begin, np.cumsum(counts) - counts) and counts.sum() == P
    assert torch.equal(raw_pack, c_pad["raw"])


@pytest.mark.parametrize("cf,fine,chunk", [(1.0, 0, 2048), (0.5, 0, 4096), (1.25, 64, 2048)])
def test_fused_tail_step_equals_separate_tail_launches_bf16(cf, fine, chunk, monkeypatch):
    """The default bf16 step runs the dense tail inside the two expert launches (SwitchNeRF._tail_fused: chain_big.hip tags 7 / 8); with
    SWN_FUSED_TAIL=0 the tail is its own pair of 64-row launches (the step of the first half of round 4).  Same weights, same batch:
    identical routing and gate values, the saved y bit-identical, rgb / loss / every gradient equal to the rounding order of layer "1"
    (the fused launch starts its accumulators at the bias) - with half the tokens dropped (capacity factor 0.5: the dropped-token
    tiles), with spare capacity (1.25: ragged last tiles) and through the hierarchical step (coarse + fine contexts); the backward alone
    (SWN_FUSED_TAIL_BWD=0) is bit-identical to the fused forward's own two-launch backward except for the gate gradient's sum order."""
    N, S = 128, 64
    rays, img, rgbs = synth.make_rays(141, N)
    pr = torch.rand(N, S, generator=torch.Generator().manual_seed(7)).cuda()
    outs = {}
    for mode in ("fused", "fwd_only", "separate"):
        if mode == "separate":
            monkeypatch.setenv("SWN_FUSED_TAIL", "0")
        elif mode == "fwd_only":
            monkeypatch.setenv("SWN_FUSED_TAIL_BWD", "0")
        m = _model(torch.bfloat16, 140, 1.0, capacity_factor=cf)
        kw = dict(fine_samples=fine, fine_u=torch.rand(N, fine, generator=torch.Generator().manual_seed(8)).cuda()) if fine else {}
        st = m.train_step(_dev(rgbs), _dev(rays), _dev(img), S, chunk, perturb=1.0, perturb_rand=pr, optimizer_step=False, **kw)
        c = st["ctx"]
        assert c["geom"] == 7 and c["tail_fused"] == (mode != "separate")
        outs[mode] = dict(idx=c["idx"].clone(), gmax=c["gmax"].clone(), y=c["y"].clone(), rgb=st["rgb"].clone(), loss=st["loss"].item(),
                          grad=m.grad.clone(), dropped=int((c["tok2row"] < 0).sum()))
        monkeypatch.delenv("SWN_FUSED_TAIL", raising=False)
        monkeypatch.delenv("SWN_FUSED_TAIL_BWD", raising=False)
    a, b, s_ = outs["fused"], outs["fwd_only"], outs["separate"]
    assert (a["dropped"] > 0) == (cf < 1.25 or a["dropped"] > 0)
    if cf == 0.5:
        assert a["dropped"] >= N * S // 2 - 8 * (N * S // chunk)
    for o in (b, s_):
        assert torch.equal(a["idx"], o["idx"]) and torch.equal(a["gmax"], o["gmax"]) and torch.equal(a["y"], o["y"])
    assert torch.equal(a["rgb"], b["rgb"]) and a["loss"] == b["loss"]          # the same forward launch
    gscale = a["grad"].abs().max().item()
    assert (a["grad"] - b["grad"]).abs().max().item() <= 2e-4 * gscale         # (the gate gradient's eight partial sums)
    d_rgb = (a["rgb"] - s_["rgb"]).abs().max().item()
    d_g = (a["grad"] - s_["grad"]).abs().max().item() / gscale
    print(f"fused tail vs separate launches (cf {cf}, fine {fine}, {a['dropped']} dropped): rgb {d_rgb:.2e}, loss {abs(a['loss'] - s_['loss']):.2e}, gradients {d_g:.2e}")
    assert d_rgb < 3e-3 and abs(a["loss"] - s_["loss"]) < 1e-3 * abs(s_["loss"]) and d_g < 1e-2


def test_fused_tail_inference_on_packed_rows_equals_separate_launches_bf16(monkeypatch):
    """The evaluation forward without token dropping (packed row space: `group_begin`, nothing dropped, no saves) through the fused
    launch - raw is all it writes - against the expert chain + 64-row tail chain (SWN_FUSED_TAIL=0): the same experts, raw to the
    rounding order of layer "1"; and the capacity-limited inference forward (dropped tokens) likewise."""
    N, S, chunk = 96, 128, 4096
    rays, img, _ = synth.make_rays(151, N)
    for no_batch in (True, False):
        outs = []
        for fused in (True, False):
            if not fused:
                monkeypatch.setenv("SWN_FUSED_TAIL", "0")
            m = _model(torch.bfloat16, 150, 1.0, capacity_factor=0.75)
            c = m.forward_rays(_dev(rays), _dev(img), S, chunk, training=False, no_batch=no_batch)
            assert c["tail_fused"] == fused and c["geom"] == 7 and ("group_begin" in c) == no_batch
            outs.append((c["idx"].clone(), c["raw"].clone(), c["rgb"].clone(), int((c["tok2row"] < 0).sum())))
            monkeypatch.delenv("SWN_FUSED_TAIL", raising=False)
        assert torch.equal(outs[0][0], outs[1][0]) and outs[0][3] == outs[1][3] and (outs[0][3] == 0) == no_batch
        d_raw = (outs[0][1] - outs[1][1]).abs().max().item()
        d_rgb = (outs[0][2] - outs[1][2]).abs().max().item()
        print(f"fused inference (no_batch={no_batch}, {outs[0][3]} dropped): raw {d_raw:.2e}, rgb {d_rgb:.2e}")
        assert d_raw < 5e-3 and d_rgb < 2e-3
