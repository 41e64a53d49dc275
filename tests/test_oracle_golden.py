"""Pin the CPU oracle (oracle/switchnerf_oracle.py) against golden vectors produced by the imported
reference (oracle/gen_golden.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

import synth
from oracle import switchnerf_oracle as O

G = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    return np.load(os.path.join(G, name + ".npz"))


@pytest.mark.parametrize("bpr", [0, 1])
@pytest.mark.parametrize("cf", [1.0, 1.25, 0.5])
def test_routing_tie_free_bit_exact(bpr, cf):
    g = load(f"route_p2048_e8_bpr{bpr}_cf{cf}")
    gates = synth.make_gates(int(g["seed"]), int(g["P"]), int(g["E"]), float(g["logit_scale"]))
    r = O.route_top1(gates, cf, bool(bpr))
    assert np.array_equal(r["idx"], g["idx"])
    assert np.array_equal(r["loc"], g["loc"])
    assert np.array_equal(r["gate"], g["gate"])          # bit-exact fp32
    assert r["capacity"] == int(g["capacity"])
    l_aux = O.load_balance_loss(torch.from_numpy(gates), torch.from_numpy(r["idx"]))
    assert np.array_equal(l_aux.numpy(), g["l_aux"])


def test_routing_ragged_e16():
    g = load("route_p1000_e16_bpr1_cf1.0")
    gates = synth.make_gates(int(g["seed"]), 1000, 16, 2.0)
    r = O.route_top1(gates, 1.0, True)
    assert np.array_equal(r["idx"], g["idx"]) and np.array_equal(r["loc"], g["loc"])
    assert r["capacity"] == int(g["capacity"]) == 63


def test_routing_with_ties_modulo_tie_groups():
    """The reference's answer is implementation-defined inside groups of exactly equal max-gate (unstable
    argsort) and between experts with exactly equal probability (topk).  Outside those, we must agree."""
    g = load("route_ties_p16384_e8")
    gates = synth.make_gates(int(g["seed"]), 16384, 8, 1.0, quantize_bits=3)
    r = O.route_top1(gates, 1.0, True)
    srt = np.sort(gates, axis=1)
    expert_tie = srt[:, -1] == srt[:, -2]
    assert expert_tie.sum() > 0, "fixture is supposed to contain expert ties"
    ok = ~expert_tie
    assert np.array_equal(r["idx"][ok], g["idx"][ok])
    # the reference's pick on an expert tie must still be one of the maxima
    rows = np.nonzero(expert_tie)[0]
    assert np.all(gates[rows, g["idx"][rows]] == srt[rows, -1])
    # rank: for tokens whose (expert, gate) class is the same in both answers, loc must fall in the same
    # [first, last] interval of its tie group, and match exactly when the group has one member.
    same = r["idx"] == g["idx"]
    # build tie-group intervals from the oracle's own assignment restricted to agreeing tokens
    key = np.stack([r["idx"].astype(np.int64), gates.max(1).view(np.int32).astype(np.int64)], 1)
    _, inv, cnt = np.unique(key, axis=0, return_inverse=True, return_counts=True)
    single = (cnt[inv] == 1) & same
    # a singleton group's rank only depends on how many same-expert tokens have larger gate; that count can
    # differ between the two answers only through expert-tie tokens that were assigned differently.
    n_diff = int((~same).sum())
    assert np.all(np.abs(r["loc"][single].astype(np.int64) - g["loc"][single]) <= n_diff)
    if n_diff == 0:
        lo = np.zeros(len(cnt), np.int64) + 10 ** 9
        hi = np.zeros(len(cnt), np.int64) - 1
        np.minimum.at(lo, inv, r["loc"])
        np.maximum.at(hi, inv, r["loc"])
        assert np.all((g["loc"] >= lo[inv]) & (g["loc"] <= hi[inv]))
        # both are permutations inside each group
        for arr in (r["loc"], g["loc"]):
            assert len(np.unique(np.stack([r["idx"], arr], 1), axis=0)) == 16384


def test_positional_encoding():
    g = load("pe")
    x = torch.from_numpy(g["x"])
    assert np.array_equal(O.positional_encoding(x, 12).numpy(), g["pe12"])
    assert np.array_equal(O.positional_encoding(x, 4).numpy(), g["pe4"])


@pytest.mark.parametrize("tag,cfg", [("m64e4", synth.small_cfg(64, 4)), ("m256e8", synth.BUILDING)])
def test_moe_layer_fwd_bwd(tag, cfg):
    g = load(f"moe_layer_{tag}")
    seed, P = int(g["seed"]), int(g["P"])
    p = O.params_from_numpy(synth.make_weights(seed, cfg), requires_grad=True)
    rng = np.random.default_rng(seed + 1000)
    x = torch.from_numpy(rng.standard_normal((P, cfg["model_dim"])).astype(np.float32)).requires_grad_(True)
    gi = torch.from_numpy(rng.standard_normal((P, cfg["gate_hidden"])).astype(np.float32)).requires_grad_(True)
    L = cfg["expert_layers"]
    W = [p[f"layers.0.experts.0.weights.{l}"] for l in range(L)]
    B = [p[f"layers.0.experts.0.bias.{l}"] for l in range(L)]
    y, l_aux, routing, gates = O.moe_layer(x, gi, p["layers.0.gates.0.wg.weight"], W, B, cfg["skips"], 1.0, True)
    assert np.array_equal(routing["idx"], g["topk"].reshape(-1))
    np.testing.assert_allclose(y.detach().numpy(), g["y"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(l_aux.detach().numpy(), g["l_aux"], rtol=1e-6)
    dy = torch.from_numpy(rng.standard_normal(y.shape).astype(np.float32))
    (y * dy).sum().backward(retain_graph=True)
    np.testing.assert_allclose(x.grad.numpy(), g["dx"], rtol=0, atol=5e-6)
    np.testing.assert_allclose(gi.grad.numpy(), g["dgate_input"], rtol=0, atol=5e-6)
    names = {"gates.0.wg.weight": "layers.0.gates.0.wg.weight"}
    for l in range(L):
        names[f"experts.0.weights.{l}"] = f"layers.0.experts.0.weights.{l}"
        names[f"experts.0.bias.{l}"] = f"layers.0.experts.0.bias.{l}"
    for short, full in names.items():
        got = p[full].grad.numpy()
        ref = g["grad__" + short]
        if got.size <= 4096:
            np.testing.assert_allclose(got, ref, rtol=0, atol=2e-5)
        else:
            np.testing.assert_allclose(synth.checksum(got), ref, rtol=2e-5)
            sl = got.reshape(-1)[:: max(1, got.size // 997)][:997]
            np.testing.assert_allclose(sl, g["gslice__" + short], rtol=0, atol=2e-5)
    gi.grad = None
    gl = torch.autograd.grad(l_aux, [gi, p["layers.0.gates.0.wg.weight"]])
    np.testing.assert_allclose(gl[0].numpy(), g["laux_dgate_input"], rtol=0, atol=1e-8)
    np.testing.assert_allclose(gl[1].numpy(), g["laux_dwg"], rtol=0, atol=1e-7)


def test_moe_layer_gate_noise_vs_reference():
    """The gate-noise branch of a training forward (--gate_noise > 0: opts.py:208, tutel_moe_layer_nobatch.py:119-122): the fixture
    carries the reference layer's own noise draw (replayed from its seed by the generator, which asserts that the replay reproduces
    the layer's routing); with it the oracle reproduces expert choice, output, l_aux and the gradients of the reference run."""
    g = load("moe_layer_noise_m256e8")
    cfg = synth.BUILDING
    seed, P, gn = int(g["seed"]), int(g["P"]), float(g["gate_noise"])
    p = O.params_from_numpy(synth.make_weights(seed, cfg), requires_grad=True)
    rng = np.random.default_rng(seed + 1000)
    x = torch.from_numpy(rng.standard_normal((P, cfg["model_dim"])).astype(np.float32)).requires_grad_(True)
    gi = torch.from_numpy(rng.standard_normal((P, cfg["gate_hidden"])).astype(np.float32)).requires_grad_(True)
    L = cfg["expert_layers"]
    W = [p[f"layers.0.experts.0.weights.{l}"] for l in range(L)]
    B = [p[f"layers.0.experts.0.bias.{l}"] for l in range(L)]
    wg = p["layers.0.gates.0.wg.weight"]
    y, l_aux, routing, _ = O.moe_layer(x, gi, wg, W, B, cfg["skips"], 1.0, True, gate_noise=gn, noise=torch.from_numpy(g["noise"]))
    assert np.array_equal(routing["idx"], g["topk"].reshape(-1))
    y0, _, r0, _ = O.moe_layer(x, gi, wg, W, B, cfg["skips"], 1.0, True)
    assert not np.array_equal(r0["idx"], routing["idx"])                 # (the noise does move expert choices in this fixture)
    np.testing.assert_allclose(y.detach().numpy(), g["y"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(l_aux.detach().numpy(), g["l_aux"], rtol=1e-6)
    dy = torch.from_numpy(rng.standard_normal(y.shape).astype(np.float32))
    (y * dy).sum().backward(retain_graph=True)
    np.testing.assert_allclose(x.grad.numpy(), g["dx"], rtol=0, atol=5e-6)
    np.testing.assert_allclose(gi.grad.numpy(), g["dgate_input"], rtol=0, atol=5e-6)
    np.testing.assert_allclose(wg.grad.numpy(), g["dwg"], rtol=0, atol=2e-5)
    gi.grad = None
    gl = torch.autograd.grad(l_aux, [gi, wg])
    np.testing.assert_allclose(gl[0].numpy(), g["laux_dgate_input"], rtol=0, atol=1e-8)
    np.testing.assert_allclose(gl[1].numpy(), g["laux_dwg"], rtol=0, atol=1e-7)


def test_moe_layer_normal_noise_vs_reference():
    """use_normal_noise (tutel_moe_layer_nobatch.py:116-117) together with gate noise: the fixture carries both draws of the reference
    layer's own run (normal noise first, gate noise second); the oracle reproduces expert choice, output, l_aux and the gradients."""
    g = load("moe_layer_normal_noise_m256e8")
    cfg = synth.BUILDING
    seed, P, gn = int(g["seed"]), int(g["P"]), float(g["gate_noise"])
    p = O.params_from_numpy(synth.make_weights(seed, cfg), requires_grad=True)
    rng = np.random.default_rng(seed + 1000)
    x = torch.from_numpy(rng.standard_normal((P, cfg["model_dim"])).astype(np.float32)).requires_grad_(True)
    gi = torch.from_numpy(rng.standard_normal((P, cfg["gate_hidden"])).astype(np.float32)).requires_grad_(True)
    L = cfg["expert_layers"]
    W = [p[f"layers.0.experts.0.weights.{l}"] for l in range(L)]
    B = [p[f"layers.0.experts.0.bias.{l}"] for l in range(L)]
    wg = p["layers.0.gates.0.wg.weight"]
    y, l_aux, routing, _ = O.moe_layer(x, gi, wg, W, B, cfg["skips"], 1.0, True, gate_noise=gn, noise=torch.from_numpy(g["noise"]),
                                       normal_noise=torch.from_numpy(g["normal_noise"]))
    assert np.array_equal(routing["idx"], g["topk"].reshape(-1))
    _, _, r0, _ = O.moe_layer(x, gi, wg, W, B, cfg["skips"], 1.0, True, gate_noise=gn, noise=torch.from_numpy(g["noise"]))
    assert not np.array_equal(r0["idx"], routing["idx"])                 # (the normal noise moves expert choices in this fixture)
    np.testing.assert_allclose(y.detach().numpy(), g["y"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(l_aux.detach().numpy(), g["l_aux"], rtol=1e-6)
    dy = torch.from_numpy(rng.standard_normal(y.shape).astype(np.float32))
    (y * dy).sum().backward(retain_graph=True)
    np.testing.assert_allclose(x.grad.numpy(), g["dx"], rtol=0, atol=5e-6)
    np.testing.assert_allclose(gi.grad.numpy(), g["dgate_input"], rtol=0, atol=5e-6)
    np.testing.assert_allclose(wg.grad.numpy(), g["dwg"], rtol=0, atol=2e-5)


@pytest.mark.parametrize("tag,cfg", [("m256e8_bpr", synth.BUILDING), ("m64e4_plain", synth.small_cfg(64, 4))])
def test_moe_layer_top2_vs_reference(tag, cfg):
    """A top-2 gate (extract_critical with top_k = 2, tutel_fast_dispatch.py:176-217): the reference layer's own run with `top_k = 2` -
    both choices' experts and locations exact (acc_base, batch-prioritised and plain), normalised gates, output, l_aux, every gradient."""
    g = load(f"moe_layer_top2_{tag}")
    seed, P, bpr, cf = int(g["seed"]), int(g["P"]), bool(g["bpr"]), float(g["cf"])
    p = O.params_from_numpy(synth.make_weights(seed, cfg), requires_grad=True)
    rng = np.random.default_rng(seed + 1000)
    x = torch.from_numpy(rng.standard_normal((P, cfg["model_dim"])).astype(np.float32)).requires_grad_(True)
    gi = torch.from_numpy(rng.standard_normal((P, cfg["gate_hidden"])).astype(np.float32)).requires_grad_(True)
    L = cfg["expert_layers"]
    W = [p[f"layers.0.experts.0.weights.{l}"] for l in range(L)]
    B = [p[f"layers.0.experts.0.bias.{l}"] for l in range(L)]
    wg = p["layers.0.gates.0.wg.weight"]
    y, l_aux, r, gates = O.moe_layer_topk(x, gi, wg, W, B, cfg["skips"], 2, cf, bpr)
    assert np.array_equal(r["idx"].T, g["topk"])
    assert np.array_equal(r["loc"], g["loc"]) and int(r["capacity"]) == int(g["capacity"])
    assert (r["loc"] >= r["capacity"]).any() and (r["loc"][1] < r["capacity"]).any()
    np.testing.assert_allclose(y.detach().numpy(), g["y"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(l_aux.detach().numpy(), g["l_aux"], rtol=1e-6)
    dy = torch.from_numpy(rng.standard_normal(y.shape).astype(np.float32))
    (y * dy).sum().backward(retain_graph=True)
    np.testing.assert_allclose(x.grad.numpy(), g["dx"], rtol=0, atol=5e-6)
    np.testing.assert_allclose(gi.grad.numpy(), g["dgate_input"], rtol=0, atol=5e-6)
    names = {"gates.0.wg.weight": "layers.0.gates.0.wg.weight"}
    for l in range(L):
        names[f"experts.0.weights.{l}"] = f"layers.0.experts.0.weights.{l}"
        names[f"experts.0.bias.{l}"] = f"layers.0.experts.0.bias.{l}"
    for short, full in names.items():
        got = p[full].grad.numpy()
        ref = g["grad__" + short]
        if got.size <= 4096:
            np.testing.assert_allclose(got, ref, rtol=0, atol=2e-5)
        else:
            np.testing.assert_allclose(synth.checksum(got), ref, rtol=2e-5)
            sl = got.reshape(-1)[:: max(1, got.size // 997)][:997]
            np.testing.assert_allclose(sl, g["gslice__" + short], rtol=0, atol=2e-5)
    gi.grad = None
    gl = torch.autograd.grad(l_aux, [gi, wg])
    np.testing.assert_allclose(gl[0].numpy(), g["laux_dgate_input"], rtol=0, atol=1e-8)
    np.testing.assert_allclose(gl[1].numpy(), g["laux_dwg"], rtol=0, atol=1e-7)


@pytest.mark.parametrize("tag", ["k1", "k2"])
def test_moe_layer_load_importance_vs_reference(tag):
    """--use_load_importance_loss with gate noise, training mode (extract_critical_load_importance, tutel_fast_dispatch.py:219-265): the
    reference layer's own run (top-1 and top-2 gate) with its replayed noise draw - experts, output, the load / importance loss, the
    load-balance loss of the extras, and the gradients of both losses and of the output."""
    g = load(f"moe_layer_load_importance_{tag}")
    cfg = synth.BUILDING
    seed, P, gn, K = int(g["seed"]), int(g["P"]), float(g["gate_noise"]), int(g["top_k"])
    p = O.params_from_numpy(synth.make_weights(seed, cfg), requires_grad=True)
    rng = np.random.default_rng(seed + 1000)
    x = torch.from_numpy(rng.standard_normal((P, cfg["model_dim"])).astype(np.float32)).requires_grad_(True)
    gi = torch.from_numpy(rng.standard_normal((P, cfg["gate_hidden"])).astype(np.float32)).requires_grad_(True)
    L = cfg["expert_layers"]
    W = [p[f"layers.0.experts.0.weights.{l}"] for l in range(L)]
    B = [p[f"layers.0.experts.0.bias.{l}"] for l in range(L)]
    wg = p["layers.0.gates.0.wg.weight"]
    y, l_aux, r, _ = O.moe_layer_topk(x, gi, wg, W, B, cfg["skips"], K, 1.0, True, gate_noise=gn, noise=torch.from_numpy(g["noise"]),
                                      load_importance=True)
    assert np.array_equal(r["idx"].T, g["topk"])
    np.testing.assert_allclose(y.detach().numpy(), g["y"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(l_aux.detach().numpy(), g["l_aux"], rtol=2e-6)
    np.testing.assert_allclose(r["balance_loss"].detach().numpy(), g["balance_loss"], rtol=1e-6)
    dy = torch.from_numpy(rng.standard_normal(y.shape).astype(np.float32))
    (y * dy).sum().backward(retain_graph=True)
    np.testing.assert_allclose(x.grad.numpy(), g["dx"], rtol=0, atol=5e-6)
    np.testing.assert_allclose(gi.grad.numpy(), g["dgate_input"], rtol=0, atol=5e-6)
    np.testing.assert_allclose(wg.grad.numpy(), g["dwg"], rtol=0, atol=2e-5)
    gl = torch.autograd.grad(l_aux, [gi, wg], retain_graph=True)
    np.testing.assert_allclose(gl[0].numpy(), g["laux_dgate_input"], rtol=1e-4, atol=1e-6 * np.abs(g["laux_dgate_input"]).max())
    np.testing.assert_allclose(gl[1].numpy(), g["laux_dwg"], rtol=1e-4, atol=1e-6 * np.abs(g["laux_dwg"]).max())
    gb = torch.autograd.grad(r["balance_loss"], [gi, wg])
    np.testing.assert_allclose(gb[0].numpy(), g["bal_dgate_input"], rtol=0, atol=1e-8)
    np.testing.assert_allclose(gb[1].numpy(), g["bal_dwg"], rtol=0, atol=1e-7)


@pytest.mark.parametrize("tag", ["unbalanced", "balanced"])
def test_model_forward(tag):
    g = load(f"model_fwd_{tag}")
    cfg = synth.BUILDING
    p = O.params_from_numpy(synth.make_weights(int(g["seed"]), cfg, gate_scale=float(g["gate_scale"])))
    with torch.no_grad():
        r = O.nerf_moe_forward(p, torch.from_numpy(g["x"]), cfg, 1.0, True, torch.from_numpy(g["sigma_noise"]))
    assert np.array_equal(r["routing"]["idx"], g["moe_gates"].reshape(-1))
    np.testing.assert_allclose(r["outputs"].numpy(), g["outputs"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(r["moe_loss"].numpy(), g["moe_loss"], rtol=1e-6)
    kept = (r["routing"]["loc"] < r["routing"]["capacity"]).mean()
    print(f"{tag}: kept fraction {kept:.3f}, counts {r['routing']['counts']}")


@pytest.mark.parametrize("tag", ["unbalanced", "balanced", "cf125_nobpr", "cf125_bpr", "cf050_bpr"])
def test_render_and_training_step(tag):
    g = load(f"render_train_{tag}")
    cfg = synth.BUILDING
    p = O.params_from_numpy(synth.make_weights(int(g["seed"]), cfg, gate_scale=float(g["gate_scale"])), requires_grad=True)
    N, S, chunk = int(g["N"]), int(g["S"]), int(g["chunk"])
    rays, img, rgbs = synth.make_rays(52, N)
    kw = {}
    if "capacity_factor" in g:
        kw = dict(capacity_factor=float(g["capacity_factor"]), batch_prioritized=bool(int(g["bpr"])))
    st = O.training_step(p, torch.from_numpy(rays), torch.from_numpy(img), torch.from_numpy(rgbs), cfg, S, chunk, **kw)
    res = st["results"]
    got_idx = np.concatenate([r["idx"] for r in res["routings"]]).reshape(N, S)
    assert np.array_equal(got_idx, g["moe_gates"])
    np.testing.assert_allclose(res["rgb_coarse"].detach().numpy(), g["rgb"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(res["sigma_coarse"].detach().numpy(), g["sigma"], rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(res["depth_coarse"].numpy(), g["depth"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(res["depth_variance_coarse"].numpy(), g["depth_variance"], rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(res["gate_loss_coarse"].detach().numpy(), g["gate_loss"], rtol=1e-6)
    np.testing.assert_allclose(st["loss"].detach().numpy(), g["loss"], rtol=1e-6)
    st["loss"].backward()
    for k, t in p.items():
        ref_sum = g["gsum__" + k]
        got = t.grad.numpy()
        scale = max(1e-12, float(ref_sum[1]))
        assert abs(synth.checksum(got)[0] - ref_sum[0]) <= 2e-4 * scale + 1e-9, k
        assert abs(synth.checksum(got)[1] - ref_sum[1]) <= 2e-4 * scale + 1e-9, k
        sl = got.reshape(-1)[:: max(1, got.size // 499)][:499]
        np.testing.assert_allclose(sl, g["gslice__" + k], rtol=1e-3, atol=1e-7 + 1e-4 * np.abs(g["gslice__" + k]).max(), err_msg=k)


@pytest.mark.parametrize("tag", ["unbalanced", "balanced"])
def test_training_step_bf16_autocast_vs_reference_cpu_autocast(tag):
    """oracle.Autocast (which operator of the path rounds to bf16, which runs in an fp32 island) against the REFERENCE run under
    bf16 autocast - on the CPU backend's autocast, the one this container can execute (gen_golden.gen_render_autocast), sigma noise
    on.  Same rounding points, same operator order => the forward agrees to the bit: top-1 indices equal, rgb and sigma equal
    (asserted <= 1e-6), loss 1e-6 relative."""
    g = load(f"render_train_bf16cpu_{tag}")
    cfg = synth.BUILDING
    p = O.params_from_numpy(synth.make_weights(int(g["seed"]), cfg, gate_scale=float(g["gate_scale"])), requires_grad=True)
    N, S, chunk = int(g["N"]), int(g["S"]), int(g["chunk"])
    rays, img, rgbs = synth.make_rays(52, N)
    ac = O.Autocast(torch.bfloat16, policy="cpu")
    st = O.training_step(p, torch.from_numpy(rays), torch.from_numpy(img), torch.from_numpy(rgbs), cfg, S, chunk,
                         sigma_noise=torch.from_numpy(g["sigma_noise"]), autocast=ac)
    res = st["results"]
    assert int(g["sigma_is_bf16"]) == 1      # the CPU backend leaves softplus in the 16-bit type (oracle.Autocast.FP32_OPS["cpu"])
    got_idx = np.concatenate([r["idx"] for r in res["routings"]]).reshape(N, S)
    gaps = np.concatenate([r["top2_gap"] for r in res["routings"]]).reshape(N, S)
    mis = got_idx != g["moe_gates"]
    print(f"{tag}: {int(mis.sum())} of {mis.size} top-1 indices differ from the reference's bf16 run; loss {float(st['loss']):.8f} vs {float(g['loss']):.8f}")
    assert int(mis.sum()) == 0, "same rounding points, same operator order: the forward is bit-identical (observed)"
    d_rgb = np.abs(res["rgb_coarse"].detach().numpy() - g["rgb"]).max()
    sig, sig_ref = res["sigma_coarse"].detach().float().numpy(), g["sigma"]
    ok = ~mis                                  # (a flipped token went through another expert)
    d_sig = (np.abs(sig - sig_ref)[ok] / np.maximum(np.abs(sig_ref)[ok], 1e-3)).max()
    print(f"{tag}: max |rgb diff| {d_rgb:.2e}, max relative sigma diff {d_sig:.2e}")
    assert d_rgb <= 1e-6 and d_sig <= 1e-6          # (observed: 0.0 and 0.0)
    np.testing.assert_allclose(res["gate_loss_coarse"].detach().numpy(), g["gate_loss"], rtol=1e-4)
    np.testing.assert_allclose(st["loss"].detach().numpy(), g["loss"], rtol=1e-6)
    # gradients: the reference's autocast backward forms dW = x^T dZ and the bias sums with bf16 OUTPUTS (observed: single entries
    # off by up to 20 % of the tensor's largest entry for a bias row summed in bf16), the oracle's autograd keeps them in fp32:
    # the tensors' absolute sums agree to 1 %, entries to a quarter of the largest entry
    st["loss"].backward()
    for k, t in p.items():
        ref_sum = g["gsum__" + k]
        got = t.grad.numpy()
        scale = max(1e-12, float(ref_sum[1]))
        assert abs(synth.checksum(got)[1] - ref_sum[1]) <= 1e-2 * scale + 1e-9, k
        sl = got.reshape(-1)[:: max(1, got.size // 499)][:499]
        ref = g["gslice__" + k]
        assert np.abs(sl - ref).max() <= 0.25 * np.abs(ref).max() + 1e-9, k


def test_autocast_operator_table_matches_torch_cpu():
    """oracle.Autocast.FP32_OPS["cpu"] / the lower-precision list against torch's own CPU autocast: output dtypes of the operators on
    the path for bf16 inputs.  (The CUDA table is checked the same way on the GPU box: tests/test_fullsize_gpu.py.)"""
    import torch.nn.functional as F
    x = torch.randn(4, 8).bfloat16()
    w, b = torch.randn(8), torch.randn(8)
    with torch.autocast("cpu", dtype=torch.bfloat16):
        got = dict(layer_norm=F.layer_norm(x, (8,), w, b).dtype, softplus=F.softplus(x - 1, 1, 20).dtype, sigmoid=torch.sigmoid(x).dtype,
                   relu=torch.relu(x).dtype, linear=F.linear(torch.randn(4, 8), torch.randn(3, 8), torch.randn(3)).dtype,
                   baddbmm=torch.baddbmm(torch.randn(2, 1, 3), torch.randn(2, 4, 8), torch.randn(2, 8, 3)).dtype,
                   cat=torch.cat([x, torch.randn(4, 2)], 1).dtype, softmax=torch.softmax(x.float(), 1).dtype,
                   mse_loss=F.mse_loss(x, torch.randn(4, 8)).dtype)
        y = x.clone()
        y += torch.randn(4, 8)
        got["iadd"] = y.dtype
    ac = O.Autocast(torch.bfloat16, policy="cpu")
    lo, f32 = torch.bfloat16, torch.float32
    want = dict(layer_norm=f32 if ac.fp32_op("layer_norm") else lo, softplus=f32 if ac.fp32_op("softplus") else lo, sigmoid=lo, relu=lo,
                linear=lo, baddbmm=lo, cat=f32, softmax=f32, mse_loss=f32, iadd=lo)
    assert got == want, (got, want)
    # the emulated matrix product (fp32 product of rounded operands, one rounding) equals torch's own bf16 linear to one ulp
    xi, wi, bi = torch.randn(64, 256), torch.randn(32, 256) / 16, torch.randn(32)
    with torch.autocast("cpu", dtype=torch.bfloat16):
        ref = F.linear(xi, wi, bi)
    em = ac.linear(xi, wi, bi)
    assert ref.dtype == em.dtype == torch.bfloat16
    assert ((ref.float() - em.float()).abs() <= 2.0 ** -7 * ref.float().abs() + 1e-6).all()
    assert (ref != em).float().mean().item() < 0.02


@pytest.mark.parametrize("tag", ["det", "perturbed"])
def test_hierarchical_training_step(tag):
    """Coarse pass -> _sample_pdf -> fine pass -> sort-merge -> compositing, forward and every parameter gradient, against
    the reference run with fine_samples = 96 (oracle/gen_golden.py gen_render_fine)."""
    g = load(f"render_train_fine_{tag}")
    cfg = synth.BUILDING
    p = O.params_from_numpy(synth.make_weights(int(g["seed"]), cfg, gate_scale=float(g["gate_scale"])), requires_grad=True)
    N, S, Fn, chunk = int(g["N"]), int(g["S"]), int(g["F"]), int(g["chunk"])
    rays, img, rgbs = synth.make_rays(62, N)
    kw = {}
    if float(g["perturb"]) > 0:
        kw = dict(perturb=float(g["perturb"]), perturb_rand=torch.from_numpy(g["perturb_rand"]), fine_u=torch.from_numpy(g["fine_u"]))
    st = O.training_step(p, torch.from_numpy(rays), torch.from_numpy(img), torch.from_numpy(rgbs), cfg, S, chunk,
                         fine_samples=Fn, **kw)
    res = st["results"]
    assert np.array_equal(np.concatenate([r["idx"] for r in res["routings"]]).reshape(N, S), g["moe_gates_coarse"])
    assert np.array_equal(np.concatenate([r["idx"] for r in res["routings_fine"]]).reshape(N, Fn), g["moe_gates_fine"])
    np.testing.assert_allclose(res["sigma_coarse"].detach().numpy(), g["sigma_coarse"], rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(res["sigma_fine"].detach().numpy(), g["sigma_fine"], rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(res["rgb_fine"].detach().numpy(), g["rgb"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(res["depth_fine"].numpy(), g["depth"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(res["depth_variance_fine"].numpy(), g["depth_variance"], rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(res["gate_loss_coarse"].detach().numpy(), g["gate_loss_coarse"], rtol=1e-6)
    np.testing.assert_allclose(res["gate_loss_fine"].detach().numpy(), g["gate_loss_fine"], rtol=1e-6)
    np.testing.assert_allclose(st["loss"].detach().numpy(), g["loss"], rtol=1e-6)
    st["loss"].backward()
    for k, t in p.items():
        ref_sum = g["gsum__" + k]
        got = t.grad.numpy()
        scale = max(1e-12, float(ref_sum[1]))
        assert abs(synth.checksum(got)[0] - ref_sum[0]) <= 2e-4 * scale + 1e-9, k
        assert abs(synth.checksum(got)[1] - ref_sum[1]) <= 2e-4 * scale + 1e-9, k
        sl = got.reshape(-1)[:: max(1, got.size // 499)][:499]
        np.testing.assert_allclose(sl, g["gslice__" + k], rtol=1e-3, atol=1e-7 + 1e-4 * np.abs(g["gslice__" + k]).max(), err_msg=k)


def test_composite_and_sample_pdf():
    g = load("composite")
    z = torch.from_numpy(g["z"])
    c = O.composite(torch.from_numpy(g["rgbs"]), torch.from_numpy(g["sigmas"]), z)
    np.testing.assert_allclose(c["rgb"].numpy(), g["rgb"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(c["weights"].numpy(), g["weights"], rtol=0, atol=1e-7)
    np.testing.assert_allclose(c["depth"].numpy(), g["depth"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(c["depth_variance"].numpy(), g["depth_variance"], rtol=1e-5, atol=1e-9)
    zmid = 0.5 * (z[:, :-1] + z[:, 1:])
    fine = O.sample_pdf(zmid, c["weights"][:, 1:-1], 64)
    np.testing.assert_allclose(fine.numpy(), g["fine_det"], rtol=0, atol=1e-6)
    rays = torch.from_numpy(g["rays"])
    z2 = O.sample_z(rays[:, 6:7], rays[:, 7:8], 256)
    assert np.array_equal(z2.numpy(), g["z"])


def test_model_forward_nobatch_eval_path():
    """Reference eval path (seqexperts + set_no_batch): no token is dropped."""
    g = load("model_fwd_nobatch")
    cfg = synth.BUILDING
    p = O.params_from_numpy(synth.make_weights(int(g["seed"]), cfg, gate_scale=float(g["gate_scale"])))
    with torch.no_grad():
        r = O.nerf_moe_forward(p, torch.from_numpy(g["x"]), cfg, 1.0, False, None, no_batch=True)
    assert np.array_equal(r["routing"]["idx"], g["moe_gates"].reshape(-1))
    np.testing.assert_allclose(r["outputs"].numpy(), g["outputs"], rtol=0, atol=2e-6)


def test_mip_cast_embed_resample():
    """mip_cast_rays, MipEmbedder and the deterministic level resampling against the reference's own outputs."""
    g = load("mip_kernels")
    rays, radii, z = torch.from_numpy(g["rays"]), torch.from_numpy(g["radii"]), torch.from_numpy(g["z"])
    mean, cov = O.mip_cast_rays(rays[:, 0:3], rays[:, 3:6], radii, z)
    np.testing.assert_allclose(mean.numpy(), g["mean"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(cov.numpy(), g["cov"], rtol=1e-5, atol=1e-12)
    ipe = O.mip_embed(mean, cov, 12).reshape(-1, 75)
    np.testing.assert_allclose(ipe.numpy(), g["ipe"], rtol=0, atol=2e-6)
    zs = O.mip_resample(z, torch.from_numpy(g["weights"]), 40, 0.01)
    np.testing.assert_allclose(zs.numpy(), g["z_resampled_det"], rtol=0, atol=1e-6)


@pytest.mark.parametrize("tag", ["det", "perturbed"])
def test_mip_training_step(tag):
    """Two-level mip render + loss + every parameter gradient against the reference's MipNeRFMoE run."""
    g = load(f"mip_train_{tag}")
    cfg = synth.BUILDING
    p = O.params_from_numpy(synth.make_weights(int(g["seed"]), cfg, gate_scale=float(g["gate_scale"])), requires_grad=True)
    N, S, Fn, chunk = int(g["N"]), int(g["S"]), int(g["F"]), int(g["chunk"])
    rays, img, rgbs = synth.make_rays(72, N)
    kw = {}
    if float(g["perturb"]) > 0:
        kw = dict(perturb=float(g["perturb"]), perturb_rand=torch.from_numpy(g["perturb_rand"]), fine_u=torch.from_numpy(g["fine_u"]))
    st = O.training_step_mip(p, torch.from_numpy(rays), torch.from_numpy(g["radii"]), torch.from_numpy(img), torch.from_numpy(rgbs),
                             cfg, S, Fn, chunk, **kw)
    res = st["results"]
    assert np.array_equal(np.concatenate([r["idx"] for r in res["routings"]]).reshape(N, S - 1), g["moe_gates_coarse"])
    assert np.array_equal(np.concatenate([r["idx"] for r in res["routings_fine"]]).reshape(N, Fn - 1), g["moe_gates_fine"])
    np.testing.assert_allclose(res["rgb_coarse"].detach().numpy(), g["rgb_coarse"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(res["rgb_fine"].detach().numpy(), g["rgb_fine"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(res["depth_variance_fine"].numpy(), g["depth_variance"], rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(res["gate_loss_fine"].detach().numpy(), g["gate_loss_fine"], rtol=1e-6)
    np.testing.assert_allclose(st["loss"].detach().numpy(), g["loss"], rtol=1e-6)
    st["loss"].backward()
    for k, t in p.items():
        ref_sum = g["gsum__" + k]
        got = t.grad.numpy()
        scale = max(1e-12, float(ref_sum[1]))
        assert abs(synth.checksum(got)[0] - ref_sum[0]) <= 2e-4 * scale + 1e-9, k
        assert abs(synth.checksum(got)[1] - ref_sum[1]) <= 2e-4 * scale + 1e-9, k
        sl = got.reshape(-1)[:: max(1, got.size // 499)][:499]
        np.testing.assert_allclose(sl, g["gslice__" + k], rtol=1e-3, atol=1e-7 + 1e-4 * np.abs(g["gslice__" + k]).max(), err_msg=k)


def test_checkpoint_layouts_match_reference_conversion():
    """switch_nerf_amd.checkpoint.to_seqexperts reproduces the reference's convert_to_seqexperts key for key (names, shapes,
    values), and to_expertmlp inverts it (and strips the DDP prefix)."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from switch_nerf_amd import checkpoint as ck
    g = load("checkpoint_seqexperts")
    sd = {("module." + k): torch.from_numpy(v.copy()) for k, v in synth.make_weights(91, synth.BUILDING).items()}
    conv = ck.to_seqexperts(sd, prefix="module.")
    assert sorted(conv.keys()) == list(g["keys"])
    for k, s_ref, shp in zip(g["keys"], g["sums"], g["shapes"]):
        assert str(tuple(conv[k].shape)) == shp, k
        np.testing.assert_allclose(synth.checksum(conv[k].numpy()), s_ref, rtol=1e-6, err_msg=k)
    back = ck.to_expertmlp(conv)
    ref = synth.make_weights(91, synth.BUILDING)
    assert sorted(back.keys()) == sorted(ref.keys())
    for k, v in ref.items():
        assert torch.equal(back[k], torch.from_numpy(v)), k


def test_dense_nerf_config0():
    """BASELINE configs[0]: the reference's dense NeRF (use_moe off) on 1024 rays x 64 samples - render, loss and every gradient."""
    g = load("dense_nerf_train")
    cfg = synth.DENSE
    p = O.params_from_numpy(synth.make_dense_weights(int(g["seed"]), cfg), requires_grad=True)
    N, S = int(g["N"]), int(g["S"])
    rays, img, rgbs = synth.make_rays(162, N)
    res = O.render_rays_dense(p, torch.from_numpy(rays), torch.from_numpy(img), cfg, S)
    np.testing.assert_allclose(res["rgb_coarse"].detach().numpy(), g["rgb"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(res["depth_coarse"].numpy(), g["depth"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(res["raw"][..., 3].detach().numpy()[:64], g["sigma_head"], rtol=1e-5, atol=2e-6)
    loss = F_mse(res["rgb_coarse"], torch.from_numpy(rgbs))
    np.testing.assert_allclose(loss.detach().numpy(), g["loss"], rtol=1e-6)
    loss.backward()
    for k, t in p.items():
        ref_sum = g["gsum__" + k]
        got = t.grad.numpy()
        scale = max(1e-12, float(ref_sum[1]))
        assert abs(synth.checksum(got)[0] - ref_sum[0]) <= 2e-4 * scale + 1e-9, k
        sl = got.reshape(-1)[:: max(1, got.size // 499)][:499]
        np.testing.assert_allclose(sl, g["gslice__" + k], rtol=1e-3, atol=1e-7 + 1e-4 * np.abs(g["gslice__" + k]).max(), err_msg=k)


def F_mse(a, b):
    return torch.nn.functional.mse_loss(a, b, reduction="mean")


def test_background_points_and_bound():
    """_depth2pts_outside / _intersect_sphere (rendering.py:497-570) on raw tensors."""
    g = load("bg_points")
    rays, _, _ = synth.make_bg_rays(84, 40)
    r = torch.from_numpy(rays)
    c, rad = torch.from_numpy(synth.SPHERE_CENTER), torch.from_numpy(synth.SPHERE_RADIUS)
    pts, dreal = O.depth2pts_outside(r[:, None, :3], r[:, None, 3:6], torch.from_numpy(g["depth"]), c, rad)
    np.testing.assert_allclose(pts.numpy(), g["pts"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(dreal.numpy(), g["depth_real"], rtol=1e-6)
    np.testing.assert_allclose(O.intersect_sphere(r[:, :3], r[:, 3:6], c, rad).numpy(), g["fg_far"], rtol=1e-6)
    with pytest.raises(Exception, match="bounded by the unit sphere"):
        O.intersect_sphere(r[:, :3] + 5.0, r[:, 3:6], c, rad)


@pytest.mark.parametrize("tag", ["coarse_det", "coarse", "fine"])
def test_background_render_train(tag):
    """render_rays with the dense background model and the ellipsoid bound (rendering.py:32-159): blended rgb / depth,
    loss and every gradient of both models against the reference's own run."""
    g = load(f"bg_train_{tag}")
    cfg, cfg_bg = synth.BUILDING, synth.DENSE_BG
    p = O.params_from_numpy(synth.make_weights(int(g["seed"]), cfg, gate_scale=float(g["gate_scale"])), requires_grad=True)
    pb = O.params_from_numpy(synth.make_dense_weights(int(g["seed_bg"]), cfg_bg), requires_grad=True)
    N, S, Fn, chunk = int(g["N"]), int(g["S"]), int(g["F"]), int(g["chunk"])
    rays, img, rgbs = synth.make_bg_rays(83, N)
    kw = {}
    if float(g["perturb"]) > 0:
        kw = dict(perturb=1.0, perturb_rand=torch.from_numpy(g["perturb_rand"]), perturb_rand_bg=torch.from_numpy(g["perturb_rand_bg"]))
        if Fn:
            kw.update(fine_u=torch.from_numpy(g["fine_u"]), fine_u_bg=torch.from_numpy(g["fine_u_bg"]))
    res = O.render_rays_bg(p, pb, torch.from_numpy(rays), torch.from_numpy(img), cfg, cfg_bg, S, chunk,
                           torch.from_numpy(synth.SPHERE_CENTER), torch.from_numpy(synth.SPHERE_RADIUS), fine_samples=Fn, **kw)
    typ = "fine" if Fn else "coarse"
    assert len(res["with_bg"]) == int(g["n_bg"])
    np.testing.assert_allclose(res["fg_far"].numpy(), g["fg_far"], rtol=1e-6)
    np.testing.assert_allclose(res["fg_rgb"].detach().numpy(), g["fg_rgb"], rtol=0, atol=3e-6)
    # (the background's hierarchical pass is evaluated in one chunk here and in 1024-point chunks by the reference:
    #  the CPU GEMMs round differently, hence 1e-5 instead of 3e-6 on the blended colour)
    np.testing.assert_allclose(res[f"rgb_{typ}"].detach().numpy(), g["rgb"], rtol=0, atol=1e-5 if Fn else 3e-6)
    np.testing.assert_allclose(res[f"depth_{typ}"].detach().numpy(), g["depth"], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(res[f"depth_variance_{typ}"].numpy(), g["depth_variance"], rtol=1e-4, atol=1e-7)
    photo = F_mse(res[f"rgb_{typ}"], torch.from_numpy(rgbs))
    gl = res["gate_loss_coarse"].mean()
    if Fn:
        gl = (res["gate_loss_fine"].mean() + gl) / 2
    loss = photo + 5e-4 * gl
    np.testing.assert_allclose(loss.detach().numpy(), g["loss"], rtol=2e-6)
    loss.backward()
    for pre, params in (("", p), ("bg__", pb)):
        for k, t in params.items():
            ref_sum = g["gsum__" + pre + k]
            got = t.grad.numpy()
            scale = max(1e-12, float(ref_sum[1]))
            assert abs(synth.checksum(got)[0] - ref_sum[0]) <= 5e-4 * scale + 1e-9, pre + k
            sl = got.reshape(-1)[:: max(1, got.size // 499)][:499]
            ref = g["gslice__" + pre + k]
            np.testing.assert_allclose(sl, ref, rtol=2e-3, atol=1e-7 + 2e-4 * np.abs(ref).max(), err_msg=pre + k)


def test_hash_encode_restatement_properties():
    """Hash-grid encoding (no reference counterpart, "parity unpinned"): properties of the restatement itself - trilinear
    weights sum to one, grid points reproduce their table entry, the encoding is continuous across cell faces, dense levels
    index without collisions."""
    hc = dict(n_levels=6, log2_table=10, base_res=2, per_level_scale=2.0, aabb_lo=(0.0, 0.0, 0.0), aabb_hi=(1.0, 1.0, 1.0))
    T = 1 << hc["log2_table"]
    x = torch.from_numpy(np.random.default_rng(5).uniform(0, 1, (500, 3)).astype(np.float32))
    ones = torch.ones(hc["n_levels"], T, 2)
    np.testing.assert_allclose(O.hash_encode(x, ones, hc).numpy(), 1.0, atol=1e-6)
    table = torch.from_numpy(np.random.default_rng(6).standard_normal((hc["n_levels"], T, 2)).astype(np.float32))
    lv = O.hash_levels(hc)
    assert [d for _, _, d in lv] == [True, True, True, False, False, False]     # 3^3, 4^3, 6^3 <= 1024 < 10^3
    s, r, _ = lv[1]                                                             # level 1: scale 3, 5 grid points... r = 5
    assert (float(s), r) == (3.0, 5)
    # a grid point of level 1: pos = x * 3 + 0.5 integral -> weight 1 on one corner
    g = torch.tensor([[(2 - 0.5) / 3.0, (1 - 0.5) / 3.0, (3 - 0.5) / 3.0]], dtype=torch.float32)
    got = O.hash_encode(g, table, hc)[0, 2:4]
    np.testing.assert_allclose(got.numpy(), table[1][2 + 5 * (1 + 5 * 3)].numpy(), atol=2e-6)
    # continuity across a cell face
    a = O.hash_encode(torch.tensor([[0.5 - 1e-6, 0.3, 0.7]]), table, hc)
    b = O.hash_encode(torch.tensor([[0.5 + 1e-6, 0.3, 0.7]]), table, hc)
    assert (a - b).abs().max().item() < 1e-3


def test_nobatch_dispatcher_vs_reference_golden():
    """The evaluation path's dispatcher (tutel_fast_dispatch_nobatch.py: extract_critical, encode, decode with autograd) - the
    oracle's restatement against tensors produced by the reference's own classes: indices / locations / expert_input_nums /
    expert_locations_begin bit-exact, dispatched rows, decode output and the three gradients to fp32 rounding."""
    g = np.load(os.path.join(G, "dispatch_nobatch_plain.npz"))
    r = O.route_top1_nobatch(g["gates"])
    assert np.array_equal(r["idx"], g["indices"]) and np.array_equal(r["loc"], g["locations"])
    assert np.array_equal(r["expert_input_nums"], g["expert_input_nums"]) and np.array_equal(r["expert_locations_begin"], g["expert_locations_begin"])
    gates = torch.from_numpy(g["gates"]).requires_grad_(True)
    x = torch.from_numpy(g["x"]).requires_grad_(True)
    idx, loc, begin = (torch.from_numpy(r[k].astype(np.int64)) for k in ("idx", "loc", "expert_locations_begin"))
    d = O.dispatch_nobatch(x, idx, loc, begin)
    np.testing.assert_allclose(d.detach().numpy(), g["dispatched"], rtol=0, atol=0)
    eo = torch.tanh(d @ torch.from_numpy(g["w"]))
    eo.retain_grad()
    gate_s = gates.gather(1, idx.unsqueeze(1)).squeeze(1)
    y = O.combine_nobatch(eo, idx, loc, begin, gate_s)
    np.testing.assert_allclose(y.detach().numpy(), g["y"], rtol=1e-6, atol=1e-7)
    (y * torch.from_numpy(g["dy"])).sum().backward()
    np.testing.assert_allclose(x.grad.numpy(), g["dx"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(gates.grad.numpy(), g["dgates"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(eo.grad.numpy(), g["d_expert_out"], rtol=1e-6, atol=1e-7)
