"""The stand-alone MoE layer mirror (switch_nerf_amd.moe.MoELayer) under torch autograd against the golden vectors of the
reference's own `moe_layer` module (oracle/gen_golden.py gen_moe_layer: forward, l_aux, top-1 indices, input / gate-input /
parameter gradients, and the gradients of l_aux)."""
import os

import numpy as np
import pytest
import torch

import synth

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def _layer(dtype, **kw):
    from switch_nerf_amd.moe import moe_layer
    cfg = synth.BUILDING
    return moe_layer(gate_type=dict(type="top", k=1, fp32_gate=True, capacity_factor=1.0, batch_prioritized_routing=True, gate_noise=-1.0,
                                    compute_balance_loss=False, dispatcher_no_score=False, is_postscore=True, gate_dim=cfg["gate_hidden"]),
                     model_dim=cfg["model_dim"],
                     experts=dict(type="expertmlp", count_per_node=cfg["num_experts"], hidden_size_per_expert=cfg["model_dim"],
                                  layer_num=cfg["expert_layers"], skips=list(cfg["skips"])),
                     seeds=(1, 1, 1), return_gates=True, dtype=dtype, **kw).cuda()


def _load(moe, seed):
    sd = synth.make_weights(seed, synth.BUILDING)
    sub = {k[len("layers.0."):]: torch.from_numpy(v) for k, v in sd.items() if k.startswith("layers.0.")}
    assert set(sub) == set(moe.state_dict()), sorted(set(sub) ^ set(moe.state_dict()))
    moe.load_state_dict(sub)


def test_moe_layer_vs_reference_golden_fp32():
    g = np.load(os.path.join(G, "moe_layer_m256e8.npz"))
    seed, P = int(g["seed"]), int(g["P"])
    moe = _layer(torch.float32)
    _load(moe, seed)
    rng = np.random.default_rng(seed + 1000)
    x = rng.standard_normal((P, 256)).astype(np.float32)
    gi = rng.standard_normal((P, 256)).astype(np.float32)
    xt = torch.from_numpy(x).cuda().requires_grad_(True)
    gt = torch.from_numpy(gi).cuda().requires_grad_(True)
    y = moe(xt, gate_input=gt)
    np.testing.assert_array_equal(y.gate_extras["gates"].cpu().numpy().reshape(-1), g["topk"].reshape(-1))      # bit-exact routing
    np.testing.assert_allclose(y.detach().cpu().numpy(), g["y"], rtol=1e-4, atol=5e-5)
    np.testing.assert_allclose(y.l_aux.item(), float(g["l_aux"]), rtol=1e-6)
    dy = rng.standard_normal((P, 256)).astype(np.float32)
    (y * torch.from_numpy(dy).cuda()).sum().backward(retain_graph=True)
    np.testing.assert_allclose(xt.grad.cpu().numpy(), g["dx"], rtol=1e-3, atol=2e-4 * np.abs(g["dx"]).max())
    np.testing.assert_allclose(gt.grad.cpu().numpy(), g["dgate_input"], rtol=1e-3, atol=2e-4 * np.abs(g["dgate_input"]).max())
    for n, p in moe.named_parameters():
        got = p.grad.cpu().numpy()
        if "grad__" + n in g and g["grad__" + n].shape == got.shape:
            ref = g["grad__" + n]
            np.testing.assert_allclose(got, ref, rtol=1e-3, atol=5e-4 * np.abs(ref).max(), err_msg=n)
        else:
            ref_sum = g["grad__" + n]
            scale = max(1e-12, float(ref_sum[1]))
            assert abs(synth.checksum(got)[0] - ref_sum[0]) <= 1e-3 * scale, n
            sl = got.reshape(-1)[:: max(1, got.size // 997)][:997]
            ref = g["gslice__" + n]
            np.testing.assert_allclose(sl, ref, rtol=2e-3, atol=5e-4 * np.abs(ref).max(), err_msg=n)
    # gradients of the load-balance loss alone
    d_g, d_wg = torch.autograd.grad(y.l_aux, [gt, moe.gates[0].wg.weight])
    np.testing.assert_allclose(d_g.cpu().numpy(), g["laux_dgate_input"], rtol=1e-3, atol=2e-4 * np.abs(g["laux_dgate_input"]).max())
    np.testing.assert_allclose(d_wg.cpu().numpy(), g["laux_dwg"], rtol=1e-3, atol=2e-4 * np.abs(g["laux_dwg"]).max())


def test_moe_layer_bf16_shapes_no_batch_and_errors():
    moe = _layer(torch.bfloat16)
    _load(moe, 33)
    gen = torch.Generator(device="cuda").manual_seed(7)
    x = torch.randn(4, 300, 256, device="cuda", generator=gen)        # leading dims are flattened like the reference (:741-745)
    gi = torch.randn(4, 300, 256, device="cuda", generator=gen)
    y = moe(x, gate_input=gi)
    assert y.shape == x.shape and y.dtype == x.dtype and y.l_aux.ndim == 0
    ref = _layer(torch.float32)
    ref.load_state_dict(moe.state_dict())
    y32 = ref(x, gate_input=gi)
    same = (y.gate_extras["gates"] == y32.gate_extras["gates"]).float().mean().item()
    assert same > 0.99                                                # bf16 gate inputs may flip near-ties
    agree = (y.gate_extras["gates"] == y32.gate_extras["gates"]).view(4, 300)        # same expert in both precisions
    kept = (y32.abs().sum(-1) > 0) & (y.abs().sum(-1) > 0) & agree
    assert (y - y32)[kept].abs().max().item() < 0.15 * y32.abs().max().item()
    dropped = (y32.abs().sum(-1) == 0).float().mean().item()
    assert dropped > 0.0                                              # capacity 1.0 on random-init gates drops tokens
    nb = _layer(torch.float32, moe_no_batch=True)
    nb.load_state_dict(moe.state_dict())
    with torch.no_grad():
        ynb = nb(x, gate_input=gi)
    assert (ynb.abs().sum(-1) == 0).float().mean().item() == 0.0      # eval path: nothing is dropped
    with pytest.raises(RuntimeError, match="HIP library only"):
        moe(x.cpu(), gate_input=gi.cpu())
    from switch_nerf_amd.moe import moe_layer
    with pytest.raises(NotImplementedError):
        moe_layer(gate_type=dict(type="cosine_top", k=1), model_dim=256, experts=dict(type="expertmlp", count_per_node=8, layer_num=7))
    with pytest.raises(ValueError, match="exceeds"):
        moe_layer(gate_type=dict(type="top", k=9), model_dim=256, experts=dict(type="expertmlp", count_per_node=8, layer_num=7))


def test_moe_layer_gate_noise_vs_reference_golden_fp32():
    """--gate_noise > 0 (opts.py:208; tutel_moe_layer_nobatch.py:119-122: in training the router's logits get gate_noise * randn / E before
    the softmax) against the REFERENCE layer's own run with the same noise draw (replayed from its seed by oracle/gen_golden.py, which
    asserts that the replay reproduces the layer's routing): top-1 indices bit-exact, y, l_aux, input / gate-input / router gradients and
    the gradients of l_aux alone; an evaluation forward adds no noise."""
    g = np.load(os.path.join(G, "moe_layer_noise_m256e8.npz"))
    seed, P, gn = int(g["seed"]), int(g["P"]), float(g["gate_noise"])
    from switch_nerf_amd.moe import moe_layer
    cfg = synth.BUILDING
    moe = moe_layer(gate_type=dict(type="top", k=1, fp32_gate=True, capacity_factor=1.0, batch_prioritized_routing=True, gate_noise=gn,
                                   gate_dim=cfg["gate_hidden"]), model_dim=cfg["model_dim"],
                    experts=dict(type="expertmlp", count_per_node=cfg["num_experts"], hidden_size_per_expert=cfg["model_dim"],
                                 layer_num=cfg["expert_layers"], skips=list(cfg["skips"])), seeds=(1, 1, 1), return_gates=True,
                    dtype=torch.float32).cuda()
    _load(moe, seed)
    moe.train()
    rng = np.random.default_rng(seed + 1000)
    x = rng.standard_normal((P, 256)).astype(np.float32)
    gi = rng.standard_normal((P, 256)).astype(np.float32)
    xt = torch.from_numpy(x).cuda().requires_grad_(True)
    gt = torch.from_numpy(gi).cuda().requires_grad_(True)
    y = moe(xt, gate_input=gt, gate_noise_draw=torch.from_numpy(g["noise"]))
    np.testing.assert_array_equal(y.gate_extras["gates"].cpu().numpy().reshape(-1), g["topk"].reshape(-1))
    np.testing.assert_allclose(y.detach().cpu().numpy(), g["y"], rtol=1e-4, atol=5e-5)
    np.testing.assert_allclose(y.l_aux.item(), float(g["l_aux"]), rtol=1e-6)
    dy = rng.standard_normal((P, 256)).astype(np.float32)
    (y * torch.from_numpy(dy).cuda()).sum().backward(retain_graph=True)
    np.testing.assert_allclose(xt.grad.cpu().numpy(), g["dx"], rtol=1e-3, atol=2e-4 * np.abs(g["dx"]).max())
    np.testing.assert_allclose(gt.grad.cpu().numpy(), g["dgate_input"], rtol=1e-3, atol=2e-4 * np.abs(g["dgate_input"]).max())
    np.testing.assert_allclose(moe.gates[0].wg.weight.grad.cpu().numpy(), g["dwg"], rtol=1e-3, atol=5e-4 * np.abs(g["dwg"]).max())
    d_g, d_wg = torch.autograd.grad(y.l_aux, [gt, moe.gates[0].wg.weight])
    np.testing.assert_allclose(d_g.cpu().numpy(), g["laux_dgate_input"], rtol=1e-3, atol=2e-4 * np.abs(g["laux_dgate_input"]).max())
    np.testing.assert_allclose(d_wg.cpu().numpy(), g["laux_dwg"], rtol=1e-3, atol=2e-4 * np.abs(g["laux_dwg"]).max())
    # evaluation: no noise (self.training is False) - the routing of the noise-free layer; a second training forward draws its own noise
    moe.eval()
    with torch.no_grad():
        ye = moe(xt, gate_input=gt)
    assert (ye.gate_extras["gates"].cpu().numpy().reshape(-1) != g["topk"].reshape(-1)).any()
    moe.train()
    with torch.no_grad():
        a, b = moe(xt, gate_input=gt), moe(xt, gate_input=gt)
    assert (a.gate_extras["gates"] != b.gate_extras["gates"]).any()                 # two draws, two routings
    with pytest.raises(AssertionError, match="gate_noise"):          # (the reference's own assert, tutel_fast_dispatch.py:154)
        moe_layer(gate_type=dict(type="top", k=1, use_load_importance_loss=True), model_dim=256,
                  experts=dict(type="expertmlp", count_per_node=8, layer_num=7))


def test_moe_layer_normal_noise_vs_reference_golden_fp32():
    """use_normal_noise (tutel_moe_layer_nobatch.py:116-117: `logits + randn / E` in training) together with --gate_noise: against the
    REFERENCE layer's own run with both of its draws replayed (oracle/gen_golden.py::gen_moe_layer_normal_noise): top-1 indices
    bit-exact, y, l_aux and the gradients; the two additive terms reach the router kernel as one noise operand."""
    g = np.load(os.path.join(G, "moe_layer_normal_noise_m256e8.npz"))
    seed, P, gn = int(g["seed"]), int(g["P"]), float(g["gate_noise"])
    from switch_nerf_amd.moe import moe_layer
    cfg = synth.BUILDING
    moe = moe_layer(gate_type=dict(type="top", k=1, fp32_gate=True, capacity_factor=1.0, batch_prioritized_routing=True, gate_noise=gn,
                                   use_normal_noise=True, gate_dim=cfg["gate_hidden"]), model_dim=cfg["model_dim"],
                    experts=dict(type="expertmlp", count_per_node=cfg["num_experts"], hidden_size_per_expert=cfg["model_dim"],
                                 layer_num=cfg["expert_layers"], skips=list(cfg["skips"])), seeds=(1, 1, 1), return_gates=True,
                    dtype=torch.float32).cuda()
    _load(moe, seed)
    moe.train()
    rng = np.random.default_rng(seed + 1000)
    x = rng.standard_normal((P, 256)).astype(np.float32)
    gi = rng.standard_normal((P, 256)).astype(np.float32)
    xt = torch.from_numpy(x).cuda().requires_grad_(True)
    gt = torch.from_numpy(gi).cuda().requires_grad_(True)
    y = moe(xt, gate_input=gt, gate_noise_draw=torch.from_numpy(g["noise"]), normal_noise_draw=torch.from_numpy(g["normal_noise"]))
    np.testing.assert_array_equal(y.gate_extras["gates"].cpu().numpy().reshape(-1), g["topk"].reshape(-1))
    np.testing.assert_allclose(y.detach().cpu().numpy(), g["y"], rtol=1e-4, atol=5e-5)
    np.testing.assert_allclose(y.l_aux.item(), float(g["l_aux"]), rtol=1e-6)
    dy = rng.standard_normal((P, 256)).astype(np.float32)
    (y * torch.from_numpy(dy).cuda()).sum().backward()
    np.testing.assert_allclose(xt.grad.cpu().numpy(), g["dx"], rtol=1e-3, atol=2e-4 * np.abs(g["dx"]).max())
    np.testing.assert_allclose(gt.grad.cpu().numpy(), g["dgate_input"], rtol=1e-3, atol=2e-4 * np.abs(g["dgate_input"]).max())
    np.testing.assert_allclose(moe.gates[0].wg.weight.grad.cpu().numpy(), g["dwg"], rtol=1e-3, atol=5e-4 * np.abs(g["dwg"]).max())
    moe.eval()                   # evaluation: neither noise
    with torch.no_grad():
        ye = moe(xt, gate_input=gt)
    assert (ye.gate_extras["gates"].cpu().numpy().reshape(-1) != g["topk"].reshape(-1)).any()


def _top2_layer(cfg, bpr, cf, dtype):
    from switch_nerf_amd.moe import moe_layer
    return moe_layer(gate_type=dict(type="top", k=2, fp32_gate=True, capacity_factor=cf, batch_prioritized_routing=bpr, gate_noise=-1.0,
                                    gate_dim=cfg["gate_hidden"]), model_dim=cfg["model_dim"],
                     experts=dict(type="expertmlp", count_per_node=cfg["num_experts"], hidden_size_per_expert=cfg["model_dim"],
                                  layer_num=cfg["expert_layers"], skips=list(cfg["skips"])), seeds=(1, 1, 1), return_gates=True,
                     dtype=dtype).cuda()


def test_moe_layer_top2_vs_reference_golden_fp32():
    """`k: 2` (extract_critical with top_k > 1, tutel_fast_dispatch.py:176-217; the encoder / decoder loops over the choices, :17-78)
    against the REFERENCE layer's own run with top_k = 2: both choices' experts bit-exact, output, l_aux, input / gate-input / parameter
    gradients and the gradients of l_aux alone; underneath, swn_topk_select / swn_route_topk give the reference's locations (acc_base,
    batch-prioritised by the token's max gate) bit for bit and its normalised gates."""
    from switch_nerf_amd import ops
    g = np.load(os.path.join(G, "moe_layer_top2_m256e8_bpr.npz"))
    cfg = synth.BUILDING
    seed, P, bpr, cf, cap = int(g["seed"]), int(g["P"]), bool(g["bpr"]), float(g["cf"]), int(g["capacity"])
    moe = _top2_layer(cfg, bpr, cf, torch.float32)
    _load(moe, seed)
    rng = np.random.default_rng(seed + 1000)
    x = rng.standard_normal((P, 256)).astype(np.float32)
    gi = rng.standard_normal((P, 256)).astype(np.float32)
    xt = torch.from_numpy(x).cuda().requires_grad_(True)
    gt = torch.from_numpy(gi).cuda().requires_grad_(True)
    # the routing kernels on their own
    gates, idx0, gmax, _ = ops.gate_fwd(gt.detach(), None, None, moe.gates[0].wg.weight.detach().float().contiguous())
    idx, gsel, gn = ops.topk_select(gates, 2)
    np.testing.assert_array_equal(idx.cpu().numpy().T, g["topk"])
    np.testing.assert_array_equal(idx[0].cpu().numpy(), idx0.cpu().numpy())
    np.testing.assert_allclose(gn.cpu().numpy(), g["gnorm"], rtol=1e-5)
    loc, counts, perm, tok2row, group_rows, l_aux = ops.route_topk(idx, gmax, gates, P, 8, cap, bpr, want_tok2row=True)
    np.testing.assert_array_equal(loc.cpu().numpy(), g["loc"])                                   # bit-exact locations, both choices
    np.testing.assert_array_equal(group_rows.cpu().numpy(), counts.sum(0).view(-1).cpu().numpy())
    locn, idxn, permn, t2r = g["loc"], g["topk"].T, perm.cpu().numpy().reshape(-1), tok2row.cpu().numpy()
    for j in range(2):                                                                           # every kept (token, choice) owns its row
        kept = np.nonzero(locn[j] < cap)[0]
        rows = idxn[j][kept] * cap + locn[j][kept]
        np.testing.assert_array_equal(permn[rows], kept)
        np.testing.assert_array_equal(t2r[j][kept], rows)
        assert (t2r[j][locn[j] >= cap] == -1).all()
    assert (permn >= 0).sum() == (locn < cap).sum()
    # the layer
    y = moe(xt, gate_input=gt)
    np.testing.assert_array_equal(y.gate_extras["gates"].cpu().numpy(), g["topk"])
    np.testing.assert_allclose(y.detach().cpu().numpy(), g["y"], rtol=1e-4, atol=5e-5)
    np.testing.assert_allclose(y.l_aux.item(), float(g["l_aux"]), rtol=1e-6)
    dy = rng.standard_normal((P, 256)).astype(np.float32)
    (y * torch.from_numpy(dy).cuda()).sum().backward(retain_graph=True)
    np.testing.assert_allclose(xt.grad.cpu().numpy(), g["dx"], rtol=1e-3, atol=2e-4 * np.abs(g["dx"]).max())
    np.testing.assert_allclose(gt.grad.cpu().numpy(), g["dgate_input"], rtol=1e-3, atol=2e-4 * np.abs(g["dgate_input"]).max())
    for n, p in moe.named_parameters():
        got = p.grad.cpu().numpy()
        if "grad__" + n in g and g["grad__" + n].shape == got.shape:
            ref = g["grad__" + n]
            np.testing.assert_allclose(got, ref, rtol=1e-3, atol=5e-4 * np.abs(ref).max(), err_msg=n)
        else:
            ref_sum = g["grad__" + n]
            scale = max(1e-12, float(ref_sum[1]))
            assert abs(synth.checksum(got)[0] - ref_sum[0]) <= 1e-3 * scale, n
            sl = got.reshape(-1)[:: max(1, got.size // 997)][:997]
            ref = g["gslice__" + n]
            np.testing.assert_allclose(sl, ref, rtol=2e-3, atol=5e-4 * np.abs(ref).max(), err_msg=n)
    d_g, d_wg = torch.autograd.grad(y.l_aux, [gt, moe.gates[0].wg.weight])
    np.testing.assert_allclose(d_g.cpu().numpy(), g["laux_dgate_input"], rtol=1e-3, atol=2e-4 * np.abs(g["laux_dgate_input"]).max())
    np.testing.assert_allclose(d_wg.cpu().numpy(), g["laux_dwg"], rtol=1e-3, atol=2e-4 * np.abs(g["laux_dwg"]).max())


def test_moe_layer_top2_plain_locations_small_and_bf16():
    """The second fixture (64 features, 4 experts, token-order locations, capacity factor 0.75) through the routing kernels and the
    fp32 layer's forward; then the 16-bit top-2 layer on the building shapes (persistent 256-row chains: capacity >= 256) against the
    fp32 one - same experts off near-ties, outputs within 16-bit rounding; a capacity-limited 16-bit step has finite gradients; no-batch
    evaluation drops nothing."""
    from switch_nerf_amd import ops
    g = np.load(os.path.join(G, "moe_layer_top2_m64e4_plain.npz"))
    cfg = synth.small_cfg(64, 4)
    seed, P, bpr, cf, cap = int(g["seed"]), int(g["P"]), bool(g["bpr"]), float(g["cf"]), int(g["capacity"])
    sd = synth.make_weights(seed, cfg)
    wg = torch.from_numpy(sd["layers.0.gates.0.wg.weight"]).cuda()
    rng = np.random.default_rng(seed + 1000)
    x = rng.standard_normal((P, 64)).astype(np.float32)
    gi = torch.from_numpy(rng.standard_normal((P, 64)).astype(np.float32)).cuda()
    logits = gi @ wg.t()
    gates = torch.softmax(logits, dim=1).contiguous()                        # (64-feature routers have no kernel shape: test input only)
    gmax = gates.max(dim=1).values.contiguous()
    idx, gsel, gn = ops.topk_select(gates, 2)
    np.testing.assert_array_equal(idx.cpu().numpy().T, g["topk"])
    loc, counts, perm, _, group_rows, l_aux = ops.route_topk(idx, gmax, gates, P, 4, cap, bpr)
    np.testing.assert_array_equal(loc.cpu().numpy(), g["loc"])
    np.testing.assert_allclose(gn.cpu().numpy(), g["gnorm"], rtol=1e-5)
    np.testing.assert_allclose(l_aux.item(), float(g["l_aux"]), rtol=1e-5)
    # 16-bit layer on the building shapes
    cfg = synth.BUILDING
    # (capacity factor 4: nothing is dropped, so a token whose experts agree is the same computation in both precisions - with drops a
    # rank that moves by one place in 16 bits keeps a (token, choice) pair in one layer and drops it in the other)
    m16, m32 = _top2_layer(cfg, True, 4.0, torch.bfloat16), _top2_layer(cfg, True, 4.0, torch.float32)
    _load(m16, 36)
    m32.load_state_dict(m16.state_dict())
    gen = torch.Generator(device="cuda").manual_seed(11)
    xb = torch.randn(4096, 256, device="cuda", generator=gen).requires_grad_(True)
    gb = torch.randn(4096, 256, device="cuda", generator=gen)
    y16, y32 = m16(xb, gate_input=gb), m32(xb, gate_input=gb)
    same = (y16.gate_extras["gates"] == y32.gate_extras["gates"]).all(dim=1)
    assert same.float().mean().item() > 0.97
    err = (y16 - y32)[same].abs().max().item()
    assert err < 0.05 * y32.abs().max().item(), err
    mc = _top2_layer(cfg, True, 0.5, torch.bfloat16)
    mc.load_state_dict(m16.state_dict())
    yc = mc(xb, gate_input=gb)
    assert 0.0 < (yc.abs().sum(-1) == 0).float().mean().item() < 0.9          # capacity factor 0.5: some tokens lose both choices
    (yc.float().square().sum() + yc.l_aux).backward()
    assert torch.isfinite(xb.grad).all() and xb.grad.abs().max().item() > 0
    for p_ in mc.parameters():
        assert p_.grad is not None and torch.isfinite(p_.grad).all()
    nb = _top2_layer(cfg, True, 1.0, torch.float32)
    nb.moe_no_batch = True
    nb.load_state_dict(m16.state_dict())
    with torch.no_grad():
        ynb = nb(xb, gate_input=gb)
        kept_rows = (ynb.abs().sum(-1) > 0).float().mean().item()
    assert kept_rows == 1.0


@pytest.mark.parametrize("tag", ["k1", "k2"])
def test_moe_layer_load_importance_vs_reference_golden_fp32(tag):
    """--use_load_importance_loss (opts.py:210; load_importance_loss / extract_critical_load_importance, tutel_fast_dispatch.py:152-174,
    219-265) with gate noise in training mode and --compute_balance_loss, against the REFERENCE layer's own run (top-1 and top-2 gate,
    noise draw replayed): experts bit-exact, output, the load / importance loss as l_aux, the load-balance loss in the extras, the
    gradients of the output and of both losses."""
    g = np.load(os.path.join(G, f"moe_layer_load_importance_{tag}.npz"))
    seed, P, gn, K = int(g["seed"]), int(g["P"]), float(g["gate_noise"]), int(g["top_k"])
    from switch_nerf_amd.moe import moe_layer
    cfg = synth.BUILDING
    mk = lambda **kw: moe_layer(gate_type=dict(type="top", k=K, fp32_gate=True, capacity_factor=1.0, batch_prioritized_routing=True,
                                               gate_dim=cfg["gate_hidden"], **kw), model_dim=cfg["model_dim"],
                                experts=dict(type="expertmlp", count_per_node=cfg["num_experts"], hidden_size_per_expert=cfg["model_dim"],
                                             layer_num=cfg["expert_layers"], skips=list(cfg["skips"])), seeds=(1, 1, 1), return_gates=True,
                                dtype=torch.float32)
    with pytest.raises(AssertionError, match="gate_noise"):
        mk(use_load_importance_loss=True, gate_noise=-1.0)
    with pytest.raises(ValueError, match="compute_balance_loss"):
        mk(compute_balance_loss=True, gate_noise=gn)
    moe = mk(use_load_importance_loss=True, compute_balance_loss=True, gate_noise=gn).cuda()
    _load(moe, seed)
    moe.train()
    rng = np.random.default_rng(seed + 1000)
    x = rng.standard_normal((P, 256)).astype(np.float32)
    gi = rng.standard_normal((P, 256)).astype(np.float32)
    xt = torch.from_numpy(x).cuda().requires_grad_(True)
    gt = torch.from_numpy(gi).cuda().requires_grad_(True)
    wgp = moe.gates[0].wg.weight
    y = moe(xt, gate_input=gt, gate_noise_draw=torch.from_numpy(g["noise"]))
    np.testing.assert_array_equal(y.gate_extras["gates"].cpu().numpy(), g["topk"])
    np.testing.assert_allclose(y.detach().cpu().numpy(), g["y"], rtol=1e-4, atol=5e-5)
    np.testing.assert_allclose(y.l_aux.item(), float(g["l_aux"]), rtol=2e-4)
    bal = y.gate_extras["balance_loss"]
    np.testing.assert_allclose(bal.item(), float(g["balance_loss"]), rtol=1e-6)
    dy = rng.standard_normal((P, 256)).astype(np.float32)
    (y * torch.from_numpy(dy).cuda()).sum().backward(retain_graph=True)
    np.testing.assert_allclose(xt.grad.cpu().numpy(), g["dx"], rtol=1e-3, atol=2e-4 * np.abs(g["dx"]).max())
    np.testing.assert_allclose(gt.grad.cpu().numpy(), g["dgate_input"], rtol=1e-3, atol=2e-4 * np.abs(g["dgate_input"]).max())
    np.testing.assert_allclose(wgp.grad.cpu().numpy(), g["dwg"], rtol=1e-3, atol=5e-4 * np.abs(g["dwg"]).max())
    d_g, d_wg = torch.autograd.grad(y.l_aux, [gt, wgp], retain_graph=True)
    np.testing.assert_allclose(d_g.cpu().numpy(), g["laux_dgate_input"], rtol=2e-3, atol=1e-3 * np.abs(g["laux_dgate_input"]).max())
    np.testing.assert_allclose(d_wg.cpu().numpy(), g["laux_dwg"], rtol=2e-3, atol=1e-3 * np.abs(g["laux_dwg"]).max())
    b_g, b_wg = torch.autograd.grad(bal, [gt, wgp])
    np.testing.assert_allclose(b_g.cpu().numpy(), g["bal_dgate_input"], rtol=1e-3, atol=2e-4 * np.abs(g["bal_dgate_input"]).max())
    np.testing.assert_allclose(b_wg.cpu().numpy(), g["bal_dwg"], rtol=1e-3, atol=2e-4 * np.abs(g["bal_dwg"]).max())
    # evaluation: no noise, the threshold is the clean k-th logit - the loss still evaluates (tutel_moe_layer_nobatch.py:119-122)
    moe.eval()
    with torch.no_grad():
        ye = moe(xt, gate_input=gt)
    assert np.isfinite(ye.l_aux.item()) and ye.l_aux.item() != y.l_aux.item()


@pytest.mark.parametrize("K", [1, 2])
def test_moe_layer_load_importance_bf16_and_larger_batch(K):
    """The load / importance branch in the 16-bit layer (persistent 256-row chains, 16-bit router input) at 4096 tokens against the fp32 layer
    with the same weights and the same noise draw: the loss within 16-bit rounding of the gate input, finite gradients everywhere, the
    loss's own gradient reaches the gate weight; sigma = gate_noise / E is what normalises it (tutel_fast_dispatch.py:154-160)."""
    from switch_nerf_amd.moe import moe_layer
    cfg = synth.BUILDING
    mk = lambda dt: moe_layer(gate_type=dict(type="top", k=K, fp32_gate=True, capacity_factor=2.0, batch_prioritized_routing=True,
                                             gate_dim=cfg["gate_hidden"], use_load_importance_loss=True, compute_balance_loss=True, gate_noise=1.0),
                              model_dim=cfg["model_dim"],
                              experts=dict(type="expertmlp", count_per_node=cfg["num_experts"], hidden_size_per_expert=cfg["model_dim"],
                                           layer_num=cfg["expert_layers"], skips=list(cfg["skips"])), seeds=(1, 1, 1), return_gates=True,
                              dtype=dt).cuda()
    m16, m32 = mk(torch.bfloat16), mk(torch.float32)
    _load(m16, 41)
    m32.load_state_dict(m16.state_dict())
    m16.train(), m32.train()
    gen = torch.Generator(device="cuda").manual_seed(5)
    P = 4096
    x = torch.randn(P, 256, device="cuda", generator=gen).requires_grad_(True)
    gi = torch.randn(P, 256, device="cuda", generator=gen).requires_grad_(True)
    draw = torch.randn(P, cfg["num_experts"], device="cuda", generator=gen)
    y16, y32 = m16(x, gate_input=gi, gate_noise_draw=draw), m32(x, gate_input=gi, gate_noise_draw=draw)
    l16, l32 = y16.l_aux.item(), y32.l_aux.item()
    assert np.isfinite(l16) and l32 > 0 and abs(l16 - l32) <= 0.05 * l32, (l16, l32)
    b16, b32 = y16.gate_extras["balance_loss"].item(), y32.gate_extras["balance_loss"].item()
    assert abs(b16 - b32) <= 0.02 * b32, (b16, b32)
    (y16.float().square().mean() + y16.l_aux + y16.gate_extras["balance_loss"]).backward()
    assert torch.isfinite(x.grad).all() and torch.isfinite(gi.grad).all() and gi.grad.abs().max().item() > 0
    for p_ in m16.parameters():
        assert p_.grad is not None and torch.isfinite(p_.grad).all()
    wg16 = m16.gates[0].wg.weight
    d16, = torch.autograd.grad(m16(x, gate_input=gi, gate_noise_draw=draw).l_aux, [wg16])
    d32, = torch.autograd.grad(m32(x, gate_input=gi, gate_noise_draw=draw).l_aux, [m32.gates[0].wg.weight])
    assert d32.abs().max().item() > 0
    cos = torch.nn.functional.cosine_similarity(d16.flatten().float(), d32.flatten().float(), dim=0).item()
    assert cos > 0.98, cos
