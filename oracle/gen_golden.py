"""Generate tests/golden/*.npz by importing and running the REFERENCE itself (CPU, this container only).

Usage (build container only; /root/reference does not exist on the GPU box):
    python oracle/gen_golden.py

TEST INFRASTRUCTURE ONLY.  Nothing from the reference is copied: the fixtures hold inputs and the
reference's numeric outputs.  `tutel`/`timm` (absent third-party dependencies) are replaced by the
test-only stubs in oracle/stubs/ (see its README.md for the semantics adopted, "parity unpinned").
"""
import argparse
import os
import sys
import time
from argparse import Namespace

import numpy as np
import torch
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference"
sys.path.insert(0, os.path.join(HERE, "stubs"))
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import synth  # noqa: E402
from switch_nerf.models import model_utils  # noqa: E402
from switch_nerf.models.nerf import Embedding  # noqa: E402
from switch_nerf import rendering  # noqa: E402
from switch_nerf import rendering_mip  # noqa: E402
from switch_nerf.modules.tutel_moe_ext import tutel_fast_dispatch as tfd  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def building_model_cfg(cfg):
    """building.yaml's `model:` block with sizes taken from cfg (so that small fixtures can use small dims)."""
    with open(os.path.join(REF, "switch_nerf/configs/switch_nerf/building.yaml")) as f:
        y = yaml.safe_load(f)
    m = y["model"]
    M = cfg["model_dim"]
    L = m["layers"]
    L["xyz"]["out_ch"] = M
    L["0"].update(in_ch=M, h_ch=M, out_ch=M, num=cfg["expert_layers"], skips=list(cfg["skips"]), gate_dim=cfg["gate_hidden"])
    L["1"].update(in_ch=M, out_ch=M)
    L["2"].update(in_ch=M + 27 + cfg["appearance_dim"], out_ch=cfg["layer2_out"])
    L["sigma"].update(in_ch=M)
    L["color"].update(in_ch=cfg["layer2_out"])
    L["moe_external_gate"].update(in_ch=M, h_ch=cfg["gate_hidden"], out_ch=cfg["gate_hidden"])
    L["gate_input_norm"].update(in_ch=cfg["gate_hidden"])
    return m


def make_hparams(cfg, capacity_factor=1.0, bpr=True, coarse=256, chunk=131072, perturb=0.0,
                 sigma_noise=False, fine=0, mip=False):
    h = Namespace(
        container_path=None, use_cascade=False, train_mega_nerf=None, use_moe=True, ckpt_path=None,
        pos_xyz_dim=cfg["pos_xyz_dim"], pos_dir_dim=cfg["pos_dir_dim"], appearance_dim=cfg["appearance_dim"],
        affine_appearance=False, sh_deg=None, shifted_softplus=True, layer_dim=cfg["model_dim"],
        moe_expert_num=cfg["num_experts"], moe_local_expert_num=cfg["num_experts"],
        moe_capacity_factor=capacity_factor, batch_prioritized_routing=bpr, gate_noise=-1.0,
        compute_balance_loss=False, dispatcher_no_score=False, dispatcher_no_postscore=False,
        moe_expert_type="expertmlp", no_expert_parallel=True, single_data_group=None,
        parallel_env=Namespace(global_rank=0), moe_return_gates=True, moe_return_gate_logits=False,
        use_moe_external_gate=True, use_gate_input_norm=True, amp_use_bfloat16=False,
        nerfmoe_class_name="MipNeRFMoE" if mip else "NeRFMoE", model=building_model_cfg(cfg), perturb=perturb,
        use_mip=mip, weights_resample_padding=0.01, stop_level_grad=True, rgb_padding=0.001,
        coarse_samples=coarse, fine_samples=fine, model_chunk_size=chunk, use_sigma_noise=sigma_noise,
        sigma_noise_std=1.0, return_pts=False, return_pts_rgb=False, return_pts_alpha=False, return_sigma=True,
        return_alpha=False, bg_use_moe=False, use_load_importance_loss=False, white_bkgd=False,
        use_random_background_color=False, expertmlp2seqexperts=False, bg_use_cfg=False,
    )
    return h


def build_reference_model(cfg, sd_np, **kw):
    h = make_hparams(cfg, **kw)
    torch.manual_seed(0)
    nerf = model_utils.get_nerf(h, cfg["appearance_count"])
    ref_sd = nerf.state_dict()
    assert set(ref_sd.keys()) == set(sd_np.keys()), (sorted(set(ref_sd) ^ set(sd_np)))
    for k, v in sd_np.items():
        assert tuple(ref_sd[k].shape) == tuple(v.shape), (k, ref_sd[k].shape, v.shape)
    nerf.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd_np.items()})
    return nerf, h


def save(name, **arrs):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrs.items()})
    print(f"  wrote {path}  ({os.path.getsize(path)/1024:.1f} KiB)")


# ------------------------------------------------------------------------------------------ G1/G2 routing
def gen_routing():
    print("[G1] routing, tie-free, P=2048 E=8")
    gates = synth.make_gates(101, 2048, 8, logit_scale=1.0)
    g = torch.from_numpy(gates)
    assert len(np.unique(gates.max(1))) == 2048, "fixture must be tie-free"
    for bpr in (False, True):
        for cf in (1.0, 1.25, 0.5):
            crit, l_aux = tfd.extract_critical(g, 1, cf, True, bpr)
            E, idx, loc, gs, cap = crit
            save(f"route_p2048_e8_bpr{int(bpr)}_cf{cf}", seed=101, P=2048, E=8, logit_scale=1.0, bpr=bpr, cf=cf,
                 idx=idx[0].numpy().astype(np.int32), loc=loc[0].numpy().astype(np.int32), gate=gs[0].numpy(),
                 capacity=cap, l_aux=l_aux.numpy())
    print("[G1b] routing, ragged P=1000 (not a multiple of E=16 nor of 64)")
    gates = synth.make_gates(102, 1000, 16, logit_scale=2.0)
    assert len(np.unique(gates.max(1))) == 1000
    crit, l_aux = tfd.extract_critical(torch.from_numpy(gates), 1, 1.0, True, True)
    save("route_p1000_e16_bpr1_cf1.0", seed=102, P=1000, E=16, logit_scale=2.0, bpr=True, cf=1.0,
         idx=crit[1][0].numpy().astype(np.int32), loc=crit[2][0].numpy().astype(np.int32), gate=crit[3][0].numpy(),
         capacity=crit[4], l_aux=l_aux.numpy())
    print("[G2] routing at scale with exact ties, P=16384 E=8 (compared modulo tie groups)")
    gates = synth.make_gates(103, 16384, 8, logit_scale=1.0, quantize_bits=3)
    crit, l_aux = tfd.extract_critical(torch.from_numpy(gates), 1, 1.0, True, True)
    save("route_ties_p16384_e8", seed=103, P=16384, E=8, logit_scale=1.0, quantize_bits=3, bpr=True, cf=1.0,
         idx=crit[1][0].numpy().astype(np.int32), loc=crit[2][0].numpy().astype(np.int32), gate=crit[3][0].numpy(),
         capacity=crit[4], l_aux=l_aux.numpy())


# ------------------------------------------------------------------------------------------ G8 PE
def gen_pe():
    print("[G8] positional encoding")
    rng = np.random.default_rng(8)
    x = (rng.uniform(-1, 1, size=(256, 3))).astype(np.float32)
    save("pe", x=x, pe12=Embedding(12)(torch.from_numpy(x)).numpy(), pe4=Embedding(4)(torch.from_numpy(x)).numpy())


# ------------------------------------------------------------------------------------------ G3 MoE layer
def gen_moe_layer():
    print("[G3] moe_layer fwd+bwd (small dims, weights in fixture-free form: regenerated from synth seed)")
    for tag, cfg, P, seed in (("m64e4", synth.small_cfg(64, 4), 1024, 31), ("m256e8", synth.BUILDING, 1024, 32)):
        sd = synth.make_weights(seed, cfg)
        nerf, h = build_reference_model(cfg, sd)
        moe = nerf.layers["0"]
        rng = np.random.default_rng(seed + 1000)
        x = rng.standard_normal((P, cfg["model_dim"])).astype(np.float32)
        gi = rng.standard_normal((P, cfg["gate_hidden"])).astype(np.float32)
        xt = torch.from_numpy(x).requires_grad_(True)
        gt = torch.from_numpy(gi).requires_grad_(True)
        y = moe(xt, gate_input=gt)
        l_aux = y.l_aux
        dy = rng.standard_normal(y.shape).astype(np.float32)
        (y * torch.from_numpy(dy)).sum().backward(retain_graph=True)
        grads = {n: p.grad.clone() for n, p in moe.named_parameters()}
        gx, gg = xt.grad.clone(), gt.grad.clone()
        for p in moe.parameters():
            p.grad = None
        gl = torch.autograd.grad(l_aux, [gt] + list(moe.gates.parameters()), allow_unused=True)
        out = dict(seed=seed, P=P, y=y.detach().numpy(), l_aux=l_aux.detach().numpy(), topk=y.gate_extras["gates"].numpy().astype(np.int32),
                   dx=gx.numpy(), dgate_input=gg.numpy(), laux_dgate_input=gl[0].numpy(), laux_dwg=gl[1].numpy())
        for n, g_ in grads.items():
            out["grad__" + n] = g_.numpy() if g_.numel() <= 4096 else synth.checksum(g_.numpy())
            if g_.numel() > 4096:
                out["gslice__" + n] = g_.numpy().reshape(-1)[:: max(1, g_.numel() // 997)][:997]
        save(f"moe_layer_{tag}", **out)


def gen_moe_layer_noise():
    """The gate-noise branch (--gate_noise > 0, opts.py:208; tutel_moe_layer_nobatch.py:119-122): in training the router's logits get
    gate_noise * randn_like(logits) / E before the softmax.  The noise is the FIRST draw of the layer's forward: re-seeding and drawing a
    [P, E] tensor replays it, and the fixture carries it so that the other side can be fed the same values."""
    print("[G3n] moe_layer with gate noise (training mode)")
    cfg, P, seed, gate_noise, rng_seed = synth.BUILDING, 512, 34, 1.0, 777
    sd = synth.make_weights(seed, cfg)
    nerf, h = build_reference_model(cfg, sd)
    moe = nerf.layers["0"]
    moe.train()
    gate = moe.gates[0]
    gate.gate_noise = gate_noise
    E = cfg["num_experts"]
    rng = np.random.default_rng(seed + 1000)
    x = rng.standard_normal((P, cfg["model_dim"])).astype(np.float32)
    gi = rng.standard_normal((P, cfg["gate_hidden"])).astype(np.float32)
    xt = torch.from_numpy(x).requires_grad_(True)
    gt = torch.from_numpy(gi).requires_grad_(True)
    torch.manual_seed(rng_seed)
    y = moe(xt, gate_input=gt)
    torch.manual_seed(rng_seed)
    noise = torch.randn(P, E)
    # the replay is right iff the noisy logits reproduce the layer's expert choice
    logits = gt.detach() @ gate.wg.weight.detach().float().t()
    top = torch.softmax(logits + gate_noise * noise / E, dim=1).argmax(1)
    assert torch.equal(top, y.gate_extras["gates"].view(-1)), "noise replay does not reproduce the reference's routing"
    assert not torch.equal(torch.softmax(logits, dim=1).argmax(1), top), "the noise must move some expert choices"
    dy = rng.standard_normal(y.shape).astype(np.float32)
    (y * torch.from_numpy(dy)).sum().backward(retain_graph=True)
    gl = torch.autograd.grad(y.l_aux, [gt] + list(moe.gates.parameters()), allow_unused=True)
    save("moe_layer_noise_m256e8", seed=seed, P=P, gate_noise=gate_noise, noise=noise.numpy(), y=y.detach().numpy(),
         l_aux=y.l_aux.detach().numpy(), topk=y.gate_extras["gates"].numpy().astype(np.int32), dx=xt.grad.numpy(),
         dgate_input=gt.grad.numpy(), dwg=gate.wg.weight.grad.numpy(), laux_dgate_input=gl[0].numpy(), laux_dwg=gl[1].numpy())


def gen_moe_layer_normal_noise():
    """use_normal_noise (tutel_moe_layer_nobatch.py:116-117: in training `logits + randn_like(logits) / E`, "Scaling Vision with Sparse
    Mixture of Experts") TOGETHER with gate_noise (:119-122): the layer draws the normal noise first, the gate noise second - re-seeding
    and drawing two [P, E] tensors replays both."""
    print("[G3nn] moe_layer with use_normal_noise + gate noise (training mode)")
    cfg, P, seed, gate_noise, rng_seed = synth.BUILDING, 512, 35, 0.5, 778
    sd = synth.make_weights(seed, cfg)
    nerf, h = build_reference_model(cfg, sd)
    moe = nerf.layers["0"]
    moe.train()
    gate = moe.gates[0]
    gate.gate_noise = gate_noise
    gate.use_normal_noise = True
    E = cfg["num_experts"]
    rng = np.random.default_rng(seed + 1000)
    x = rng.standard_normal((P, cfg["model_dim"])).astype(np.float32)
    gi = rng.standard_normal((P, cfg["gate_hidden"])).astype(np.float32)
    xt = torch.from_numpy(x).requires_grad_(True)
    gt = torch.from_numpy(gi).requires_grad_(True)
    torch.manual_seed(rng_seed)
    y = moe(xt, gate_input=gt)
    torch.manual_seed(rng_seed)
    n1 = torch.randn(P, E)
    n2 = torch.randn(P, E)
    logits = gt.detach() @ gate.wg.weight.detach().float().t()
    top = torch.softmax((logits + n1 / E) + gate_noise * n2 / E, dim=1).argmax(1)
    assert torch.equal(top, y.gate_extras["gates"].view(-1)), "noise replay does not reproduce the reference's routing"
    assert not torch.equal(torch.softmax(logits + gate_noise * n2 / E, dim=1).argmax(1), top), "the normal noise must move some expert choices"
    dy = rng.standard_normal(y.shape).astype(np.float32)
    (y * torch.from_numpy(dy)).sum().backward(retain_graph=True)
    gl = torch.autograd.grad(y.l_aux, [gt] + list(moe.gates.parameters()), allow_unused=True)
    save("moe_layer_normal_noise_m256e8", seed=seed, P=P, gate_noise=gate_noise, normal_noise=n1.numpy(), noise=n2.numpy(), y=y.detach().numpy(),
         l_aux=y.l_aux.detach().numpy(), topk=y.gate_extras["gates"].numpy().astype(np.int32), dx=xt.grad.numpy(),
         dgate_input=gt.grad.numpy(), dwg=gate.wg.weight.grad.numpy(), laux_dgate_input=gl[0].numpy(), laux_dwg=gl[1].numpy())


def gen_moe_layer_top2():
    """A top-2 gate (`k: 2` in the model yaml's moe block - every shipped config has 1; extract_critical with top_k > 1,
    tutel_fast_dispatch.py:176-217): the reference layer's own run with `top_k = 2`, plain and batch-prioritised locations."""
    print("[G3k] moe_layer with a top-2 gate")
    for tag, cfg, P, seed, bpr, cf in (("m256e8_bpr", synth.BUILDING, 768, 36, True, 1.0), ("m64e4_plain", synth.small_cfg(64, 4), 512, 37, False, 0.75)):
        sd = synth.make_weights(seed, cfg)
        nerf, h = build_reference_model(cfg, sd, bpr=bpr, capacity_factor=cf)
        moe = nerf.layers["0"]
        gate = moe.gates[0]
        gate.top_k = 2
        E = cfg["num_experts"]
        rng = np.random.default_rng(seed + 1000)
        x = rng.standard_normal((P, cfg["model_dim"])).astype(np.float32)
        gi = rng.standard_normal((P, cfg["gate_hidden"])).astype(np.float32)
        xt = torch.from_numpy(x).requires_grad_(True)
        gt = torch.from_numpy(gi).requires_grad_(True)
        y = moe(xt, gate_input=gt)
        topk = y.gate_extras["gates"]
        assert topk.shape == (P, 2)
        gates = torch.softmax(gt.detach() @ gate.wg.weight.detach().float().t(), dim=1)
        crit, _ = tfd.extract_critical(gates, 2, cf, True, bpr)
        cap = crit[4]
        loc = np.stack([l.numpy().astype(np.int32) for l in crit[2]])
        assert (loc >= cap).any(), "the fixture must drop some (token, choice) pairs"
        dy = rng.standard_normal(y.shape).astype(np.float32)
        (y * torch.from_numpy(dy)).sum().backward(retain_graph=True)
        grads = {n: p.grad.clone() for n, p in moe.named_parameters()}
        gx, gg = xt.grad.clone(), gt.grad.clone()
        gl = torch.autograd.grad(y.l_aux, [gt] + list(moe.gates.parameters()), allow_unused=True)
        out = dict(seed=seed, P=P, bpr=bpr, cf=cf, capacity=cap, y=y.detach().numpy(), l_aux=y.l_aux.detach().numpy(),
                   topk=topk.numpy().astype(np.int32), loc=loc, gnorm=np.stack([g.numpy() for g in crit[3]]),
                   dx=gx.numpy(), dgate_input=gg.numpy(), laux_dgate_input=gl[0].numpy(), laux_dwg=gl[1].numpy())
        for n, g_ in grads.items():
            out["grad__" + n] = g_.numpy() if g_.numel() <= 4096 else synth.checksum(g_.numpy())
            if g_.numel() > 4096:
                out["gslice__" + n] = g_.numpy().reshape(-1)[:: max(1, g_.numel() // 997)][:997]
        save(f"moe_layer_top2_{tag}", **out)


def gen_moe_layer_load_importance():
    """--use_load_importance_loss (opts.py:210; extract_critical_load_importance, tutel_fast_dispatch.py:219-265) with gate noise in
    training mode, --compute_balance_loss on: the layer's loss is the load / importance loss, the load-balance loss travels in the
    extras.  One run with the shipped top-1 gate, one with a top-2 gate; the noise draw is replayed like gen_moe_layer_noise."""
    print("[G3li] moe_layer with the load / importance loss")
    for tag, top_k, seed, rng_seed in (("k1", 1, 38, 779), ("k2", 2, 39, 780)):
        cfg, P, gate_noise = synth.BUILDING, 512, 1.0
        sd = synth.make_weights(seed, cfg)
        nerf, h = build_reference_model(cfg, sd)
        moe = nerf.layers["0"]
        moe.train()
        gate = moe.gates[0]
        gate.gate_noise = gate_noise
        gate.top_k = top_k
        gate.use_load_importance_loss = True
        gate.compute_balance_loss = True
        E = cfg["num_experts"]
        rng = np.random.default_rng(seed + 1000)
        x = rng.standard_normal((P, cfg["model_dim"])).astype(np.float32)
        gi = rng.standard_normal((P, cfg["gate_hidden"])).astype(np.float32)
        xt = torch.from_numpy(x).requires_grad_(True)
        gt = torch.from_numpy(gi).requires_grad_(True)
        torch.manual_seed(rng_seed)
        y = moe(xt, gate_input=gt)
        torch.manual_seed(rng_seed)
        noise = torch.randn(P, E)
        logits = gt.detach() @ gate.wg.weight.detach().float().t()
        top = torch.topk(torch.softmax(logits + gate_noise * noise / E, dim=1), top_k, dim=1).indices
        assert torch.equal(top, y.gate_extras["gates"]), "noise replay does not reproduce the reference's routing"
        bal = y.gate_extras["balance_loss"]
        dy = rng.standard_normal(y.shape).astype(np.float32)
        (y * torch.from_numpy(dy)).sum().backward(retain_graph=True)
        dx, dgi, dwg = xt.grad.clone(), gt.grad.clone(), gate.wg.weight.grad.clone()
        gl = torch.autograd.grad(y.l_aux, [gt, gate.wg.weight], retain_graph=True)
        gb = torch.autograd.grad(bal, [gt, gate.wg.weight])
        save(f"moe_layer_load_importance_{tag}", seed=seed, P=P, top_k=top_k, gate_noise=gate_noise, noise=noise.numpy(), y=y.detach().numpy(),
             l_aux=y.l_aux.detach().numpy(), balance_loss=bal.detach().numpy(), topk=y.gate_extras["gates"].numpy().astype(np.int32),
             dx=dx.numpy(), dgate_input=dgi.numpy(), dwg=dwg.numpy(), laux_dgate_input=gl[0].numpy(), laux_dwg=gl[1].numpy(),
             bal_dgate_input=gb[0].numpy(), bal_dwg=gb[1].numpy())


# ------------------------------------------------------------------------------------------ G4 model fwd
def gen_model_forward():
    print("[G4] NeRFMoE.forward building shapes, P=4096")
    cfg = synth.BUILDING
    for tag, gate_scale in (("unbalanced", 1.0), ("balanced", 0.02)):
        sd = synth.make_weights(41, cfg, gate_scale=gate_scale)
        nerf, h = build_reference_model(cfg, sd)
        rng = np.random.default_rng(42)
        P = 4096
        x = np.concatenate([rng.uniform(-1, 1, (P, 3)), rng.standard_normal((P, 3)), rng.integers(0, 10, (P, 1))], 1).astype(np.float32)
        noise = rng.standard_normal((P, 1)).astype(np.float32)
        nerf.eval()
        with torch.no_grad():
            r = nerf(torch.from_numpy(x), sigma_noise=torch.from_numpy(noise))
        save(f"model_fwd_{tag}", seed=41, gate_scale=gate_scale, x=x, sigma_noise=noise, outputs=r["outputs"].numpy(),
             moe_loss=r["extras"]["moe_loss"].numpy(), moe_gates=r["extras"]["moe_gates"][0].numpy().astype(np.int32))


# ------------------------------------------------------------------------------------------ G9 eval / nobatch path
def gen_model_forward_nobatch():
    """The reference's eval path (README.md:172-185): --moe_expert_type=seqexperts --expertmlp2seqexperts, set_no_batch:
    apply_on_expert_fn_nobatch, no capacity drop, rows packed per expert, in-tree kernels tutel_sparse_nobatch.py."""
    print("[G9] NeRFMoE.forward, nobatch/eval path (seqexperts, no token dropping), P=4096")
    cfg = synth.BUILDING
    sd = synth.make_weights(41, cfg, gate_scale=1.0)
    h = make_hparams(cfg, bpr=False)
    h.moe_expert_type = "seqexperts"
    torch.manual_seed(0)
    nerf = model_utils.get_nerf(h, cfg["appearance_count"])
    conv = model_utils.convert_to_seqexperts({("module." + k): torch.from_numpy(v.copy()) for k, v in sd.items()})
    conv = {k[len("module."):]: v for k, v in conv.items()}
    missing = set(nerf.state_dict().keys()) ^ set(conv.keys())
    assert not missing, sorted(missing)[:5]
    nerf.load_state_dict(conv)
    nerf.set_no_batch(True)
    nerf.eval()
    g = np.load(os.path.join(OUT, "model_fwd_unbalanced.npz"))
    with torch.no_grad():
        r = nerf(torch.from_numpy(g["x"]), sigma_noise=None)
    save("model_fwd_nobatch", seed=41, gate_scale=1.0, x=g["x"], outputs=r["outputs"].numpy(),
         moe_gates=r["extras"]["moe_gates"][0].numpy().astype(np.int32))


def gen_dispatch_nobatch():
    """The reference's no-batch dispatcher on its own call sites: extract_critical (tutel_fast_dispatch_nobatch.py:205-251) ->
    TutelMoeFastDispatcher.update / encode / decode (:98-160) with autograd through GatingEncoder / GatingDecoder (:16-96), i.e. the
    three kernels of tutel_sparse_nobatch.py:24-133 with expert_locations_begin (run through the CPU stub of the JIT kernels, whose
    row rule `begin[idx] + loc`, no capacity test, is the one of those CUDA strings)."""
    print("[G10] no-batch dispatcher: extract_critical + encode / decode + gradients, S=768, E=8, M=32")
    from switch_nerf.modules.tutel_moe_ext import tutel_fast_dispatch_nobatch as fdn
    rng = np.random.default_rng(61)
    S, E, M = 768, 8, 32
    logits = torch.from_numpy((rng.standard_normal((S, E)) * 1.5).astype(np.float32))
    gates = torch.softmax(logits, 1).requires_grad_(True)
    x = torch.from_numpy(rng.standard_normal((S, M)).astype(np.float32)).requires_grad_(True)
    # (position-order ranking only: with batch_prioritized_routing the reference's `expert_input_nums = locations1[-1, :] + 1`
    #  is not the per-expert count and its own `sample_num * top_k == dispatched_input_numel` assertion fires; the reference's
    #  evaluation recipe runs without it)
    for tag, bpr in (("plain", False),):
        crit, l_loss = fdn.extract_critical(gates, 1, 1.0, True, bpr)
        n_exp, indices_s, locations_s, gates_s, expert_input_nums, capacity = crit
        fdr = fdn.fast_dispatcher(num_global_experts=E, capacity=capacity, model_dim=M, dispatch_dtype=torch.float32)
        fdr.update(indices_s, locations_s, gates_s, expert_input_nums, capacity=capacity, is_postscore=True, dispatcher_no_score=False)
        disp = fdr.encode(x)                                     # [S, M] rows packed per expert
        w = torch.from_numpy(rng.standard_normal((M, M)).astype(np.float32) / 8)
        eo = torch.tanh(disp @ w)                                # a stand-in expert (row-wise, so the packing order matters)
        eo.retain_grad()
        y = fdr.decode(eo)
        dy = torch.from_numpy(rng.standard_normal((S, M)).astype(np.float32))
        gx, gg = torch.autograd.grad((y * dy).sum(), [x, gates], retain_graph=True)
        begin = (torch.cumsum(expert_input_nums, 0) - expert_input_nums).to(torch.int32)
        d_eo, = torch.autograd.grad((y * dy).sum(), [eo], retain_graph=True)
        save(f"dispatch_nobatch_{tag}", bpr=int(bpr), gates=gates.detach().numpy(), x=x.detach().numpy(), w=w.numpy(), dy=dy.numpy(),
             indices=indices_s[0].numpy().astype(np.int32), locations=locations_s[0].numpy().astype(np.int32),
             expert_input_nums=expert_input_nums.numpy().astype(np.int32), expert_locations_begin=begin.numpy(), capacity=int(capacity),
             dispatched=disp.detach().numpy(), y=y.detach().numpy(), dx=gx.numpy(), dgates=gg.numpy(), d_expert_out=d_eo.numpy(),
             l_aux=l_loss.detach().numpy())


# ------------------------------------------------------------------------------------------ G5 render + train step
def gen_render(variants=(("unbalanced", 1.0, 1.0, True), ("balanced", 0.02, 1.0, True))):
    print("[G5] render_rays / training step (64 rays x 64 samples, chunk 1024 -> 4 chunks), fwd + grads")
    cfg = synth.BUILDING
    for tag, gate_scale, cf, bpr in variants:
        sd = synth.make_weights(51, cfg, gate_scale=gate_scale)
        N, S, chunk = 64, 64, 1024
        nerf, h = build_reference_model(cfg, sd, coarse=S, chunk=chunk, perturb=0.0, sigma_noise=False, capacity_factor=cf, bpr=bpr)
        rays, img, rgbs = synth.make_rays(52, N)
        nerf.train()
        t0 = time.time()
        res, _ = rendering.render_rays(nerf, None, torch.from_numpy(rays), torch.from_numpy(img), h, None, None,
                                       get_depth=True, get_depth_variance=True, get_bg_fg_rgb=False)
        photo = torch.nn.functional.mse_loss(res["rgb_coarse"], torch.from_numpy(rgbs))
        gate_loss = res["gate_loss_coarse"].mean()
        loss = photo + 5e-4 * gate_loss
        loss.backward()
        print(f"    reference fwd+bwd {time.time()-t0:.2f}s")
        out = dict(seed=51, gate_scale=gate_scale, capacity_factor=cf, bpr=int(bpr), N=N, S=S, chunk=chunk, rgb=res["rgb_coarse"].detach().numpy(),
                   depth=res["depth_coarse"].numpy(), depth_variance=res["depth_variance_coarse"].numpy(),
                   sigma=res["sigma_coarse"].detach().numpy(), gate_loss=res["gate_loss_coarse"].detach().numpy(),
                   moe_gates=res["moe_gates_coarse"].numpy().astype(np.int32).reshape(N, S),
                   loss=loss.detach().numpy(), photo=photo.detach().numpy())
        for n, p in nerf.named_parameters():
            g_ = p.grad
            out["gsum__" + n] = synth.checksum(g_.numpy())
            out["gslice__" + n] = g_.numpy().reshape(-1)[:: max(1, g_.numel() // 499)][:499]
        save(f"render_train_{tag}", **out)


class _CpuAutocastShim:
    """Stands in for torch.cuda.amp.autocast while the reference runs on the CPU: the reference opens
    `torch.cuda.amp.autocast(enabled=..., dtype=...)` regions by name (runner.py:598; the fp32 islands of the router,
    tutel_moe_layer_nobatch.py:109, and of the sigma head, nerf_moe.py:399), which only toggle the CUDA autocast state - a no-op on
    a CPU-only build.  Mapped onto torch.autocast("cpu") the same regions toggle the state that actually governs CPU tensors, so the
    reference's own code decides where the 16-bit type is used.  (Generator-side harness; nothing of the reference is modified.)"""

    def __init__(self, enabled=True, dtype=torch.bfloat16, cache_enabled=True):
        self.ctx = torch.autocast("cpu", dtype=dtype if dtype is not None else torch.bfloat16, enabled=enabled)

    def __enter__(self):
        return self.ctx.__enter__()

    def __exit__(self, *a):
        return self.ctx.__exit__(*a)


def gen_render_autocast():
    """[G5b] the training step of G5 under bf16 autocast (the reference's amp_use_bfloat16 recipes, runner.py:593-598), run on
    the CPU backend's autocast: pins oracle.Autocast(policy="cpu") - i.e. WHERE the reference's code uses the 16-bit type - on the
    reference itself.  (The CUDA backend's operator table differs for layer_norm and softplus only; see oracle.Autocast.)"""
    print("[G5b] render_rays / training step under torch.autocast('cpu', bfloat16)")
    cfg = synth.BUILDING
    real = torch.cuda.amp.autocast
    torch.cuda.amp.autocast = _CpuAutocastShim
    try:
        for tag, gate_scale in (("unbalanced", 1.0), ("balanced", 0.02)):
            sd = synth.make_weights(51, cfg, gate_scale=gate_scale)
            N, S, chunk = 64, 64, 1024
            nerf, h = build_reference_model(cfg, sd, coarse=S, chunk=chunk, perturb=0.0, sigma_noise=True)
            h.amp_use_bfloat16 = True
            nerf.args.amp_use_bfloat16 = True
            rays, img, rgbs = synth.make_rays(52, N)
            nerf.train()
            # sigma noise on (rendering.py:366 draws randn per chunk): replay the draws from the seed and hand them to the oracle
            torch.manual_seed(1234)
            noise = torch.cat([torch.randn(chunk, 1) for _ in range(N * S // chunk)], 0)
            torch.manual_seed(1234)
            with torch.cuda.amp.autocast(enabled=True, dtype=torch.bfloat16):               # runner.py:598
                res, _ = rendering.render_rays(nerf, None, torch.from_numpy(rays), torch.from_numpy(img), h, None, None,
                                               get_depth=True, get_depth_variance=True, get_bg_fg_rgb=False)
                photo = torch.nn.functional.mse_loss(res["rgb_coarse"], torch.from_numpy(rgbs))
                gate_loss = res["gate_loss_coarse"].mean()
                loss = photo + 5e-4 * gate_loss
            loss.backward()
            out = dict(seed=51, gate_scale=gate_scale, N=N, S=S, chunk=chunk, sigma_noise=noise.numpy(), rgb=res["rgb_coarse"].detach().float().numpy(),
                       sigma=res["sigma_coarse"].detach().float().numpy(), sigma_is_bf16=int(res["sigma_coarse"].dtype == torch.bfloat16),
                       gate_loss=res["gate_loss_coarse"].detach().float().numpy(),
                       moe_gates=res["moe_gates_coarse"].numpy().astype(np.int32).reshape(N, S),
                       loss=loss.detach().float().numpy(), photo=photo.detach().float().numpy())
            for n, p_ in nerf.named_parameters():
                g_ = p_.grad.float()
                out["gsum__" + n] = synth.checksum(g_.numpy())
                out["gslice__" + n] = g_.numpy().reshape(-1)[:: max(1, g_.numel() // 499)][:499]
            save(f"render_train_bf16cpu_{tag}", **out)
    finally:
        torch.cuda.amp.autocast = real


def gen_render_capacity():
    """Other capacity factors / ranking modes (BASELINE configs[4]: capacity_factor 1.25 with token dropping; no BPR =
    position-order ranking, tutel_fast_dispatch.py:177-191)."""
    gen_render((("cf125_nobpr", 1.0, 1.25, False), ("cf125_bpr", 1.0, 1.25, True), ("cf050_bpr", 0.02, 0.5, True)))


def gen_render_fine():
    print("[G5f] hierarchical render_rays / training step (64 rays x (64 coarse + 96 fine), chunk 1024), fwd + grads")
    cfg = synth.BUILDING
    for tag, perturb in (("det", 0.0), ("perturbed", 1.0)):
        sd = synth.make_weights(61, cfg, gate_scale=0.02)
        N, S, Fn, chunk = 64, 64, 96, 1024
        nerf, h = build_reference_model(cfg, sd, coarse=S, chunk=chunk, perturb=perturb, sigma_noise=False, fine=Fn)
        rays, img, rgbs = synth.make_rays(62, N)
        nerf.train()
        # the reference draws rand_like(z_vals) (rendering.py:583) then torch.rand(N, fine) (:608): replay the stream
        torch.manual_seed(77)
        pr = torch.rand(N, S)
        fu = torch.rand(N, Fn)
        torch.manual_seed(77)
        res, _ = rendering.render_rays(nerf, None, torch.from_numpy(rays), torch.from_numpy(img), h, None, None,
                                       get_depth=True, get_depth_variance=True, get_bg_fg_rgb=False)
        assert "rgb_coarse" not in res
        photo = torch.nn.functional.mse_loss(res["rgb_fine"], torch.from_numpy(rgbs))
        gate_loss = (res["gate_loss_fine"].mean() + res["gate_loss_coarse"].mean()) / 2.0       # runner.py:1104-1111
        loss = photo + 5e-4 * gate_loss
        loss.backward()
        out = dict(seed=61, gate_scale=0.02, N=N, S=S, F=Fn, chunk=chunk, perturb=perturb, rgb=res["rgb_fine"].detach().numpy(),
                   depth=res["depth_fine"].numpy(), depth_variance=res["depth_variance_fine"].numpy(),
                   sigma_coarse=res["sigma_coarse"].detach().numpy(), sigma_fine=res["sigma_fine"].detach().numpy(),
                   gate_loss_coarse=res["gate_loss_coarse"].detach().numpy(),
                   gate_loss_fine=res["gate_loss_fine"].detach().numpy(),
                   moe_gates_coarse=res["moe_gates_coarse"].numpy().astype(np.int32).reshape(N, S),
                   moe_gates_fine=res["moe_gates_fine"].numpy().astype(np.int32).reshape(N, Fn),
                   loss=loss.detach().numpy(), photo=photo.detach().numpy())
        if perturb > 0:
            out["perturb_rand"], out["fine_u"] = pr.numpy(), fu.numpy()
        for n, p in nerf.named_parameters():
            g_ = p.grad
            out["gsum__" + n] = synth.checksum(g_.numpy())
            out["gslice__" + n] = g_.numpy().reshape(-1)[:: max(1, g_.numel() // 499)][:499]
        save(f"render_train_fine_{tag}", **out)


def gen_mip():
    print("[G7] mip path: mip_cast_rays + MipEmbedder + MipNeRFMoE + resampling, 2 levels, loss = (fine + coarse) / 2, fwd + grads")
    cfg = synth.BUILDING
    for tag, perturb in (("det", 0.0), ("perturbed", 1.0)):
        sd = synth.make_weights(71, cfg, gate_scale=0.02)
        N, S, Fn, chunk = 64, 65, 65, 1024          # 64 intervals per level -> 4096 points = 4 chunks
        nerf, h = build_reference_model(cfg, sd, coarse=S, chunk=chunk, perturb=perturb, sigma_noise=False, fine=Fn, mip=True)
        rays, img, rgbs = synth.make_rays(72, N)
        radii = (np.random.default_rng(73).uniform(0.5, 2.0, (N, 1)) * 1e-3).astype(np.float32)
        nerf.train()
        # random draws of the reference in call order: rand_like(z_vals) (rendering_mip.py, _expand_and_perturb_z_vals),
        # then torch.rand([N, fine]) inside sorted_piecewise_constant_pdf1 - replayed from the seed
        torch.manual_seed(79)
        pr = torch.rand(N, S)
        fu = torch.rand(N, Fn)
        torch.manual_seed(79)
        res, _ = rendering_mip.render_rays(nerf, torch.from_numpy(rays), torch.from_numpy(radii), torch.from_numpy(img), h,
                                           get_depth=True, get_depth_variance=True)
        t = torch.from_numpy(rgbs)
        photo = (torch.nn.functional.mse_loss(res["rgb_fine"], t) + torch.nn.functional.mse_loss(res["rgb_coarse"], t)) / 2
        gate_loss = (res["gate_loss_fine"].mean() + res["gate_loss_coarse"].mean()) / 2.0       # runner.py:1157-1163
        loss = photo + 5e-4 * gate_loss
        loss.backward()
        out = dict(seed=71, gate_scale=0.02, N=N, S=S, F=Fn, chunk=chunk, perturb=perturb, radii=radii,
                   rgb_coarse=res["rgb_coarse"].detach().numpy(), rgb_fine=res["rgb_fine"].detach().numpy(),
                   depth=res["depth_fine"].numpy(), depth_variance=res["depth_variance_fine"].numpy(),
                   gate_loss_coarse=res["gate_loss_coarse"].detach().numpy(), gate_loss_fine=res["gate_loss_fine"].detach().numpy(),
                   moe_gates_coarse=res["moe_gates_coarse"].numpy().astype(np.int32).reshape(N, S - 1),
                   moe_gates_fine=res["moe_gates_fine"].numpy().astype(np.int32).reshape(N, Fn - 1),
                   loss=loss.detach().numpy(), photo=photo.detach().numpy())
        if perturb > 0:
            out["perturb_rand"], out["fine_u"] = pr.numpy(), fu.numpy()
        for n, p in nerf.named_parameters():
            g_ = p.grad
            out["gsum__" + n] = synth.checksum(g_.numpy())
            out["gslice__" + n] = g_.numpy().reshape(-1)[:: max(1, g_.numel() // 499)][:499]
        save(f"mip_train_{tag}", **out)
    # kernel-level vectors: cast + integrated positional encoding, resampling
    rng = np.random.default_rng(75)
    N, S = 32, 33
    rays, _, _ = synth.make_rays(76, N)
    radii = (rng.uniform(0.5, 2.0, (N, 1)) * 1e-3).astype(np.float32)
    z = torch.from_numpy(rays[:, 6:7] * (1 - np.linspace(0, 1, S, dtype=np.float32)) + rays[:, 7:8] * np.linspace(0, 1, S, dtype=np.float32))
    mean, cov = rendering_mip.mip_cast_rays(torch.from_numpy(rays[:, 0:3]), torch.from_numpy(rays[:, 3:6]), torch.from_numpy(radii), z)
    from switch_nerf.models.nerf import MipEmbedder
    ipe = MipEmbedder(12)(torch.cat([mean, cov], -1).view(-1, 6))
    w = torch.from_numpy((rng.uniform(0, 1, (N, S - 1)) ** 3).astype(np.float32))
    w[5] = 0
    wp = torch.cat([w[..., :1], w, w[..., -1:]], -1)
    wm = torch.maximum(wp[..., :-1], wp[..., 1:])
    blur = 0.5 * (wm[..., :-1] + wm[..., 1:]) + 0.01
    zs = rendering_mip.sorted_piecewise_constant_pdf1(z, blur.clone(), 40, randomized=False)
    save("mip_kernels", rays=rays, radii=radii, z=z.numpy(), mean=mean.numpy(), cov=cov.numpy(), ipe=ipe.numpy(), weights=w.numpy(),
         z_resampled_det=zs.numpy())


def gen_checkpoint_layout():
    print("[G9] checkpoint layouts: expertmlp state_dict -> the reference's convert_to_seqexperts (model_utils.py:12-28)")
    cfg = synth.BUILDING
    sd = {("module." + k): torch.from_numpy(v.copy()) for k, v in synth.make_weights(91, cfg).items()}   # DDP-prefixed, as saved
    conv = model_utils.convert_to_seqexperts(dict(sd))
    keys = sorted(conv.keys())
    save("checkpoint_seqexperts", keys=np.array(keys), sums=np.stack([synth.checksum(conv[k].numpy()) for k in keys]),
         shapes=np.array([str(tuple(conv[k].shape)) for k in keys]))


def gen_dense():
    print("[G6] BASELINE configs[0]: dense NeRF (use_moe off), 1024 rays x 64 samples, fwd + grads of mse")
    cfg = synth.DENSE
    sd = synth.make_dense_weights(161, cfg)
    h = make_hparams(synth.BUILDING, coarse=64, chunk=65536, perturb=0.0)
    h.use_moe = False
    h.layers, h.skip_layers, h.layer_dim = cfg["layers"], list(cfg["skip_layers"]), cfg["layer_dim"]
    torch.manual_seed(0)
    nerf = model_utils.get_nerf(h, cfg["appearance_count"])
    ref_sd = nerf.state_dict()
    assert set(ref_sd.keys()) == set(sd.keys()), sorted(set(ref_sd) ^ set(sd))
    nerf.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()})
    N, S = 1024, 64
    rays, img, rgbs = synth.make_rays(162, N)
    nerf.train()
    res, _ = rendering.render_rays(nerf, None, torch.from_numpy(rays), torch.from_numpy(img), h, None, None,
                                   get_depth=True, get_depth_variance=True, get_bg_fg_rgb=False)
    loss = torch.nn.functional.mse_loss(res["rgb_coarse"], torch.from_numpy(rgbs))
    loss.backward()
    out = dict(seed=161, N=N, S=S, rgb=res["rgb_coarse"].detach().numpy(), depth=res["depth_coarse"].numpy(),
               sigma_head=res["sigma_coarse"].detach().numpy()[:64], loss=loss.detach().numpy())
    for n, p in nerf.named_parameters():
        g_ = p.grad
        out["gsum__" + n] = synth.checksum(g_.numpy())
        out["gslice__" + n] = g_.numpy().reshape(-1)[:: max(1, g_.numel() // 499)][:499]
    save("dense_nerf_train", **out)


# ------------------------------------------------------------------------------------------ G5b compositing + sample_pdf
def gen_composite():
    print("[G5b] compositing / _sample_pdf on raw tensors")
    rng = np.random.default_rng(61)
    N, S = 32, 256
    rays, _, _ = synth.make_rays(62, N)
    h = make_hparams(synth.BUILDING, coarse=S)
    rgbs = rng.uniform(0, 1, (N, S, 3)).astype(np.float32)
    sig = np.abs(rng.standard_normal((N, S))).astype(np.float32) * 20

    class Fake(torch.nn.Module):
        def forward(self, x, sigma_noise=None):
            return torch.cat([self.rgb, self.sig], -1)
    f = Fake()
    f.rgb = torch.from_numpy(rgbs.reshape(-1, 3))
    f.sig = torch.from_numpy(sig.reshape(-1, 1))
    h.use_moe = False
    h.return_sigma = False
    h.model_chunk_size = N * S
    res, _ = rendering.render_rays(f.eval(), None, torch.from_numpy(rays), torch.zeros(N, dtype=torch.long), h, None, None,
                                   get_depth=True, get_depth_variance=True, get_bg_fg_rgb=False)
    zs = torch.linspace(0, 1, S)
    z = torch.from_numpy(rays[:, 6:7]) * (1 - zs) + torch.from_numpy(rays[:, 7:8]) * zs
    # weights via the get_weights branch
    r2 = {}
    rendering._inference(r2, "coarse", f, torch.from_numpy(rays[:, None, 3:6]), None, h,
                         torch.zeros(N, S, 3), z, 1e10 * torch.ones(N, 1), True, True, True, True, False, False, None)
    zmid = 0.5 * (z[:, :-1] + z[:, 1:])
    fine_det = rendering._sample_pdf(zmid, r2["weights_coarse"][:, 1:-1], 64, det=True)
    save("composite", rays=rays, rgbs=rgbs, sigmas=sig, rgb=res["rgb_coarse"].numpy(), depth=res["depth_coarse"].numpy(),
         depth_variance=res["depth_variance_coarse"].numpy(), weights=r2["weights_coarse"].numpy(), z=z.numpy(),
         fine_det=fine_det.numpy())


def gen_bg():
    print("[G8] background model + ellipsoid bound: NeRFMoE foreground + dense 4-D background NeRF (rendering.py:32-159), fwd + grads")
    cfg, cfg_bg = synth.BUILDING, synth.DENSE_BG
    center, radius = torch.from_numpy(synth.SPHERE_CENTER), torch.from_numpy(synth.SPHERE_RADIUS)
    for tag, perturb, Fn in (("coarse_det", 0.0, 0), ("coarse", 1.0, 0), ("fine", 1.0, 64)):
        sd = synth.make_weights(81, cfg, gate_scale=0.02)
        sd_bg = synth.make_dense_weights(82, cfg_bg)
        N, S, chunk = 96, 64, 1024
        nerf, h = build_reference_model(cfg, sd, coarse=S, chunk=chunk, perturb=perturb, sigma_noise=False, fine=Fn)
        h.layers, h.skip_layers, h.bg_layer_dim = cfg_bg["layers"], list(cfg_bg["skip_layers"]), cfg_bg["layer_dim"]
        bg = model_utils.get_bg_nerf(h, cfg_bg["appearance_count"])
        assert set(bg.state_dict().keys()) == set(sd_bg.keys())
        bg.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd_bg.items()})
        rays, img, rgbs = synth.make_bg_rays(83, N)
        nerf.train()
        bg.train()
        r = torch.from_numpy(rays)
        fg_far = torch.maximum(rendering._intersect_sphere(r[:, :3], r[:, 3:6], center, radius), r[:, 6])
        Nb = int((r[:, 7] > fg_far).sum())
        assert 0 < Nb < N, Nb
        # the reference's random draws in call order (background first): rand_like(bg z) :583, rand(Nb, fine // 2) :608,
        # rand_like(z) :583, rand(N, fine) :608 - replayed from the seed
        torch.manual_seed(87)
        draws = dict(perturb_rand_bg=torch.rand(Nb, S // 2))
        if Fn:
            draws["fine_u_bg"] = torch.rand(Nb, Fn // 2)
        draws["perturb_rand"] = torch.rand(N, S)
        if Fn:
            draws["fine_u"] = torch.rand(N, Fn)
        torch.manual_seed(87)
        res, present = rendering.render_rays(nerf, bg, r, torch.from_numpy(img), h, center, radius,
                                             get_depth=True, get_depth_variance=True, get_bg_fg_rgb=True)
        assert present
        typ = "fine" if Fn else "coarse"
        photo = torch.nn.functional.mse_loss(res[f"rgb_{typ}"], torch.from_numpy(rgbs))
        gate_loss = res["gate_loss_coarse"].mean()
        if Fn:
            gate_loss = (res["gate_loss_fine"].mean() + gate_loss) / 2.0
        loss = photo + 5e-4 * gate_loss
        loss.backward()
        out = dict(seed=81, seed_bg=82, gate_scale=0.02, N=N, S=S, F=Fn, chunk=chunk, perturb=perturb, n_bg=Nb,
                   rgb=res[f"rgb_{typ}"].detach().numpy(), depth=res[f"depth_{typ}"].detach().numpy(),
                   depth_variance=res[f"depth_variance_{typ}"].numpy(), fg_rgb=res[f"fg_rgb_{typ}"].detach().numpy(),
                   bg_rgb=res[f"bg_rgb_{typ}"].detach().numpy(), fg_far=fg_far.numpy(),
                   loss=loss.detach().numpy(), photo=photo.detach().numpy())
        if perturb > 0:
            out.update({k: v.numpy() for k, v in draws.items()})
        for pre, mod in (("", nerf), ("bg__", bg)):
            for n, p in mod.named_parameters():
                g_ = p.grad
                out["gsum__" + pre + n] = synth.checksum(g_.numpy())
                out["gslice__" + pre + n] = g_.numpy().reshape(-1)[:: max(1, g_.numel() // 499)][:499]
        save(f"bg_train_{tag}", **out)
    # _depth2pts_outside / _intersect_sphere on their own (kernel-level fixture)
    rays, _, _ = synth.make_bg_rays(84, 40)
    r = torch.from_numpy(rays)
    depth = torch.from_numpy(np.random.default_rng(85).uniform(1e-3, 1.0, (40, 24)).astype(np.float32))
    pts, dreal = rendering._depth2pts_outside(r[:, None, :3], r[:, None, 3:6], depth, center, radius, False, False)
    save("bg_points", depth=depth.numpy(), pts=pts.numpy(), depth_real=dreal.numpy(),
         fg_far=rendering._intersect_sphere(r[:, :3], r[:, 3:6], center, radius).numpy())


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    todo = dict(routing=gen_routing, pe=gen_pe, moe=gen_moe_layer, moe_noise=gen_moe_layer_noise, moe_normal_noise=gen_moe_layer_normal_noise, moe_top2=gen_moe_layer_top2, moe_li=gen_moe_layer_load_importance, model=gen_model_forward, nobatch=gen_model_forward_nobatch, dispatch_nobatch=gen_dispatch_nobatch,
                render=gen_render, autocast=gen_render_autocast, capacity=gen_render_capacity, fine=gen_render_fine, mip=gen_mip, ckpt=gen_checkpoint_layout, dense=gen_dense, composite=gen_composite, bg=gen_bg)
    for k, fn in todo.items():
        if a.only and k not in a.only.split(","):
            continue
        fn()
