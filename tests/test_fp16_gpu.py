"""IEEE-half compute (BASELINE configs[4]: "fp16 MFMA"): the second build of the library (libswn_hip_f16.so, v_mfma_f32_32x32x16_f16
in the chain and weight-gradient kernels) with the reference's loss-scaling contract (torch GradScaler, runner.py:483, 679)."""
import numpy as np
import pytest
import torch

import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _restore_bf16_build():
    from switch_nerf_amd import _lib
    yield
    _lib.use_half("bf16")


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _model(dtype, seed, gate_scale=0.05, **kw):
    from switch_nerf_amd.model import SwitchNeRF
    m = SwitchNeRF(synth.BUILDING, dtype=dtype, **kw)
    m.load_state_dict(synth.make_weights(seed, synth.BUILDING, gate_scale=gate_scale))
    return m


def test_fp16_step_close_to_fp32_and_loss_scaling():
    """fp16 forward / backward against the fp32 run of the same batch (capacity 512 per group: the expert chains take the 256-row
    geometry, fp16 MFMA): same routing up to near-ties, rgb and loss at half-precision tolerance, gradients (unscaled) close;
    then Adam steps under the loss scaler lower the loss, and an overflowing scale skips the step and backs off like GradScaler."""
    N, S, chunk = 128, 128, 4096
    rays, img, rgbs = synth.make_rays(502, N)
    m32 = _model(torch.float32, 501)
    a = m32.train_step(_dev(rgbs), _dev(rays), _dev(img), S, chunk, perturb=0.0, optimizer_step=False)
    g32 = m32.grad.clone()
    idx32, rgb32 = a["ctx"]["idx"].clone(), a["ctx"]["rgb"].clone()
    m16 = _model(torch.float16, 501)
    from switch_nerf_amd import _lib
    assert _lib.half_kind() == "f16" and m16.loss_scaler is not None and m16.loss_scaler.scale == 65536.0
    b = m16.train_step(_dev(rgbs), _dev(rays), _dev(img), S, chunk, perturb=0.0, optimizer_step=False)
    assert b["ctx"]["geom"] == 7 and b["ctx"]["h0"].dtype == torch.float16
    assert (b["ctx"]["idx"] != idx32).float().mean().item() < 5e-3
    assert (b["ctx"]["rgb"] - rgb32).abs().max().item() < 5e-3            # fp16 has 3 more mantissa bits than bf16
    assert abs(a["loss"].item() - b["loss"].item()) < 5e-3 * abs(a["loss"].item())
    g16 = m16.grad / m16.loss_scaler.scale                                  # the backward ran on the scaled loss
    assert torch.isfinite(g16).all()
    rel = (g16 - g32).abs().max().item() / g32.abs().max().item()
    print(f"fp16 vs fp32: max |rgb diff| {(b['ctx']['rgb'] - rgb32).abs().max().item():.2e}, relative gradient difference {rel:.2e}")
    assert rel < 3e-2
    l0 = None
    for it in range(8):
        st = m16.train_step(_dev(rgbs), _dev(rays), _dev(img), S, chunk, perturb=0.0)
        l0 = st["loss"].item() if l0 is None else l0
    assert st["loss"].item() < l0 and m16.step_count == 8 and m16.loss_scaler.skipped == 0
    # overflow: a scale beyond the fp16 range makes the scaled gradients non-finite -> the step is skipped, the scale halves
    before = m16.flat.clone()
    m16.loss_scaler.scale = 2.0 ** 40
    m16.train_step(_dev(rgbs), _dev(rays), _dev(img), S, chunk, perturb=0.0)
    assert m16.step_count == 8 and m16.loss_scaler.skipped == 1 and m16.loss_scaler.scale == 2.0 ** 39 and torch.equal(before, m16.flat)


@pytest.mark.parametrize("geometry", [1, 2])
def test_fp16_expert_chain_and_wgrad_vs_fp32_math(geometry):
    """The fp16 instantiations of both chain geometries and of the weight-gradient GEMM against fp32 torch math on fp16-rounded
    operands."""
    from switch_nerf_amd import _lib, ops as o
    _lib.use_half("f16")
    dt = torch.float16
    g = torch.Generator().manual_seed(7)
    E, M, L, cap, ng = 8, 256, 7, 512, 8
    rows = ng * cap
    counts = torch.tensor([512, 0, 300, 512, 17, 512, 256, 257], dtype=torch.int32)
    x = torch.randn(rows, M, generator=g)
    W = [torch.randn(E, M, M, generator=g) / 16 for _ in range(L)]
    B = [torch.randn(E, M, generator=g) * 0.1 for _ in range(L)]
    r16 = lambda t: t.to(dt).float()
    wf = [o.pack_weights(w.cuda(), dt, True) for w in W]
    saves = [torch.zeros(rows, M, dtype=dt, device="cuda") for _ in range(L - 1)]
    masks = [torch.zeros(o.chain_mask_words(dt, ng, cap, M), dtype=torch.int32, device="cuda") for _ in range(L - 1)]
    y = torch.zeros(rows, M, dtype=dt, device="cuda")
    layers = [o.Layer(wf[l], B[l].cuda(), relu=1 if l < L - 1 else 0, skip=(l == 3), save=saves[l] if l < L - 1 else None,
                      mask=masks[l] if l < L - 1 else None) for l in range(L)]
    o.mlp_chain(x.cuda().to(dt), layers, y, n_groups=ng, n_wsets=E, group_stride=cap, group_rows=counts.cuda(), group_rows_clamp=cap,
                tag=1, geometry=geometry)
    worst = 0.0
    for gi in range(ng):
        c = int(counts[gi])
        if not c:
            continue
        h = x0 = r16(x[gi * cap: gi * cap + c])
        for l in range(L):
            h = h @ r16(W[l][gi % E]) + B[l][gi % E]
            if l == 3:
                h = h + x0
            if l < L - 1:
                h = torch.relu(h)
            h = r16(h)
            if l == 2:
                worst = max(worst, (saves[2][gi * cap: gi * cap + c].float().cpu() - h).abs().max().item())
        worst = max(worst, (y[gi * cap: gi * cap + c].float().cpu() - h).abs().max().item())
    assert worst < 2e-2, worst
    # weight gradient: dW = A^T B per group, fp16 operands, fp32 accumulation
    a16, b16 = saves[0], saves[1]
    dw = torch.zeros(E, M, M, device="cuda")
    db = torch.zeros(E, M, device="cuda")
    o.wgrad(a16, b16, dw, db, n_groups=ng, n_wsets=E, group_stride=cap, group_rows=counts.cuda(), group_rows_clamp=cap, n_splits=2)
    ref = torch.zeros(E, M, M)
    for gi in range(ng):
        c = int(counts[gi])
        ref[gi % E] += a16[gi * cap: gi * cap + c].float().cpu().t() @ b16[gi * cap: gi * cap + c].float().cpu()
    assert (dw.cpu() - ref).abs().max().item() <= 2e-3 * ref.abs().max().item()


def test_graphed_fp16_step_follows_the_loss_scale():
    """graph.GraphedTrainStep on a float16 model.  The captured step reads the loss scale from a device scalar, the
    step after the replay goes through SwitchNeRF.apply_step (inf check, unscale, skipped step, scale update): three replays equal three
    eager steps of a twin model; a scale beyond the fp16 range is detected (step skipped, parameters untouched, scale halved) and the
    NEXT replay already runs with the halved scale."""
    from switch_nerf_amd import _lib
    from switch_nerf_amd.graph import GraphedTrainStep
    N, S, chunk = 256, 64, 4096
    rays, img, rgbs = synth.make_rays(522, N)
    try:
        ma, mb = _model(torch.float16, 521), _model(torch.float16, 521)
        step = GraphedTrainStep(ma, _dev(rgbs), _dev(rays), _dev(img), S, chunk, perturb=0.0, noise_std=0.0)
        ma.load_state_dict(synth.make_weights(521, synth.BUILDING, gate_scale=0.05))
        ma.m.zero_(); ma.v.zero_(); ma.step_count = 0
        ma.refresh_compute_copies()
        for _ in range(3):
            ra = step()
            rb = mb.train_step(_dev(rgbs), _dev(rays), _dev(img), S, chunk, perturb=0.0)
            assert abs(ra["loss"].item() - rb["loss"].item()) <= 1e-3 * abs(rb["loss"].item())
        assert ma.step_count == mb.step_count == 3 and ma.loss_scaler.skipped == 0
        d = (ma.flat - mb.flat).abs().max().item() / mb.flat.abs().max().item()
        assert d <= 5e-3, d                       # Adam actually consumed UNSCALED gradients (65536 x would move the weights by lr each)
        before = ma.flat.clone()
        ma.loss_scaler.scale = 2.0 ** 40
        ma._loss_scale_tensor()                   # (a caller that sets the scale by hand syncs the device copy; _unscale_ok does it itself)
        step()
        assert ma.step_count == 3 and ma.loss_scaler.skipped == 1 and ma.loss_scaler.scale == 2.0 ** 39 and torch.equal(before, ma.flat)
        assert float(ma._ls_dev.item()) == 2.0 ** 39
    finally:
        _lib.use_half("bf16")
