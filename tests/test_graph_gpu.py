"""hipGraph replay of the training step (switch_nerf_amd/graph.py, the default timed region of bench.py) against the eager step."""
import numpy as np
import pytest
import torch

import synth

pytestmark = pytest.mark.gpu


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _model(dtype, seed):
    from switch_nerf_amd.model import SwitchNeRF
    m = SwitchNeRF(synth.BUILDING, dtype=dtype)
    m.load_state_dict(synth.make_weights(seed, synth.BUILDING))
    return m


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_graphed_train_step_equals_eager(dtype):
    """Three optimizer steps replayed from the captured graph (deterministic sampling: no jitter, no sigma noise) equal three eager
    train_step calls on an identical model: same routing, same losses, same parameters (to the order of the atomically accumulated
    weight gradients).  New ray batches are copied into the graph's static inputs."""
    from switch_nerf_amd.graph import GraphedTrainStep
    N, S, chunk = 512, 64, 8192
    batches = [synth.make_rays(300 + i, N) for i in range(3)]
    ma, mb = _model(dtype, 41), _model(dtype, 41)
    rays0, img0, rgbs0 = batches[0]
    step = GraphedTrainStep(ma, _dev(rgbs0), _dev(rays0), _dev(img0), S, chunk, perturb=0.0, noise_std=0.0)
    ma.load_state_dict(synth.make_weights(41, synth.BUILDING))          # (the capture's warm-up steps did not touch the parameters,
    ma.m.zero_(); ma.v.zero_(); ma.step_count = 0                      #  but be explicit) - same start as the eager model
    ma.refresh_compute_copies()
    for rays, img, rgbs in batches:
        ra = step(_dev(rgbs), _dev(rays), _dev(img))
        la, idx_a = float(ra["loss"].item()), ra["ctx"]["idx"].clone()
        rb = mb.train_step(_dev(rgbs), _dev(rays), _dev(img), S, chunk, perturb=0.0)
        lb = float(rb["loss"].item())
        assert torch.equal(idx_a, rb["ctx"]["idx"]), "routing"
        assert abs(la - lb) <= 2e-5 * max(1.0, abs(lb)), (la, lb)
    tol = 2e-3 if dtype == torch.bfloat16 else 2e-5
    d = (ma.flat - mb.flat).abs().max().item() / mb.flat.abs().max().item()
    assert d <= tol, d
    assert ma.step_count == mb.step_count == 3
