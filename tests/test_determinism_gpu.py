"""The training step gives the same bits on every run: every parameter-gradient sum is added in a fixed order (weight gradients: slab
partials + ordered reduce, wgrad.hip; head / router / LayerNorm gradients: block partials + ordered_reduce_kernel; embedding gradient:
rays in ascending order, swn_emb_grad) - no floating-point atomics on the default path.  Two models started from the same weights stay
bit-identical over optimizer steps, and a step replayed from a hipGraph equals the eager step bit for bit."""
import numpy as np
import pytest
import torch

import synth

pytestmark = pytest.mark.gpu


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _model(dtype, seed, cfg=None):
    from switch_nerf_amd.model import SwitchNeRF
    cfg = cfg or synth.BUILDING
    m = SwitchNeRF(cfg, dtype=dtype)
    m.load_state_dict(synth.make_weights(seed, cfg))
    return m


@pytest.mark.parametrize("dtype,fine", [(torch.bfloat16, 0), (torch.float32, 0), (torch.bfloat16, 32)])
def test_twin_models_stay_bit_identical(dtype, fine):
    N, S, chunk = 1024, 64, 8192
    a, b = _model(dtype, 51), _model(dtype, 51)
    for it in range(4):
        rays, img, rgbs = synth.make_rays(700 + it, N)
        ra = a.train_step(_dev(rgbs), _dev(rays), _dev(img), S, chunk, perturb=0.0, fine_samples=fine)
        ga = a.grad.clone()
        rb = b.train_step(_dev(rgbs), _dev(rays), _dev(img), S, chunk, perturb=0.0, fine_samples=fine)
        assert torch.equal(ra["ctx"]["idx"], rb["ctx"]["idx"]), it
        diff = [(n, (ga[o_:o_ + int(np.prod(sh))] - b.grad[o_:o_ + int(np.prod(sh))]).abs().max().item())
                for n, (o_, sh) in a.spec.items()]
        assert torch.equal(ga, b.grad), (it, [d for d in diff if d[1] > 0])
        assert torch.equal(a.flat, b.flat), it
        assert ra["loss"].item() == rb["loss"].item()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_graph_replay_is_bit_identical_to_eager(dtype):
    from switch_nerf_amd.graph import GraphedTrainStep
    N, S, chunk = 512, 64, 8192
    batches = [synth.make_rays(720 + i, N) for i in range(5)]
    a, b = _model(dtype, 52), _model(dtype, 52)
    rays0, img0, rgbs0 = batches[0]
    step = GraphedTrainStep(a, _dev(rgbs0), _dev(rays0), _dev(img0), S, chunk, perturb=0.0, noise_std=0.0)
    a.load_state_dict(synth.make_weights(52, synth.BUILDING))
    a.m.zero_(); a.v.zero_(); a.step_count = 0
    a.refresh_compute_copies()
    for it, (rays, img, rgbs) in enumerate(batches):
        ra = step(_dev(rgbs), _dev(rays), _dev(img))
        la = ra["loss"].item()
        rb = b.train_step(_dev(rgbs), _dev(rays), _dev(img), S, chunk, perturb=0.0)
        assert torch.equal(ra["ctx"]["idx"], rb["ctx"]["idx"]), it
        assert la == rb["loss"].item(), it
        assert torch.equal(a.flat, b.flat), it
