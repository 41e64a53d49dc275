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
    train_step calls on an identical model: same routing, same losses, same parameters - bit for bit.  New ray batches are copied into
    the graph's static inputs."""
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
        assert la == lb, (la, lb)
    # every gradient sum is added in a fixed order (tests/test_determinism_gpu.py): the replayed step IS the eager step, bit for bit
    assert torch.equal(ma.flat, mb.flat), (ma.flat - mb.flat).abs().max().item()
    assert ma.step_count == mb.step_count == 3


def _probe(*args, env=None, timeout=400):
    """scripts/graph_probe.py in its own process (a GPU fault must not take the test session down) -> its result dict."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "scripts", "graph_probe.py")] + list(args), capture_output=True, text=True,
                       timeout=timeout, env=dict(os.environ, **(env or {})))
    assert p.returncode == 0, (p.returncode, p.stderr[-800:])
    line = [l for l in p.stdout.splitlines() if l.startswith("PROBE ")]
    assert line, p.stdout[-400:]
    return json.loads(line[-1][6:])


@pytest.mark.parametrize("variant", [("--rays", "2048"), ("--rays", "1024", "--no-batch"), ("--rays", "1024", "--fine", "128")])
def test_graphed_render_50_replays_equal_eager(variant):
    """The inference forward (matrix-pipe router kernel inside) replayed 50 times from one hipGraph on alternating ray batches: every
    checked replay equals the eager forward bit for bit (rgb and top-1 indices) - round 2's fault on the second replay of such a graph
    does not occur."""
    r = _probe("render", "--replays", "50", *variant)
    assert r["ok"] and r["routing_mismatches"] == 0 and r["rgb_max_diff"] == 0.0, r


def test_graphed_train_50_replays_equal_eager():
    r = _probe("train", "--replays", "50", "--rays", "1024")
    assert r["ok"] and r["routing_mismatches"] == 0 and r["param_abs_diff_end"] == 0.0, r


def test_expert_chain_200_back_to_back_launches_bit_exact():
    """200 full-size (2,097,152-row) launches of the phase-shifted expert chains: geometries 5 and 6 (the persistent form, in turn) are
    bit-identical to the 64-row kernels on every launch (outputs and all six saved activations) - the store-data hazard of round 2 (a
    later register value stored a few times per 1e5 stores) would show here -, geometry 7 (the model's default) is bit-identical to
    geometry 4 and within the bias-first accumulation bound."""
    r = _probe("chain", "--launches", "200", timeout=900)
    assert r["ok"] and r["geometry5_launches_with_a_difference"] == 0 and r["geometry7_launches_that_differ_from_geometry4"] == 0, r


@pytest.mark.parametrize("env,launches", [({}, 200), ({"SWN_FUSED_TAIL": "0"}, 100)])
def test_full_step_back_to_back_bit_exact(env, launches):
    """The launches the step actually runs, 200 full-size steps (8192 rays x 256 samples) back to back: the fused expert forward /
    backward launches (chainq tags 7 / 8), the front chains (tags 3 / 6) and the weight-gradient kernels that read what they stored;
    with SWN_FUSED_TAIL=0 the plain expert chains (tags 1 / 2) and the tail chains as their own launches.  Every repetition's gradient,
    rgb and saved tensors equal the first one's bit for bit, and the tile-queue counters are left at zero."""
    r = _probe("step", "--launches", str(launches), "--rays", "8192", env=env, timeout=900)
    assert r["ok"] and r["steps_that_differ_from_the_first"] == 0 and r["tile_queue_counters_left_nonzero"] == 0, r
    if not env:
        ks = r["kernel_set"]
        assert ks["geom"] == 7 and ks["front_geom"] == 7 and ks["tail_fused"] and ks["fused_backward"] and ks["comb_dwsig"], ks


def test_render_rays_graph_eval_matches_eager():
    """rendering.render_rays with nerf.graph_eval = True (the path Runner.render_image's pixel-batch loop takes): same results as
    the eager evaluation, for two different batches through the same cached graph, after an optimizer moved the weights."""
    import argparse
    from switch_nerf_amd import rendering
    m = _model(torch.bfloat16, 43)
    hp = argparse.Namespace(coarse_samples=64, fine_samples=32, model_chunk_size=8192, perturb=1.0, use_sigma_noise=True,
                            sigma_noise_std=1.0, moe_return_gates=True, return_sigma=True)
    m.eval()
    m.set_no_batch(True)
    for seed in (600, 601):
        rays, img, _ = synth.make_rays(seed, 256)
        m.graph_eval = False
        a, _ = rendering.render_rays(m, None, _dev(rays), _dev(img), hp, None, None, True, True)
        m.graph_eval = True
        b, _ = rendering.render_rays(m, None, _dev(rays), _dev(img), hp, None, None, True, True)
        assert set(a) == set(b)
        for k in a:
            assert torch.equal(a[k], b[k]), k
    assert len(m._render_graphs) == 1
