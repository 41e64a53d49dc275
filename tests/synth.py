"""Deterministic synthetic inputs and weights shared by the golden generator, the tests and bench.py.

Everything here is derived from numpy's PCG64 `default_rng(seed)` only, so the same arrays are
reproduced bit-for-bit in this container (where the reference is imported to make the golden
vectors) and on the GPU box (where the reference does not exist).

Shapes follow the reference's `building.yaml`
(/root/reference/switch_nerf/configs/switch_nerf/building.yaml:6-83) and parameter names follow the
reference's `state_dict()` key layout so that a real checkpoint of the
reference can be loaded through the same path.
"""
from __future__ import annotations

import numpy as np

BUILDING = dict(
    model_dim=256, num_experts=8, expert_layers=7, skips=(3,),
    pos_xyz_dim=12, pos_dir_dim=4, appearance_dim=48, appearance_count=10,
    gate_hidden=256, gate_layers=2, layer2_out=128,
)


def small_cfg(model_dim=64, num_experts=4, expert_layers=7, skips=(3,), appearance_count=10,
              appearance_dim=48, layer2_out=128):
    c = dict(BUILDING)
    c.update(model_dim=model_dim, num_experts=num_experts, expert_layers=expert_layers, skips=tuple(skips),
             gate_hidden=model_dim, appearance_count=appearance_count, appearance_dim=appearance_dim,
             layer2_out=layer2_out)
    return c


def _linear(rng, out_f, in_f, scale=1.0):
    """torch.nn.Linear default init distribution: U(-1/sqrt(in), 1/sqrt(in)) for weight and bias."""
    b = 1.0 / np.sqrt(in_f)
    w = rng.uniform(-b, b, size=(out_f, in_f)).astype(np.float32) * np.float32(scale)
    bias = rng.uniform(-b, b, size=(out_f,)).astype(np.float32) * np.float32(scale)
    return w, bias


def make_weights(seed: int, cfg=BUILDING, gate_scale: float = 1.0):
    """Return {state_dict key: np.float32 array} with the reference's key layout and shapes.

    gate_scale multiplies the router weight `wg`: 1.0 reproduces the (heavily unbalanced) routing of a
    random-init model, a small value (e.g. 1e-2) gives near-uniform softmax and a balanced routing.
    """
    rng = np.random.default_rng(seed)
    M, E, L = cfg["model_dim"], cfg["num_experts"], cfg["expert_layers"]
    in_xyz = 3 + 3 * 2 * cfg["pos_xyz_dim"]
    in_dir = 3 + 3 * 2 * cfg["pos_dir_dim"]
    H2 = cfg["layer2_out"]
    sd = {}
    sd["layers.xyz.fcs.0.weight"], sd["layers.xyz.fcs.0.bias"] = _linear(rng, M, in_xyz)
    for l in range(L):
        w = np.empty((E, M, M), np.float32)
        b = np.empty((E, 1, M), np.float32)
        for e in range(E):
            wt, bt = _linear(rng, M, M)
            w[e] = wt.T  # reference stores [E, in, out] (tutel_moe_layer_nobatch.py:861-864)
            b[e, 0] = bt
        sd[f"layers.0.experts.0.weights.{l}"] = w
        sd[f"layers.0.experts.0.bias.{l}"] = b
    wg, _ = _linear(rng, E, cfg["gate_hidden"])
    sd["layers.0.gates.0.wg.weight"] = wg * np.float32(gate_scale)
    sd["layers.1.fcs.0.weight"], sd["layers.1.fcs.0.bias"] = _linear(rng, M, M)
    sd["layers.2.fcs.0.weight"], sd["layers.2.fcs.0.bias"] = _linear(rng, H2, M + in_dir + cfg["appearance_dim"])
    sd["layers.sigma.fcs.0.weight"], sd["layers.sigma.fcs.0.bias"] = _linear(rng, 1, M)
    sd["layers.color.fcs.0.weight"], sd["layers.color.fcs.0.bias"] = _linear(rng, 3, H2)
    G = cfg["gate_hidden"]
    sd["layers.moe_external_gate.fcs.0.weight"], sd["layers.moe_external_gate.fcs.0.bias"] = _linear(rng, G, M)
    sd["layers.moe_external_gate.fcs.1.weight"], sd["layers.moe_external_gate.fcs.1.bias"] = _linear(rng, G, G)
    sd["layers.gate_input_norm.weight"] = (1.0 + 0.1 * rng.standard_normal(G)).astype(np.float32)
    sd["layers.gate_input_norm.bias"] = (0.1 * rng.standard_normal(G)).astype(np.float32)
    sd["embedding_a.weight"] = rng.standard_normal((cfg["appearance_count"], cfg["appearance_dim"])).astype(np.float32)
    return sd


DENSE = dict(layer_dim=256, layers=8, skip_layers=(4,), pos_xyz_dim=12, pos_dir_dim=4, appearance_dim=48, appearance_count=10,
             xyz_dim=3)


def make_dense_weights(seed: int, cfg=DENSE):
    """{state_dict key: np.float32 array} of the reference's dense NeRF (models/nerf.py NeRF: BASELINE configs[0], and the
    background network with xyz_dim = 4)."""
    rng = np.random.default_rng(seed)
    W, xd = cfg["layer_dim"], cfg["xyz_dim"]
    in_xyz = xd + xd * 2 * cfg["pos_xyz_dim"]
    in_dir = 3 + 3 * 2 * cfg["pos_dir_dim"]
    sd = {}
    for i in range(cfg["layers"]):
        k = in_xyz if i == 0 else (W + in_xyz if i in cfg["skip_layers"] else W)
        sd[f"xyz_encodings.{i}.0.weight"], sd[f"xyz_encodings.{i}.0.bias"] = _linear(rng, W, k)
    sd["embedding_a.weight"] = rng.standard_normal((cfg["appearance_count"], cfg["appearance_dim"])).astype(np.float32)
    sd["xyz_encoding_final.weight"], sd["xyz_encoding_final.bias"] = _linear(rng, W, W)
    sd["dir_a_encoding.0.weight"], sd["dir_a_encoding.0.bias"] = _linear(rng, W // 2, W + in_dir + cfg["appearance_dim"])
    sd["sigma.weight"], sd["sigma.bias"] = _linear(rng, 1, W)
    sd["rgb.weight"], sd["rgb.bias"] = _linear(rng, 3, W // 2)
    return sd


def make_rays(seed: int, n_rays: int, appearance_count: int = 10, near=0.05, far=1.0):
    """SURVEY.md section 8(d) synthetic rays: o~U(-0.1,0.1)^3, d=normalize(N(0,I)), near/far constants."""
    rng = np.random.default_rng(seed)
    o = rng.uniform(-0.1, 0.1, size=(n_rays, 3))
    d = rng.standard_normal((n_rays, 3))
    d = d / np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.concatenate([o, d, np.full((n_rays, 1), near), np.full((n_rays, 1), far)], axis=1).astype(np.float32)
    image_indices = rng.integers(0, appearance_count, size=(n_rays,)).astype(np.int64)
    rgbs = rng.uniform(0.0, 1.0, size=(n_rays, 3)).astype(np.float32)
    return rays, image_indices, rgbs


DENSE_BG = dict(DENSE, xyz_dim=4)          # the background model (models/model_utils.py:73-84 get_bg_nerf: xyz_dim 4)
SPHERE_CENTER = np.array([0.02, -0.03, 0.01], np.float32)      # ellipsoidal foreground bound (runner.py:221-243)
SPHERE_RADIUS = np.array([0.6, 0.8, 0.7], np.float32)


def make_bg_rays(seed: int, n_rays: int, appearance_count: int = 10):
    """Rays for the background path: origins inside the foreground ellipsoid, far in U(0.3, 1.5) so that some rays stop
    inside the bound (no background) and the others leave it (continued by the background model)."""
    rays, img, rgbs = make_rays(seed, n_rays, appearance_count)
    rng = np.random.default_rng(seed + 7919)
    rays[:, 7] = rng.uniform(0.3, 1.5, n_rays).astype(np.float32)
    return rays, img, rgbs


def make_gates(seed: int, n_tokens: int, n_experts: int, logit_scale: float = 1.0, quantize_bits: int = 0):
    """Softmax probabilities [P,E] fp32.  quantize_bits>0 rounds logits to a coarse grid to force ties."""
    rng = np.random.default_rng(seed)
    logits = (logit_scale * rng.standard_normal((n_tokens, n_experts))).astype(np.float32)
    if quantize_bits > 0:
        q = np.float32(2.0 ** quantize_bits)
        logits = np.round(logits * q) / q
    m = logits.max(axis=1, keepdims=True)
    e = np.exp(logits - m, dtype=np.float32)
    return (e / e.sum(axis=1, keepdims=True, dtype=np.float32)).astype(np.float32)


def checksum(a) -> np.ndarray:
    """Order-independent-ish summary used for large gradient tensors: [sum, abs-sum, sq-sum] in fp64."""
    a = np.asarray(a, dtype=np.float64)
    return np.array([a.sum(), np.abs(a).sum(), (a * a).sum()], dtype=np.float64)
