"""CPU oracle for the Switch-NeRF train hot path.  TEST INFRASTRUCTURE ONLY.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this module.
The product path (`switch_nerf_amd/`) never imports it and fails loudly when its HIP library is missing.

This is a from-scratch restatement (plain torch-CPU fp32 for the floating-point maths, numpy integer code
for the routing) of what the reference computes on the path
`render_rays -> _inference -> NeRFMoE.forward -> MOELayer.forward -> extract_critical / encode /
ExpertMLP / decode -> volumetric compositing`.  Each function cites the reference lines it follows
(paths relative to /root/reference/switch_nerf/).

Pinning: the reference holds no tests or golden vectors for this path, so this
oracle is pinned against outputs of the reference itself, imported and run in the build container by
`oracle/gen_golden.py` (fixtures under tests/golden/, checked by tests/test_oracle_golden.py).
The Tutel kernels the reference calls are a third-party dependency that is absent from /root/reference
(tutel @ 56dbd664341cf6485c9fa292955f77d3ac918a65); their semantics are restated in oracle/stubs/ and
are "parity unpinned" at that boundary (no upstream vectors exist).

Tie semantics: the reference ranks tokens with an unstable argsort and picks experts with
topk, so its answer is implementation-defined when two gate values are exactly equal.  This oracle (and
the HIP path) define the deterministic answer: expert = FIRST index of the row maximum, rank = STABLE
descending order of the row maximum (earlier token wins a tie).  On tie-free inputs this equals the
reference bit for bit (tests/golden/route_*.npz).
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------------------------
# the reference's mixed-precision mode (torch autocast), restated
# --------------------------------------------------------------------------------------------


class Autocast:
    """The dtype policy the reference TRAINS under: runner.py:593-598 wraps the training step in
    `torch.cuda.amp.autocast(enabled=hparams.amp, dtype=bfloat16 if hparams.amp_use_bfloat16 else float16)` (rendering: :2845-2846).
    torch's autocast is a per-operator table; the operators on this path and what the table does to them:

      * lower-precision list (inputs cast to the autocast dtype, output in it): linear, baddbmm  -> every nn.Linear of NeRFMoE
        (models/nerf_moe.py:30-49, 330-441) and ExpertMLP's baddbmm (tutel_moe_layer_nobatch.py:908) - weights AND biases are
        rounded to the 16-bit type, products accumulate in fp32, the result is rounded once;
      * fp32 list of the CUDA backend (inputs cast up, fp32 output): layer_norm (the gate-input norm, nerf_moe.py:370-372),
        softplus (the sigma activation :416), softmax, cumprod, sum, exp, pow, mse_loss.  The CPU backend's fp32 list holds NEITHER
        layer_norm NOR softplus: there both run - and round - in the 16-bit type of their input.  That is the only difference
        between the two backends on this path (`policy`);
      * everything else (relu, sigmoid, add, sub, mul, cat, indexing) runs in the promoted type of its inputs: relu / the skip add
        on 16-bit activations stay 16-bit, sigmoid of the 16-bit colour logits is 16-bit (rounded), `sigma += sigma_noise` is an
        in-place add on a 16-bit tensor (stays 16-bit), cat([h (16-bit), dir encoding, appearance (fp32)]) promotes to fp32 and the
        next Linear rounds it again, cat([rgb (16-bit), sigma]) promotes to fp32 under CUDA.
    Explicit islands in the reference's code (independent of the table): the router runs with autocast OFF on `.float()` inputs
    (tutel_moe_layer_nobatch.py:105-113, fp32_gate), the dispatcher encodes / decodes in fp32 and casts the result back to the
    activation dtype (tutel_fast_dispatch.py:89-93, 119-127), the sigma head runs in the autocast dtype only with
    amp_use_bfloat16, otherwise in an autocast-off fp32 island (nerf_moe.py:396-400).

    policy "cuda" is what the reference's GPU training computes (the benchmarked dtype).  policy "cpu" exists to PIN this
    restatement: oracle/gen_golden.py runs the reference itself under torch.autocast("cpu", bfloat16) (its torch.cuda.amp.autocast
    islands mapped onto the CPU autocast state) and tests/test_oracle_golden.py checks this class against those outputs; the
    operator table of either backend is checked against torch itself (tests/test_oracle_golden.py on the CPU,
    tests/test_fullsize_gpu.py under torch.autocast("cuda") on the GPU box).
    Matrix products are evaluated as fp32 products of the ROUNDED operands (products of two 16-bit floats are exact in fp32,
    the sums run in fp32, one final rounding) - what cuBLAS / hipBLASLt / oneDNN kernels do, up to the order of the sums."""

    FP32_OPS = {"cuda": ("layer_norm", "softplus", "softmax", "cumprod", "sum", "exp", "mse_loss"),
                "cpu": ("mse_loss",)}

    def __init__(self, dtype=torch.bfloat16, policy: str = "cuda", sigma_head_lowp: Optional[bool] = None):
        assert policy in ("cuda", "cpu") and dtype in (torch.bfloat16, torch.float16)
        self.dtype, self.policy = dtype, policy
        # nerf_moe.py:396-400: the sigma Linear is autocast only with amp_use_bfloat16 (= the bf16 recipes)
        self.sigma_head_lowp = (dtype == torch.bfloat16) if sigma_head_lowp is None else bool(sigma_head_lowp)

    def lp(self, t: torch.Tensor) -> torch.Tensor:
        return t.to(self.dtype)

    def linear(self, x, w, b=None):
        y = F.linear(self.lp(x).float(), self.lp(w).float(), None if b is None else self.lp(b).float())
        return y.to(self.dtype)

    def baddbmm(self, b, x, w):
        return torch.baddbmm(self.lp(b).float(), self.lp(x).float(), self.lp(w).float()).to(self.dtype)

    def fp32_op(self, name: str) -> bool:
        return name in self.FP32_OPS[self.policy]

    def layer_norm(self, x, w, b, eps):
        if self.fp32_op("layer_norm"):
            return F.layer_norm(x.float(), (x.shape[1],), w, b, eps)
        # CPU backend: not in any list -> the kernel runs on the 16-bit input (fp32 statistics inside), output rounded to 16 bits
        return F.layer_norm(x.float(), (x.shape[1],), w, b, eps).to(x.dtype)

    def softplus(self, x, beta, threshold):
        if self.fp32_op("softplus"):
            return F.softplus(x.float(), beta, threshold)
        return F.softplus(x.float(), beta, threshold).to(x.dtype)


# --------------------------------------------------------------------------------------------
# positional encoding / sampling
# --------------------------------------------------------------------------------------------


def positional_encoding(x: torch.Tensor, num_freqs: int) -> torch.Tensor:
    """models/nerf.py:21-26 - [x, sin(2^k x), cos(2^k x)] for k = 0..L-1, concatenated per frequency."""
    out = [x]
    for k in range(num_freqs):
        f = float(2 ** k)
        out += [torch.sin(f * x), torch.cos(f * x)]
    return torch.cat(out, -1)


def sample_z(near: torch.Tensor, far: torch.Tensor, n_samples: int, perturb: float = 0.0,
             perturb_rand: Optional[torch.Tensor] = None) -> torch.Tensor:
    """rendering.py:85-88 and :573-584.  near/far: [N,1].  perturb_rand: the U[0,1) tensor the reference
    draws with rand_like (supplied by the caller so that runs are reproducible)."""
    t = torch.linspace(0, 1, n_samples, dtype=near.dtype)
    z = near * (1 - t) + far * t
    if perturb > 0:
        mid = 0.5 * (z[:, :-1] + z[:, 1:])
        upper = torch.cat([mid, z[:, -1:]], -1)
        lower = torch.cat([z[:, :1], mid], -1)
        z = lower + (upper - lower) * (perturb * perturb_rand)
    return z


# --------------------------------------------------------------------------------------------
# routing (integer work, numpy)
# --------------------------------------------------------------------------------------------


def capacity_of(n_tokens: int, n_experts: int, capacity_factor: float, top_k: int = 1) -> int:
    """tutel_fast_dispatch.py:211 - top_k * int(cf * ceil(P / E))."""
    return top_k * int(capacity_factor * ((int(n_tokens) + n_experts - 1) // n_experts))


def route_top1(gates: np.ndarray, capacity_factor: float, batch_prioritized: bool):
    """extract_critical, tutel_fast_dispatch.py:176-217, for top_k = 1.

    gates: [P, E] fp32 softmax probabilities.
    Returns dict(idx int32 [P], loc int32 [P], gate fp32 [P], capacity int, counts int32 [E]).
    loc[i] = number of tokens routed to the same expert that are ranked before token i, where the
    ranking is token order (plain) or stable descending order of max-gate (batch prioritized,
    compute_sorted_location :136-139).  A token is kept iff loc < capacity.
    """
    gates = np.ascontiguousarray(gates, dtype=np.float32)
    P, E = gates.shape
    idx = gates.argmax(axis=1).astype(np.int32)            # first max index (topk on tie-free rows)
    gmax = gates[np.arange(P), idx]                          # == (gates * one_hot).sum(1), :182
    if batch_prioritized:
        order = np.argsort(-gmax, kind="stable")            # importance = -max gate, ascending
    else:
        order = np.arange(P)
    loc = np.empty(P, np.int32)
    counts = np.zeros(E, np.int64)
    idx_sorted = idx[order]
    # cumsum-1 of the one-hot mask in `order`, read back at each token's own expert column (:138, :194)
    for e in range(E):
        sel = order[idx_sorted == e]
        loc[sel] = np.arange(sel.shape[0], dtype=np.int32)
        counts[e] = sel.shape[0]
    cap = capacity_of(P, E, capacity_factor)
    # top-2 gap of every token (max - second max of its gate row): how close its expert choice is to a tie - parity tests only accept
    # an index that differs from this oracle's where the gap is at rounding-noise level
    part = np.partition(gates, E - 2, axis=1) if E > 1 else gates
    gap = (part[:, E - 1] - part[:, E - 2]).astype(np.float32) if E > 1 else np.ones(P, np.float32)
    return dict(idx=idx, loc=loc, gate=gmax.astype(np.float32), capacity=cap, counts=counts.astype(np.int32), top2_gap=gap)


def route_topk(gates: np.ndarray, top_k: int, capacity_factor: float, batch_prioritized: bool):
    """extract_critical, tutel_fast_dispatch.py:176-217, for top_k >= 1: choice j of every token = its j-th largest gate (torch.topk,
    :177); locations of choice j = the rank among the tokens whose choice j is the same expert - token order, or stable descending order
    of the TOKEN's max gate (:187: the same importance for every choice) - plus acc_base = the number of tokens earlier choices sent to
    that expert (:199-201); capacity = top_k * int(cf * ceil(P / E)) (:211).
    Returns dict(idx int32 [k, P], loc int32 [k, P], capacity, counts int32 [k, E])."""
    gates = np.ascontiguousarray(gates, dtype=np.float32)
    P, E = gates.shape
    topk = np.argsort(-gates, axis=1, kind="stable")[:, :top_k]          # descending, lower expert first on exact ties
    gmax = gates.max(axis=1)
    order = np.argsort(-gmax, kind="stable") if batch_prioritized else np.arange(P)
    idx = np.ascontiguousarray(topk.T).astype(np.int32)
    loc = np.empty((top_k, P), np.int32)
    counts = np.zeros((top_k, E), np.int64)
    base = np.zeros(E, np.int64)
    for j in range(top_k):
        ij = idx[j][order]
        for e in range(E):
            sel = order[ij == e]
            loc[j, sel] = base[e] + np.arange(sel.shape[0], dtype=np.int64)
            counts[j, e] = sel.shape[0]
        base = base + counts[j]
    return dict(idx=idx, loc=loc, capacity=capacity_of(P, E, capacity_factor, top_k), counts=counts.astype(np.int32))


def load_importance_loss(scores_wo_noise: torch.Tensor, topk_logits: torch.Tensor, n_experts: int, gate_noise: float) -> torch.Tensor:
    """load_importance_loss, tutel_fast_dispatch.py:152-174: (cv2(importance) + cv2(load)) / 2 with importance_e = sum_t scores[t, e],
    load_e = sum_t Normal(0, gate_noise / E).cdf(scores[t, e] - the token's k-th largest noisy logit), cv2(v) = var(v) / (mean(v)^2 + 1e-10)
    (unbiased variance)."""
    assert gate_noise > 0
    sigma = gate_noise / n_experts
    thr = topk_logits[:, -1].view(-1, 1).float()
    prob = 0.5 * (1 + torch.erf((scores_wo_noise.float() - thr) / sigma / math.sqrt(2.0)))          # Normal.cdf
    load = prob.sum(0)
    imp = scores_wo_noise.float().sum(0)
    cv2 = lambda v: v.var() / (v.mean() ** 2 + 1e-10)
    return (cv2(imp) + cv2(load)) / 2.0


def moe_layer_topk(h: torch.Tensor, gate_input: torch.Tensor, wg: torch.Tensor, weights, biases, skips, top_k: int,
                   capacity_factor: float, batch_prioritized: bool, gate_noise: float = 0.0, noise: Optional[torch.Tensor] = None,
                   load_importance: bool = False):
    """TopKGate.apply_on_expert_fn (tutel_moe_layer_nobatch.py:98-235) with k > 1, fp32 gate, postscore: gates normalised by the sum of the
    token's k gates (tutel_fast_dispatch.py:204-206), dispatch / combine summed over the choices (:26-27, :59-62), l_aux from the first
    choice's mask (:184).  Returns (y, l_aux, routing dict, gates)."""
    E = wg.shape[0]
    logits = gate_input.float() @ wg.float().t()
    logits_w = logits + gate_noise * noise / E if (gate_noise > 0 and noise is not None) else logits     # tutel_moe_layer_nobatch.py:119-122
    gates = torch.softmax(logits_w, dim=1)
    r = route_topk(gates.detach().numpy(), top_k, capacity_factor, batch_prioritized)
    cap = int(r["capacity"])
    idx = [torch.from_numpy(r["idx"][j].astype(np.int64)) for j in range(top_k)]
    loc = [torch.from_numpy(r["loc"][j].astype(np.int64)) for j in range(top_k)]
    gs = [gates.gather(1, i.unsqueeze(1)).squeeze(1) for i in idx]
    if top_k > 1:                                                              # tutel_fast_dispatch.py:196, 204-206
        denom = torch.clamp(sum(gs), min=torch.finfo(gs[0].dtype).eps)
        gs = [g / denom for g in gs]
    l_aux = load_balance_loss(gates, idx[0])
    if load_importance:                                                        # :232 (l_aux -> routing["balance_loss"], the extras' tensor)
        r = dict(r, balance_loss=l_aux)
        l_aux = load_importance_loss(torch.softmax(logits, dim=1), logits_w.gather(1, torch.stack(idx, dim=1)), E, gate_noise)
    d = sum(dispatch(h, idx[j], loc[j], E, cap) for j in range(top_k)).view(E, cap, -1)
    o = expert_mlp(d, weights, biases, skips).reshape(E * cap, -1)
    y = sum(combine(o, idx[j], loc[j], gs[j], cap) for j in range(top_k))
    return y, l_aux, r, gates


def load_balance_loss(gates: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """load_balance fp32 branch, tutel_fast_dispatch.py:141-145: sum_e(me*ce) * E / P^2."""
    P, E = gates.shape
    me = gates.float().sum(0)
    ce = F.one_hot(idx.long(), E).to(me.dtype).sum(0)
    return (me * ce).sum() * (E / (P * P))


# --------------------------------------------------------------------------------------------
# dispatch / expert MLP / combine
# --------------------------------------------------------------------------------------------


def dispatch(x: torch.Tensor, idx: torch.Tensor, loc: torch.Tensor, n_experts: int, capacity: int) -> torch.Tensor:
    """GatingEncoder.forward with is_postscore (gate = 1), tutel_fast_dispatch.py:17-29: zero [E*C, M],
    row idx*C+loc receives x[i] for kept tokens."""
    keep = loc < capacity
    rows = (idx.long() * capacity + loc.long())[keep]
    d = torch.zeros(n_experts * capacity, x.shape[1], dtype=x.dtype)
    return d.index_add(0, rows, x[keep])


def combine(d: torch.Tensor, idx: torch.Tensor, loc: torch.Tensor, gate: torch.Tensor, capacity: int) -> torch.Tensor:
    """GatingDecoder.forward, tutel_fast_dispatch.py:50-63: y[i] = gate[i] * D[idx*C+loc], 0 if dropped."""
    keep = loc < capacity
    rows = (idx.long() * capacity + loc.long()).clamp(max=d.shape[0] - 1)
    y = gate.unsqueeze(1) * d[rows]
    return torch.where(keep.unsqueeze(1), y, torch.zeros_like(y))


def route_top1_nobatch(gates: np.ndarray):
    """extract_critical of the no-batch path, tutel_fast_dispatch_nobatch.py:205-251, top_k = 1, position-order ranking
    (fast_cumsum_sub_one): idx = argmax, loc = rank of the token among the earlier tokens of its expert, expert_input_nums =
    tokens per expert (:222), expert_locations_begin = their exclusive prefix sum (:26-27).  Nothing is dropped."""
    r = route_top1(gates, 1.0, False)
    nums = r["counts"].astype(np.int32)
    begin = (np.cumsum(nums) - nums).astype(np.int32)
    return dict(idx=r["idx"], loc=r["loc"], gate=r["gate"], expert_input_nums=nums, expert_locations_begin=begin)


def dispatch_nobatch(x: torch.Tensor, idx: torch.Tensor, loc: torch.Tensor, begin: torch.Tensor) -> torch.Tensor:
    """GatingEncoder (tutel_fast_dispatch_nobatch.py:16-37) + the forward kernel tutel_sparse_nobatch.py:24-36, postscore (gate = 1):
    D[begin[idx] + loc] += x, rows packed contiguously per expert, no capacity test.  Differentiable in x."""
    rows = begin.long()[idx.long()] + loc.long()
    return torch.zeros(x.shape[0], x.shape[1], dtype=x.dtype).index_add(0, rows, x)


def combine_nobatch(d: torch.Tensor, idx: torch.Tensor, loc: torch.Tensor, begin: torch.Tensor, gate: torch.Tensor) -> torch.Tensor:
    """GatingDecoder (:62-78) + the backward-data kernel tutel_sparse_nobatch.py:39-61: y[i] = gate[i] * D[begin[idx] + loc].
    Differentiable in d and gate (the kernels :24-36 and :64-133 are its autograd backward)."""
    rows = begin.long()[idx.long()] + loc.long()
    return gate.unsqueeze(1) * d[rows]


def expert_mlp(d: torch.Tensor, weights: Sequence[torch.Tensor], biases: Sequence[torch.Tensor],
               skips: Sequence[int], autocast: Optional[Autocast] = None) -> torch.Tensor:
    """ExpertMLP.forward, tutel_moe_layer_nobatch.py:887-924.  d: [E, C, M]; weights[l]: [E, in, out];
    biases[l]: [E, 1, out].  ReLU after every layer but the last; at a skip layer the layer input saved
    at the previous skip (initially the expert input) is added before the ReLU."""
    L = len(weights)
    x = d
    h = d
    for l in range(L):
        h = torch.baddbmm(biases[l], h, weights[l]) if autocast is None else autocast.baddbmm(biases[l], h, weights[l])
        if l in skips:
            h = h + x
            if l < L - 1:
                h = torch.relu(h)
            x = h
        elif l < L - 1:
            h = torch.relu(h)
    return h


def moe_layer(h: torch.Tensor, gate_input: torch.Tensor, wg: torch.Tensor, weights, biases, skips,
              capacity_factor: float, batch_prioritized: bool, routing: Optional[dict] = None, no_batch: bool = False,
              autocast: Optional[Autocast] = None, gate_noise: float = 0.0, noise: Optional[torch.Tensor] = None,
              normal_noise: Optional[torch.Tensor] = None):
    """TopKGate.apply_on_expert_fn, tutel_moe_layer_nobatch.py:98-235 (k=1, fp32 gate, postscore).
    Returns (y [P,M], l_aux, routing dict, gates [P,E]).  `routing` may be injected (idx/loc numpy) to
    decouple numerics tests from near-tie routing flips.  gate_noise / noise: the gate-noise branch of a training forward;
    normal_noise: the use_normal_noise branch's draw (:116-117), added first."""
    E = wg.shape[0]
    logits = gate_input.float() @ wg.float().t()                               # :105-113
    if normal_noise is not None:                                               # use_normal_noise and training (:116-117)
        logits = logits + normal_noise / E
    if gate_noise > 0 and noise is not None:                                   # training, --gate_noise > 0 (:119-122): `noise` stands for
        logits = logits + gate_noise * noise / E                               # the layer's torch.randn_like(logits) draw
    gates = torch.softmax(logits, dim=1)                                       # :126
    if routing is None:
        routing = route_top1(gates.detach().numpy(), capacity_factor, batch_prioritized)
        if no_batch:
            # eval path, apply_on_expert_fn_nobatch (tutel_moe_layer_nobatch.py:237-352) + tutel_sparse_nobatch.py: every
            # token is processed by its expert (no capacity test); equivalent to a capacity that nothing exceeds
            routing = dict(routing, capacity=int(gates.shape[0]))
    idx = torch.from_numpy(np.asarray(routing["idx"]).astype(np.int64))
    loc = torch.from_numpy(np.asarray(routing["loc"]).astype(np.int64))
    cap = int(routing["capacity"])
    gate_s = gates.gather(1, idx.unsqueeze(1)).squeeze(1)                      # differentiable gates_s, :182
    l_aux = load_balance_loss(gates, idx)                                      # :184
    if autocast is None:
        d = dispatch(h, idx, loc, E, cap).view(E, cap, -1)                     # :146
        o = expert_mlp(d, weights, biases, skips).reshape(E * cap, -1)         # :174
        y = combine(o, idx, loc, gate_s, cap)                                  # :225
    else:
        # the dispatcher works in fp32 and hands back the activation dtype (tutel_fast_dispatch.py:89-93, 119-127: data.to(fp32),
        # ....to(original_dtype)); the experts' baddbmm is an autocast operator (tutel_moe_layer_nobatch.py:908)
        d = dispatch(h.float(), idx, loc, E, cap).to(h.dtype).view(E, cap, -1)
        o = expert_mlp(d, weights, biases, skips, autocast).reshape(E * cap, -1)
        y = combine(o.float(), idx, loc, gate_s, cap).to(h.dtype)
    return y, l_aux, routing, gates


# --------------------------------------------------------------------------------------------
# NeRFMoE forward (building.yaml wiring)
# --------------------------------------------------------------------------------------------


def shifted_softplus(x: torch.Tensor) -> torch.Tensor:
    """models/nerf.py:68-69 - softplus(x - 1), beta 1, threshold 20."""
    return F.softplus(x - 1, 1, 20)


def params_from_numpy(sd: Dict[str, np.ndarray], requires_grad: bool = False) -> Dict[str, torch.Tensor]:
    out = {}
    for k, v in sd.items():
        t = torch.from_numpy(np.array(v, dtype=np.float32, copy=True))
        t.requires_grad_(requires_grad)
        out[k] = t
    return out


def nerf_moe_forward(p: Dict[str, torch.Tensor], x: torch.Tensor, cfg: dict, capacity_factor: float = 1.0,
                     batch_prioritized: bool = True, sigma_noise: Optional[torch.Tensor] = None,
                     routing: Optional[dict] = None, no_batch: bool = False, encoded=None, autocast: Optional[Autocast] = None,
                     gate_noise: float = 0.0, gate_noise_draw: Optional[torch.Tensor] = None):
    """NeRFMoE.forward, models/nerf_moe.py:320-455, with building.yaml's layer wiring.
    gate_noise / gate_noise_draw [P, E]: the gate-noise branch of a training forward (moe_layer; fp32 path only).
    x: [P, 7] = xyz(3) dir(3) image_index(1).  Returns dict(outputs [P,4], moe_loss [1], routing, gates).
    encoded = (xyz encoding [P, 3 + 6 L], dirs [P,3], image index [P]): MipNeRFMoE.forward (:675-810), which is the same
    network behind a different position encoder (MipEmbedder)."""
    L = cfg["expert_layers"]
    if encoded is None:
        xyz, dirs, img = x[:, :3], x[:, 3:6], x[:, 6].long()
        enc = positional_encoding(xyz, cfg["pos_xyz_dim"])
    else:
        enc, dirs, img = encoded[0], encoded[1], encoded[2].long()
    if autocast is not None:
        assert gate_noise_draw is None, "gate noise: fp32 path only"
        return _nerf_moe_forward_autocast(p, enc, dirs, img, cfg, capacity_factor, batch_prioritized, sigma_noise, routing, no_batch,
                                          autocast)
    h = F.linear(enc, p["layers.xyz.fcs.0.weight"], p["layers.xyz.fcs.0.bias"])  # :330-333
    g = F.linear(h, p["layers.moe_external_gate.fcs.0.weight"], p["layers.moe_external_gate.fcs.0.bias"])
    g = F.linear(torch.relu(g), p["layers.moe_external_gate.fcs.1.weight"], p["layers.moe_external_gate.fcs.1.bias"])  # :347-348, Mlp :30-49
    g = F.layer_norm(g, (g.shape[1],), p["layers.gate_input_norm.weight"], p["layers.gate_input_norm.bias"], 1e-5)  # :370-372
    weights = [p[f"layers.0.experts.0.weights.{l}"] for l in range(L)]
    biases = [p[f"layers.0.experts.0.bias.{l}"] for l in range(L)]
    y, l_aux, routing, gates = moe_layer(h, g, p["layers.0.gates.0.wg.weight"], weights, biases, cfg["skips"],
                                         capacity_factor, batch_prioritized, routing, no_batch, gate_noise=gate_noise, noise=gate_noise_draw)
    y = torch.relu(y)                                                          # act: relu, :385-386
    sigma = F.linear(y, p["layers.sigma.fcs.0.weight"], p["layers.sigma.fcs.0.bias"])  # :393-400
    if sigma_noise is not None:
        sigma = sigma + sigma_noise                                            # :414-415
    sigma = shifted_softplus(sigma)                                            # :416
    h1 = F.linear(y, p["layers.1.fcs.0.weight"], p["layers.1.fcs.0.bias"])     # layer "1", act none
    feat = torch.cat([h1, positional_encoding(dirs, cfg["pos_dir_dim"]), p["embedding_a.weight"][img]], -1)  # :419-429
    h2 = torch.relu(F.linear(feat, p["layers.2.fcs.0.weight"], p["layers.2.fcs.0.bias"]))
    rgb = torch.sigmoid(F.linear(h2, p["layers.color.fcs.0.weight"], p["layers.color.fcs.0.bias"]))  # :431-441
    return dict(outputs=torch.cat([rgb, sigma], -1), moe_loss=l_aux.reshape(1), routing=routing, gates=gates)


def _nerf_moe_forward_autocast(p, enc, dirs, img, cfg, capacity_factor, batch_prioritized, sigma_noise, routing, no_batch, ac: Autocast):
    """nerf_moe_forward under the reference's autocast (class Autocast: which operator rounds where).  Same wiring, same citations."""
    L = cfg["expert_layers"]
    h = ac.linear(enc, p["layers.xyz.fcs.0.weight"], p["layers.xyz.fcs.0.bias"])               # 16-bit
    g = ac.linear(h, p["layers.moe_external_gate.fcs.0.weight"], p["layers.moe_external_gate.fcs.0.bias"])
    g = ac.linear(torch.relu(g), p["layers.moe_external_gate.fcs.1.weight"], p["layers.moe_external_gate.fcs.1.bias"])
    g = ac.layer_norm(g, p["layers.gate_input_norm.weight"], p["layers.gate_input_norm.bias"], 1e-5)   # fp32 (CUDA) / 16-bit (CPU)
    weights = [p[f"layers.0.experts.0.weights.{l}"] for l in range(L)]
    biases = [p[f"layers.0.experts.0.bias.{l}"] for l in range(L)]
    y, l_aux, routing, gates = moe_layer(h, g, p["layers.0.gates.0.wg.weight"], weights, biases, cfg["skips"],
                                         capacity_factor, batch_prioritized, routing, no_batch, autocast=ac)
    y = torch.relu(y)                                                                            # 16-bit
    if ac.sigma_head_lowp:                                                                       # :396-397
        sigma = ac.linear(y, p["layers.sigma.fcs.0.weight"], p["layers.sigma.fcs.0.bias"])
    else:                                                                                        # :398-400: autocast off, fp32
        sigma = F.linear(y.float(), p["layers.sigma.fcs.0.weight"], p["layers.sigma.fcs.0.bias"])
    if sigma_noise is not None:
        sigma = (sigma.float() + sigma_noise).to(sigma.dtype)                                    # :414-415 `sigma += noise`: in place
    sigma = ac.softplus(sigma - 1, 1, 20)                                                        # nerf.py:68-69; `x - 1` in x's dtype
    h1 = ac.linear(y, p["layers.1.fcs.0.weight"], p["layers.1.fcs.0.bias"])
    feat = torch.cat([h1.float(), positional_encoding(dirs, cfg["pos_dir_dim"]), p["embedding_a.weight"][img]], -1)   # cat promotes to fp32
    h2 = torch.relu(ac.linear(feat, p["layers.2.fcs.0.weight"], p["layers.2.fcs.0.bias"]))
    rgb = torch.sigmoid(ac.linear(h2, p["layers.color.fcs.0.weight"], p["layers.color.fcs.0.bias"]).float()).to(ac.dtype)   # sigmoid: 16-bit
    outputs = torch.cat([rgb, sigma], -1) if rgb.dtype == sigma.dtype else torch.cat([rgb.float(), sigma.float()], -1)
    return dict(outputs=outputs, moe_loss=l_aux.reshape(1), routing=routing, gates=gates)


# --------------------------------------------------------------------------------------------
# volumetric compositing and the training step
# --------------------------------------------------------------------------------------------


def composite(rgbs: torch.Tensor, sigmas: torch.Tensor, z_vals: torch.Tensor, last_delta=1e10, flip: bool = False,
              depth_real: Optional[torch.Tensor] = None):
    """rendering.py:435-494.  rgbs [N,S,3], sigmas [N,S], z_vals [N,S]; last_delta a number or a [N,1] tensor (the distance
    to the foreground bound for rays that continue into the background model, :32-46, :216-217); flip: z_vals descend
    (the background's inverse-distance samples, :436-437); depth_real: the metric depths the depth map is taken over (:483-484).
    bg_lambda = the transmittance left after the last sample (:456-457)."""
    deltas = (z_vals[:, :-1] - z_vals[:, 1:]) if flip else (z_vals[:, 1:] - z_vals[:, :-1])    # :436-439
    ld = last_delta if torch.is_tensor(last_delta) else torch.full_like(z_vals[:, :1], last_delta)
    deltas = torch.cat([deltas, ld], -1)                                                       # :441
    alphas = 1 - torch.exp(-deltas * sigmas)                                                   # :442
    T = torch.cumprod(1 - alphas + 1e-8, -1)                                                   # :455
    bg_lambda = T[:, -1]                                                                       # :457
    T = torch.cat((torch.ones_like(T[:, :1]), T[:, :-1]), -1)                                  # :459
    weights = alphas * T                                                                       # :461
    rgb = (weights.unsqueeze(-1) * rgbs).sum(1)                                                # :467
    with torch.no_grad():
        depth = (weights * (z_vals if depth_real is None else depth_real)).sum(1)              # :483-486
        depth_var = (weights * (z_vals - depth.unsqueeze(1)).square()).sum(-1)                 # :491-493
    return dict(rgb=rgb, weights=weights, depth=depth, depth_variance=depth_var, alphas=alphas, bg_lambda=bg_lambda)


def sample_pdf(bins: torch.Tensor, weights: torch.Tensor, n_fine: int, u: Optional[torch.Tensor] = None):
    """rendering.py:587-637 (_sample_pdf/_sample_cdf).  u=None -> deterministic linspace (eval)."""
    w = weights + 1e-8
    pdf = w / w.sum(-1, keepdim=True)
    cdf = torch.cat([torch.zeros_like(pdf[:, :1]), torch.cumsum(pdf, -1)], -1)
    n = pdf.shape[1]
    if u is None:
        u = torch.linspace(0, 1, n_fine).expand(bins.shape[0], n_fine)
    u = u.contiguous()
    inds = torch.searchsorted(cdf, u, right=True)
    below = (inds - 1).clamp_min(0)
    above = inds.clamp_max(n)
    cdf_b, cdf_a = cdf.gather(1, below), cdf.gather(1, above)
    bin_b, bin_a = bins.gather(1, below), bins.gather(1, above)
    denom = cdf_a - cdf_b
    denom = torch.where(denom < 1e-8, torch.ones_like(denom), denom)
    return bin_b + (u - cdf_b) / denom * (bin_a - bin_b)


def _eval_points(p, rays, image_indices, z, cfg, chunk, capacity_factor, batch_prioritized, sigma_noise, routings, hash_cfg=None,
                 autocast: Optional[Autocast] = None, gate_noise: float = 0.0, gate_noise_draw: Optional[torch.Tensor] = None):
    """The chunked network evaluation of _inference (rendering.py:311-383): routing (capacity, ranking, l_aux) is per chunk.
    hash_cfg: the positions enter through the hash-grid encoding (p["embedding_xyz.table"]) instead of the frequency one."""
    N, S = z.shape
    o, d = rays[:, 0:3], rays[:, 3:6]
    xyz = o[:, None, :] + d[:, None, :] * z[:, :, None]                                      # :90 / xyz_fine_fn :103
    pts = torch.cat([xyz.reshape(-1, 3), d[:, None, :].expand(N, S, 3).reshape(-1, 3),
                     image_indices.view(N, 1, 1).expand(N, S, 1).reshape(-1, 1).to(xyz.dtype)], 1)  # :311-360
    enc = hash_encode(pts[:, :3], p["embedding_xyz.table"], hash_cfg) if hash_cfg is not None else None
    outs, losses, routes = [], [], []
    for ci, i in enumerate(range(0, pts.shape[0], chunk)):
        sn = None if sigma_noise is None else sigma_noise[i:i + chunk]
        r = nerf_moe_forward(p, pts[i:i + chunk], cfg, capacity_factor, batch_prioritized, sn,
                             None if routings is None else routings[ci],
                             encoded=None if enc is None else (enc[i:i + chunk], pts[i:i + chunk, 3:6], pts[i:i + chunk, 6]),
                             autocast=autocast, gate_noise=gate_noise,
                             gate_noise_draw=None if gate_noise_draw is None else gate_noise_draw[i:i + chunk])
        outs.append(r["outputs"])
        losses.append(r["moe_loss"])
        routes.append(r["routing"])
    return torch.cat(outs, 0).view(N, S, 4), torch.cat(losses, 0), routes


def render_rays(p, rays: torch.Tensor, image_indices: torch.Tensor, cfg: dict, n_samples: int, chunk: int,
                capacity_factor: float = 1.0, batch_prioritized: bool = True, perturb: float = 0.0,
                perturb_rand: Optional[torch.Tensor] = None, sigma_noise: Optional[torch.Tensor] = None,
                routings: Optional[list] = None, fine_samples: int = 0, fine_u: Optional[torch.Tensor] = None,
                sigma_noise_fine: Optional[torch.Tensor] = None, hash_cfg: Optional[dict] = None,
                autocast: Optional[Autocast] = None, gate_noise: float = 0.0, gate_noise_draw: Optional[torch.Tensor] = None):
    """render_rays + _get_results + _inference, rendering.py:15-196, :199-274, :277-494 (no background model, no cascade).
    gate_noise / gate_noise_draw [N * n_samples, E]: --gate_noise in a training forward (coarse pass only here).
    autocast: the reference's mixed-precision training mode (class Autocast); the renderer itself stays fp32 - its inputs are
    (z fp32) x (network outputs), which type promotion lifts to fp32 (under the CPU policy sigma and rgb arrive rounded to 16 bits).
    Points are evaluated in chunks of `chunk` (= model_chunk_size) and the routing (capacity, ranking, l_aux) is per
    chunk, exactly as the reference's loop :354-383.
    fine_samples > 0: the hierarchical pass :236-268 - fine depths drawn from the detached coarse weights (fine_u = the
    U[0,1) tensor of _sample_cdf :608, None -> the deterministic linspace), the network evaluated on them, both sample
    sets sort-merged :419-433 (stable sort here; the reference's torch.sort is only defined up to ties) and composited."""
    near, far = rays[:, 6:7], rays[:, 7:8]
    z = sample_z(near, far, n_samples, perturb, perturb_rand)
    assert gate_noise_draw is None or fine_samples == 0
    out, gl, routes = _eval_points(p, rays, image_indices, z, cfg, chunk, capacity_factor, batch_prioritized, sigma_noise,
                                   routings, hash_cfg, autocast, gate_noise, gate_noise_draw)
    out = out.float()
    comp = composite(out[..., :3], out[..., 3], z)
    res = dict(rgb_coarse=comp["rgb"], depth_variance_coarse=comp["depth_variance"], depth_coarse=comp["depth"],
               weights_coarse=comp["weights"], gate_loss_coarse=gl, sigma_coarse=out[..., 3], raw=out, z_vals=z,
               routings=routes)
    if fine_samples > 0:
        z_mid = 0.5 * (z[:, :-1] + z[:, 1:])                                                 # :238
        z_fine = sample_pdf(z_mid, comp["weights"][:, 1:-1].detach(), fine_samples, fine_u)  # :240
        out_f, gl_f, routes_f = _eval_points(p, rays, image_indices, z_fine, cfg, min(chunk, z_fine.numel()), capacity_factor,
                                             batch_prioritized, sigma_noise_fine, None, hash_cfg, autocast)
        out_f = out_f.float()
        z_all, order = torch.sort(torch.cat([z_fine, z], -1), dim=-1, stable=True)              # :421
        raw_all = torch.gather(torch.cat([out_f, out], 1), 1, order[:, :, None].expand(-1, -1, 4))   # :422-430
        comp_f = composite(raw_all[..., :3], raw_all[..., 3], z_all)
        res.update(rgb_fine=comp_f["rgb"], depth_fine=comp_f["depth"], depth_variance_fine=comp_f["depth_variance"],
                   gate_loss_fine=gl_f, sigma_fine=out_f[..., 3], z_fine=z_fine, z_merged=z_all, order=order,
                   routings_fine=routes_f)
    return res


def training_step(p, rays, image_indices, rgbs, cfg, n_samples, chunk, moe_l_aux_wt=5e-4, **kw):
    """Runner._training_step, runner.py:1077-1123 + loss assembly :646-658:
    loss = mse(rgb, rgbs) + moe_l_aux_wt * gate_loss, gate_loss = mean(gate_loss_coarse), or with a fine pass the
    average of the fine and coarse means (:1104-1111)."""
    res = render_rays(p, rays, image_indices, cfg, n_samples, chunk, **kw)
    typ = "fine" if "rgb_fine" in res else "coarse"                                            # :1094
    photo = F.mse_loss(res[f"rgb_{typ}"], rgbs, reduction="mean")
    gate_loss = res["gate_loss_coarse"].mean()
    if typ == "fine":
        gate_loss = (res["gate_loss_fine"].mean() + gate_loss) / 2
    loss = photo + moe_l_aux_wt * gate_loss
    with torch.no_grad():
        psnr = -10.0 * torch.log10(photo.detach())                                             # metrics.py:8-10
    return dict(loss=loss, photo_loss=photo, gate_loss=gate_loss, psnr=psnr,
                depth_variance=res[f"depth_variance_{typ}"].mean(), results=res)


# --------------------------------------------------------------------------------------------
# dense NeRF (BASELINE configs[0]; also the background network)
# --------------------------------------------------------------------------------------------
def nerf_dense_forward(p, x: torch.Tensor, cfg: dict, sigma_noise: Optional[torch.Tensor] = None):
    """NeRF.forward, models/nerf.py:143-190: x [P, xyz_dim + 3 + 1] = position, direction, image index.  8 x Linear + ReLU
    with the encoded position concatenated again in front of the skip layers, sigma head, then the same direction /
    appearance tail as NeRFMoE."""
    xd = cfg["xyz_dim"]
    enc = positional_encoding(x[:, :xd], cfg["pos_xyz_dim"])
    h = enc
    for i in range(cfg["layers"]):
        if i in cfg["skip_layers"]:
            h = torch.cat([enc, h], -1)                                                       # :155-156
        h = torch.relu(F.linear(h, p[f"xyz_encodings.{i}.0.weight"], p[f"xyz_encodings.{i}.0.bias"]))
    sigma = F.linear(h, p["sigma.weight"], p["sigma.bias"])
    if sigma_noise is not None:
        sigma = sigma + sigma_noise
    sigma = shifted_softplus(sigma)
    h1 = F.linear(h, p["xyz_encoding_final.weight"], p["xyz_encoding_final.bias"])
    feat = torch.cat([h1, positional_encoding(x[:, xd:xd + 3], cfg["pos_dir_dim"]), p["embedding_a.weight"][x[:, -1].long()]], -1)
    h2 = torch.relu(F.linear(feat, p["dir_a_encoding.0.weight"], p["dir_a_encoding.0.bias"]))
    rgb = torch.sigmoid(F.linear(h2, p["rgb.weight"], p["rgb.bias"]))
    return torch.cat([rgb, sigma], -1)


def render_rays_dense(p, rays, image_indices, cfg, n_samples, perturb=0.0, perturb_rand=None, sigma_noise=None):
    """render_rays for a non-MoE model, coarse pass only (rendering.py:15-196, :277-494; BASELINE configs[0])."""
    N = rays.shape[0]
    o, d, near, far = rays[:, 0:3], rays[:, 3:6], rays[:, 6:7], rays[:, 7:8]
    z = sample_z(near, far, n_samples, perturb, perturb_rand)
    xyz = o[:, None, :] + d[:, None, :] * z[:, :, None]
    pts = torch.cat([xyz.reshape(-1, 3), d[:, None, :].expand(N, n_samples, 3).reshape(-1, 3),
                     image_indices.view(N, 1, 1).expand(N, n_samples, 1).reshape(-1, 1).to(xyz.dtype)], 1)
    out = nerf_dense_forward(p, pts, cfg, sigma_noise).view(N, n_samples, 4)
    comp = composite(out[..., :3], out[..., 3], z)
    return dict(rgb_coarse=comp["rgb"], depth_coarse=comp["depth"], depth_variance_coarse=comp["depth_variance"], raw=out, z_vals=z)


# --------------------------------------------------------------------------------------------
# background model + foreground bounds (rendering.py:32-159, 497-570; the Mega-NeRF scenes' default, opts.py:89)
# --------------------------------------------------------------------------------------------
def intersect_sphere(o: torch.Tensor, d: torch.Tensor, center: Optional[torch.Tensor], radius: Optional[torch.Tensor]):
    """_intersect_sphere, rendering.py:497-518: depth at which o + t d leaves the (ellipsoidal) foreground bound."""
    if radius is not None:
        o = (o - center) / radius
        d = d / radius
    d1 = -torch.sum(d * o, -1) / torch.sum(d * d, -1)
    pm = o + d1.unsqueeze(-1) * d
    cos = 1.0 / torch.norm(d, dim=-1)
    pn = torch.sum(pm * pm, -1)
    if (pn >= 1.0).any():
        raise Exception("Not all your cameras are bounded by the unit sphere; please make sure the cameras are normalized properly!")
    return d1 + torch.sqrt(1.0 - pn) * cos


def depth2pts_outside(o: torch.Tensor, d: torch.Tensor, depth: torch.Tensor, center, radius):
    """_depth2pts_outside, rendering.py:521-570 (include_xyz_real False): o, d [N,1,3], depth [N,S] = inverse distance in
    (0,1] -> points on the unit sphere + the inverse distance [N,S,4] (the NeRF++ inverted-sphere parametrisation) and
    the metric depth along the ray [N,S]."""
    if radius is not None:
        o = (o - center) / radius
        d = d / radius
    d1 = -torch.sum(d * o, -1) / torch.sum(d * d, -1)
    p_mid = o + d1.unsqueeze(-1) * d
    p_mid_norm = torch.norm(p_mid, dim=-1)
    cos = 1.0 / d.norm(dim=-1)
    d2 = torch.sqrt(1.0 - p_mid_norm * p_mid_norm) * cos
    p_sphere = o + (d1 + d2).unsqueeze(-1) * d
    axis = torch.cross(o, p_sphere, dim=-1)
    axis = axis / (torch.norm(axis, dim=-1, keepdim=True) + 1e-8)
    phi = torch.asin(p_mid_norm)
    theta = torch.asin(p_mid_norm * depth)
    ang = (phi - theta).unsqueeze(-1)
    p_new = p_sphere * torch.cos(ang) + torch.cross(axis, p_sphere, dim=-1) * torch.sin(ang) + \
        axis * torch.sum(axis * p_sphere, -1, keepdim=True) * (1.0 - torch.cos(ang))             # Rodrigues, :548-550
    p_new = p_new / torch.norm(p_new, dim=-1, keepdim=True)
    depth_real = 1.0 / (depth + 1e-8) * torch.cos(theta) + d1                                   # :554
    return torch.cat((p_new, depth.unsqueeze(-1)), -1), depth_real


def _eval_dense(p_bg, pts, d, image_indices, cfg_bg, sigma_noise=None):
    """The background model on [Nb, S, 4] points (one chunk: nothing depends on the chunking of a dense model)."""
    Nb, S = pts.shape[:2]
    x = torch.cat([pts.reshape(-1, 4), d[:, None, :].expand(Nb, S, 3).reshape(-1, 3),
                   image_indices.view(Nb, 1, 1).expand(Nb, S, 1).reshape(-1, 1).to(pts.dtype)], 1)
    return nerf_dense_forward(p_bg, x, cfg_bg, sigma_noise).view(Nb, S, 4)


def render_rays_bg(p, p_bg, rays, image_indices, cfg, cfg_bg, n_samples: int, chunk: int, center, radius,
                   capacity_factor: float = 1.0, batch_prioritized: bool = True, perturb: float = 0.0,
                   perturb_rand: Optional[torch.Tensor] = None, perturb_rand_bg: Optional[torch.Tensor] = None,
                   sigma_noise=None, sigma_noise_bg=None, fine_samples: int = 0, fine_u=None, fine_u_bg=None,
                   sigma_noise_fine=None, sigma_noise_bg_fine=None):
    """render_rays with a background model, rendering.py:15-196: the foreground network samples [near, min(far, bound)],
    rays that leave the bound (far > fg_far, :36) are continued by the dense 4-D background model on coarse_samples // 2
    inverse-distance samples (:48-78, flip), and both renderings are blended with the foreground's leftover transmittance
    (:104-131).  perturb_rand_bg [Nb, S // 2] are the background's stratified draws (it is sampled first, :50-52).
    fine_samples > 0: both models run the hierarchical pass (:236-268, fine_samples // 2 for the background, :241)."""
    N = rays.shape[0]
    o, d, near, far = rays[:, 0:3], rays[:, 3:6], rays[:, 6:7], rays[:, 7:8]
    fg_far = torch.maximum(intersect_sphere(o, d, center, radius), near.squeeze(-1))             # :34-35
    with_bg = torch.arange(N)[far.squeeze(-1) > fg_far]                                        # :36
    last_delta = 1e10 * torch.ones(N, 1)
    res_bg = None
    Fn = fine_samples
    if with_bg.numel() > 0:
        last_delta[with_bg, 0] = fg_far[with_bg]                                               # :42
        far = torch.minimum(far.squeeze(-1), fg_far).unsqueeze(-1)                             # :44
        Sb = n_samples // 2
        ob, db, ib = o[with_bg], d[with_bg], image_indices[with_bg]
        zb = sample_z(torch.zeros(len(with_bg), 1), torch.ones(len(with_bg), 1), Sb, perturb, perturb_rand_bg)   # :46-49
        pts, depth_real = depth2pts_outside(ob[:, None, :], db[:, None, :], zb, center, radius)
        pts_f, zb_f = torch.flip(pts, dims=[-2]), torch.flip(zb, dims=[-1])                    # :302-304 (depth_real is NOT flipped)
        out_b = _eval_dense(p_bg, pts_f, db, ib, cfg_bg, sigma_noise_bg)
        comp_b = composite(out_b[..., :3], out_b[..., 3], zb_f, 1e10, flip=True, depth_real=depth_real)
        res_bg = dict(rgb=comp_b["rgb"], depth=comp_b["depth"], raw=out_b, z=zb_f, depth_real=depth_real)
        if Fn > 0:
            # _get_results' own z_vals are the UN-flipped depths (only _inference's local copy is flipped, :302-304) while
            # weights_coarse is in the flipped order: the reference pairs ascending bins with the reversed weights
            z_mid = 0.5 * (zb[:, :-1] + zb[:, 1:])                                              # :238
            zf = sample_pdf(z_mid, comp_b["weights"][:, 1:-1].detach(), Fn // 2, fine_u_bg)     # :240-241
            pts2, dreal2 = depth2pts_outside(ob[:, None, :], db[:, None, :], zf, center, radius)
            out_f = _eval_dense(p_bg, pts2, db, ib, cfg_bg, sigma_noise_bg_fine)
            z_all, order = torch.sort(torch.cat([zf, zb_f], -1), dim=-1, descending=True, stable=True)   # :421
            raw_all = torch.gather(torch.cat([out_f, out_b], 1), 1, order[:, :, None].expand(-1, -1, 4))
            dreal_all = torch.gather(torch.cat([dreal2, depth_real], 1), 1, order)              # :432-433
            comp_bf = composite(raw_all[..., :3], raw_all[..., 3], z_all, 1e10, flip=True, depth_real=dreal_all)
            res_bg.update(rgb=comp_bf["rgb"], depth=comp_bf["depth"], z_fine=zf, z_merged=z_all)
    z = sample_z(near, far, n_samples, perturb, perturb_rand)                                   # :85-88
    out, gl, routes = _eval_points(p, rays, image_indices, z, cfg, chunk, capacity_factor, batch_prioritized, sigma_noise, None)

    def bounded(zz):                                                                            # :216-217 / :249-250
        ld = last_delta.clone()
        sel = last_delta.squeeze(-1) < 1e10
        ld[sel, 0] = last_delta[sel, 0] - zz[sel].max(dim=-1)[0]
        return ld
    comp = composite(out[..., :3], out[..., 3], z, bounded(z))
    res = dict(gate_loss_coarse=gl, raw=out, z_vals=z, fg_far=fg_far, with_bg=with_bg, routings=routes, bg=res_bg)
    typ = "coarse"
    if Fn > 0:
        typ = "fine"
        z_mid = 0.5 * (z[:, :-1] + z[:, 1:])
        z_fine = sample_pdf(z_mid, comp["weights"][:, 1:-1].detach(), Fn, fine_u)
        out_f, gl_f, routes_f = _eval_points(p, rays, image_indices, z_fine, cfg, min(chunk, z_fine.numel()), capacity_factor,
                                             batch_prioritized, sigma_noise_fine, None)
        z_all, order = torch.sort(torch.cat([z_fine, z], -1), dim=-1, stable=True)
        raw_all = torch.gather(torch.cat([out_f, out], 1), 1, order[:, :, None].expand(-1, -1, 4))
        comp = composite(raw_all[..., :3], raw_all[..., 3], z_all, bounded(z_fine))             # the FINE depths' maximum, :249-250
        res.update(gate_loss_fine=gl_f, z_fine=z_fine, routings_fine=routes_f)
    rgb, depth = comp["rgb"], comp["depth"]
    res["fg_rgb"], res["bg_lambda"] = rgb, comp["bg_lambda"]
    if res_bg is not None:                                                                      # :104-131
        lam = comp["bg_lambda"][with_bg]
        add = torch.zeros_like(rgb)
        add[with_bg] = res_bg["rgb"] * lam.unsqueeze(-1)
        rgb = rgb + add
        addd = torch.zeros_like(depth)
        addd[with_bg] = res_bg["depth"] * lam.detach()
        depth = depth + addd
    res[f"rgb_{typ}"], res[f"depth_{typ}"], res[f"depth_variance_{typ}"] = rgb, depth, comp["depth_variance"]
    return res


# --------------------------------------------------------------------------------------------
# multiresolution hash-grid encoding (BASELINE configs[4]).  NOT in the reference - PARITY UNPINNED: this restates the
# conventions of include/swn.h (swn_hash_encode_fwd) for Mueller et al. 2022, section 3; it pins the HIP kernels to this file,
# nothing pins this file to the reference.
# --------------------------------------------------------------------------------------------
HASH = dict(n_levels=16, log2_table=19, base_res=16, per_level_scale=1.3819, aabb_lo=(-1.0, -1.0, -1.0), aabb_hi=(1.0, 1.0, 1.0))


def hash_levels(hc: dict):
    """[(scale fp32, grid points per axis, dense?)] per level."""
    T = 1 << hc["log2_table"]
    out = []
    for l in range(hc["n_levels"]):
        s = np.float32(float(hc["base_res"]) * float(np.float32(hc["per_level_scale"])) ** l - 1.0)
        r = int(np.ceil(float(s))) + 2
        out.append((s, r, r ** 3 <= T))
    return out


def hash_encode(x: torch.Tensor, table: torch.Tensor, hc: dict) -> torch.Tensor:
    """x [P,3] world positions, table [L, T, 2] -> [P, 2 L] (trilinear interpolation of hashed / dense grid entries)."""
    lo = torch.tensor(hc["aabb_lo"], dtype=torch.float32)
    inv = (1.0 / (torch.tensor(hc["aabb_hi"], dtype=torch.float32) - lo)).to(torch.float32)
    xn = ((x - lo) * inv).clamp(0.0, 1.0)
    T = 1 << hc["log2_table"]
    feats = []
    for l, (s, r, dense) in enumerate(hash_levels(hc)):
        pos = xn * float(s) + 0.5
        cell = torch.floor(pos)
        w = pos - cell
        c0 = cell.long()
        f = torch.zeros(x.shape[0], 2, dtype=torch.float32)
        for k in range(8):
            dx, dy, dz = k & 1, (k >> 1) & 1, k >> 2
            wk = ((w[:, 0] if dx else 1.0 - w[:, 0]) * (w[:, 1] if dy else 1.0 - w[:, 1])) * (w[:, 2] if dz else 1.0 - w[:, 2])
            cx, cy, cz = c0[:, 0] + dx, c0[:, 1] + dy, c0[:, 2] + dz
            idx = (cx + r * (cy + r * cz)) if dense else ((cx ^ (cy * 2654435761) ^ (cz * 805459861)) & (T - 1))
            f = f + wk[:, None] * table[l][idx]
        feats.append(f)
    return torch.cat(feats, -1)


# --------------------------------------------------------------------------------------------
# mip path (rendering_mip.py, MipNeRFMoE): conical-frustum casting, integrated positional encoding, level resampling
# --------------------------------------------------------------------------------------------
def mip_cast_rays(o: torch.Tensor, d: torch.Tensor, radius: torch.Tensor, t: torch.Tensor):
    """rendering_mip.py:15-25.  o, d [N,3]; radius [N,1]; t [N,S] interval edges -> mean, diagonal covariance [N,S-1,3]
    of the conical frustum between consecutive edges (mip-NeRF eq. 7, the numerically stable form)."""
    t0, t1 = t[:, :-1], t[:, 1:]
    c, hw = (t0 + t1) / 2, (t1 - t0) / 2
    den = 3 * c ** 2 + hw ** 2
    t_mean = c + (2 * c * hw ** 2) / den
    t_var = (hw ** 2) / 3 - (4 / 15) * ((hw ** 4 * (12 * c ** 2 - hw ** 2)) / den ** 2)
    r_var = radius ** 2 * ((c ** 2) / 4 + (5 / 12) * hw ** 2 - (4 / 15) * (hw ** 4) / den)
    mean = o[:, None, :] + d[:, None, :] * t_mean[..., None]
    null_outer_diag = 1 - (d ** 2) / torch.sum(d ** 2, -1, keepdim=True)
    cov = t_var[..., None] * (d ** 2)[:, None, :] + r_var[..., None] * null_outer_diag[:, None, :]
    return mean, cov


def mip_embed(mean: torch.Tensor, cov: torch.Tensor, n_freqs: int) -> torch.Tensor:
    """MipEmbedder, models/nerf.py:28-56 (logscale): [x, sin(2^k x) exp(-4^k var / 2), cos(2^k x) exp(-4^k var / 2), ...]."""
    out = [mean]
    for k in range(n_freqs):
        fy, fw = 2.0 ** k, 4.0 ** k
        damp = torch.exp(-0.5 * fw * cov)
        out += [torch.sin(mean * fy) * damp, torch.cos(mean * fy) * damp]
    return torch.cat(out, -1)


F32_EPS = float(torch.finfo(torch.float32).eps)


def mip_resample(z: torch.Tensor, weights: torch.Tensor, n_fine: int, padding: float, u_rand: Optional[torch.Tensor] = None):
    """The level hand-over of rendering_mip.py:218-229 (blurred, padded weights) + sorted_piecewise_constant_pdf1 (:75-131).
    z [N,S] edges, weights [N,S-1]; u_rand = the U[0,1) tensor [N,n_fine] of the randomized branch (None: deterministic).
    The reference finds the interval with an [N,S,F] mask; cdf is non-decreasing, starts at 0 and ends at 1 > u, so the
    last edge with cdf <= u is searchsorted(right) - 1 and the first with cdf > u is the next one."""
    w = torch.cat([weights[:, :1], weights, weights[:, -1:]], -1)
    wmax = torch.maximum(w[:, :-1], w[:, 1:])
    w = 0.5 * (wmax[:, :-1] + wmax[:, 1:]) + padding
    wsum = w.sum(-1, keepdim=True)
    pad = torch.clamp(1e-5 - wsum, min=0)
    w = w + pad / w.shape[-1]
    wsum = wsum + pad
    pdf = w / wsum
    cdf = torch.clamp(torch.cumsum(pdf[:, :-1], -1), max=1.0)
    cdf = torch.cat([torch.zeros_like(cdf[:, :1]), cdf, torch.ones_like(cdf[:, :1])], -1)
    if u_rand is not None:
        s_ = 1.0 / n_fine
        u = torch.arange(n_fine) * s_ + u_rand * (s_ - F32_EPS)
        u = torch.clamp(u, max=1.0 - F32_EPS)
    else:
        u = torch.linspace(0.0, 1.0 - F32_EPS, n_fine).expand(z.shape[0], n_fine)
    u = u.contiguous()
    i0 = torch.searchsorted(cdf, u, right=True) - 1
    i1 = i0 + 1
    b0, b1 = z.gather(1, i0), z.gather(1, i1)
    c0, c1 = cdf.gather(1, i0), cdf.gather(1, i1)
    t = torch.clamp(torch.nan_to_num((u - c0) / (c1 - c0), 0.0), 0, 1)
    return torch.sort(b0 + t * (b1 - b0), -1)[0]                                              # :223


def render_rays_mip(p, rays, radii, image_indices, cfg, n_samples: int, n_fine: int, chunk: int, capacity_factor: float = 1.0,
                    batch_prioritized: bool = True, perturb: float = 0.0, perturb_rand=None, fine_u=None, sigma_noise=None,
                    sigma_noise_fine=None, rgb_padding: float = 0.001, resample_padding: float = 0.01):
    """rendering_mip.render_rays / _get_results / _inference (rendering_mip.py:133-425) for MipNeRFMoE with direction and
    appearance inputs: n_samples edges -> n_samples - 1 frustums per level; the fine level's edges are resampled from the
    coarse weights (stop_level_grad: detached)."""
    N = rays.shape[0]
    o, d, near, far = rays[:, 0:3], rays[:, 3:6], rays[:, 6:7], rays[:, 7:8]

    def level(z, noise):
        mean, cov = mip_cast_rays(o, d, radii, z)
        S1 = z.shape[1] - 1
        enc = mip_embed(mean, cov, cfg["pos_xyz_dim"]).reshape(N * S1, -1)
        ddir = d[:, None, :].expand(N, S1, 3).reshape(-1, 3)
        idx = image_indices.view(N, 1).expand(N, S1).reshape(-1)
        outs, losses, routes = [], [], []
        for i in range(0, N * S1, chunk):
            r = nerf_moe_forward(p, None, cfg, capacity_factor, batch_prioritized, None if noise is None else noise[i:i + chunk],
                                 encoded=(enc[i:i + chunk], ddir[i:i + chunk], idx[i:i + chunk]))
            outs.append(r["outputs"]); losses.append(r["moe_loss"]); routes.append(r["routing"])
        out = torch.cat(outs, 0).view(N, S1, 4)
        rgbs = out[..., :3] * (1 + 2 * rgb_padding) - rgb_padding                              # :383-384
        zm = 0.5 * (z[:, 1:] + z[:, :-1])                                                        # :386
        comp = composite(rgbs, out[..., 3], zm)
        return comp, torch.cat(losses, 0), routes, out

    z = sample_z(near, far, n_samples, perturb, perturb_rand)
    comp_c, gl_c, routes_c, out_c = level(z, sigma_noise)
    res = dict(rgb_coarse=comp_c["rgb"], gate_loss_coarse=gl_c, routings=routes_c, z_vals=z, weights_coarse=comp_c["weights"])
    if n_fine > 0:
        z_f = mip_resample(z, comp_c["weights"], n_fine, resample_padding, fine_u).detach()     # :218-223 (stop_level_grad)
        comp_f, gl_f, routes_f, out_f = level(z_f, sigma_noise_fine)
        res.update(rgb_fine=comp_f["rgb"], depth_fine=comp_f["depth"], depth_variance_fine=comp_f["depth_variance"],
                   gate_loss_fine=gl_f, routings_fine=routes_f, z_fine=z_f)
    return res


def training_step_mip(p, rays, radii, image_indices, rgbs, cfg, n_samples, n_fine, chunk, moe_l_aux_wt=5e-4, **kw):
    """Runner._training_step_mip, runner.py:1126-1167: loss = (mse(rgb_fine) + mse(rgb_coarse)) / 2 + wt * mean gate losses."""
    res = render_rays_mip(p, rays, radii, image_indices, cfg, n_samples, n_fine, chunk, **kw)
    photo = (F.mse_loss(res["rgb_fine"], rgbs) + F.mse_loss(res["rgb_coarse"], rgbs)) / 2
    gate_loss = (res["gate_loss_fine"].mean() + res["gate_loss_coarse"].mean()) / 2
    return dict(loss=photo + moe_l_aux_wt * gate_loss, photo_loss=photo, gate_loss=gate_loss, results=res)


def adam_step(param: torch.Tensor, grad: torch.Tensor, m: torch.Tensor, v: torch.Tensor, step: int, lr: float,
              beta1=0.9, beta2=0.999, eps=1e-8):
    """torch.optim.Adam (runner.py:486) single-tensor update, no weight decay, no amsgrad."""
    m.mul_(beta1).add_(grad, alpha=1 - beta1)
    v.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    param.addcdiv_(m, denom, value=-lr / bc1)
