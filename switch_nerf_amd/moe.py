"""Stand-alone mirror of the reference's MoE layer on the HIP kernels, usable inside a torch model under autograd.

    moe = MoELayer(gate_type=dict(type="top", k=1, capacity_factor=1.0, batch_prioritized_routing=True, gate_dim=256, ...),
                   model_dim=256, experts=dict(type="expertmlp", count_per_node=8, hidden_size_per_expert=256, layer_num=7,
                                               skips=[3]), ...)
    y = moe(x, gate_input=g)          # y [..., M] in x's dtype; y.l_aux (scalar, differentiable); y.gate_extras["gates"]

Mirrors `moe_layer` / `MOELayer` of /root/reference/switch_nerf/modules/tutel_moe_ext/tutel_moe_layer_nobatch.py (ctor
:443-460, forward :733-797, TopKGate.apply_on_expert_fn :98-235, ExpertMLP :836-924) for the configuration the reference's
NeRFMoE builds (models/nerf_moe.py:278-292): top-k gate (k = 1 in every shipped config; k > 1: _MoETopKFunction) with fp32 router (optionally with gate noise in training), post-score dispatch (the gate value is applied on
the way back), capacity `int(cf * ceil(P / E))` with optional batch-prioritised ranking, `expertmlp` experts with the
residual skip, one routing problem per call (the P tokens of the call), no expert parallelism (`parallel.ExpertParallel`
covers that inside SwitchNeRF).  Parameter names equal the reference's (`gates.0.wg.weight`, `experts.0.weights.{l}`
[E, in, out], `experts.0.bias.{l}` [E, 1, out]) so its state_dict loads unchanged.

SwitchNeRF (model.py) does NOT go through this class: it fuses the layer's combine into the next chain and keeps flat
parameter buffers.  This is the drop-in for code that keeps the reference's own NeRFMoE module and only swaps the layer.
"""
from __future__ import annotations

import math
from typing import Optional

import torch
from torch import nn

from . import ops


class _ExpertParams(nn.Module):
    """Parameter container with the reference ExpertMLP's names and shapes (tutel_moe_layer_nobatch.py:846-885)."""

    def __init__(self, n_experts, model_dim, layer_num, seed_gen=None):
        super().__init__()
        b = 1.0 / math.sqrt(model_dim)
        ws, bs = [], []
        for _ in range(layer_num):
            w = (torch.rand(n_experts, model_dim, model_dim, generator=seed_gen) * 2 - 1) * b      # [E, in, out]
            bb = (torch.rand(n_experts, 1, model_dim, generator=seed_gen) * 2 - 1) * b
            ws.append(nn.Parameter(w))
            bs.append(nn.Parameter(bb))
        self.weights = nn.ParameterList(ws)
        self.bias = nn.ParameterList(bs)


class _Gate(nn.Module):
    def __init__(self, gate_dim, n_experts):
        super().__init__()
        self.wg = nn.Linear(gate_dim, n_experts, bias=False)       # tutel_moe_layer_nobatch.py:73 (fp32 router)


class _MoEFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, layer, x, gate_in, wg, gate_noise, *wb):
        o, dt = ops, layer.dtype
        L, E, M = layer.layer_num, layer.n_experts, layer.model_dim
        P = x.shape[0]
        xs = x.detach().to(dt).contiguous()
        gs = gate_in.detach().to(dt).contiguous()
        wg32 = wg.detach().float().contiguous()
        # gate_noise: the [P, E] draw of a training forward under gate_noise > 0 (tutel_moe_layer_nobatch.py:119-122) or None
        gates, idx, gmax, stats = o.gate_fwd(gs, None, None, wg32, noise=gate_noise, noise_scale=layer._noise_scale if gate_noise is not None else 0.0)
        cap = int(layer.capacity_factor * ((P + E - 1) // E))                                  # tutel_fast_dispatch.py:211
        if layer.moe_no_batch:
            cap = P
        loc, counts, perm, tok2row, l_aux = o.route_top1(idx, gmax, gates, P, E, cap, layer.bpr)
        need_grad = any(ctx.needs_input_grad[1:])
        rows = E * cap
        wf = [o.pack_weights(w.detach().float().contiguous(), dt, True) for w in wb[:L]]
        bias = [b.detach().float().reshape(E, M).contiguous() for b in wb[L:]]
        saves = [torch.empty(rows, M, dtype=dt, device=xs.device) for _ in range(L - 1)] if need_grad else [None] * (L - 1)
        nw = o.chain_mask_words(dt, E, cap, M)
        masks = [torch.empty(nw, dtype=torch.int32, device=xs.device) for _ in range(L - 1)] if need_grad else [None] * (L - 1)
        layers = [o.Layer(wf[l], bias[l], relu=1 if l < L - 1 else 0, skip=(l in layer.skips), save=saves[l] if l < L - 1 else None,
                          mask=masks[l] if l < L - 1 else None) for l in range(L)]
        eo = torch.empty(rows, M, dtype=dt, device=xs.device)
        cnt = counts.view(-1)
        geom = 7 if (M == 256 and dt != torch.float32 and cap >= 256) else 1       # persistent 256-row geometry (forward and backward alike)
        o.mlp_chain(xs, layers, eo, n_groups=E, n_wsets=E, group_stride=cap, group_rows=cnt, group_rows_clamp=cap,
                    x_gather=perm.view(-1), tag=1, geometry=geom)
        y = o.combine_fwd(gmax, idx, loc, eo, cap, P, E, False)                                # decode: gate * row, 0 if dropped
        ctx.layer, ctx.cap, ctx.x_dtype, ctx.g_dtype = layer, cap, x.dtype, gate_in.dtype
        ctx.save_for_backward(xs, gs, wg32, gates, idx, gmax, stats, loc, counts, perm, eo, *wb[:L], *[s for s in saves if s is not None],
                              *[m for m in masks if m is not None])
        ctx.mark_non_differentiable(idx)
        return y.to(x.dtype), l_aux.reshape(()), idx

    @staticmethod
    def backward(ctx, dy, d_laux, _d_idx):
        o, layer = ops, ctx.layer
        dt, L, E, M, cap = layer.dtype, layer.layer_num, layer.n_experts, layer.model_dim, ctx.cap
        sv = ctx.saved_tensors
        xs, gs, wg32, gates, idx, gmax, stats, loc, counts, perm, eo = sv[:11]
        ws = sv[11:11 + L]
        saves = list(sv[11 + L:11 + L + (L - 1)])
        masks = list(sv[11 + L + (L - 1):])
        P = xs.shape[0]
        dev = xs.device
        dy = dy.to(dt).contiguous()
        if d_laux is None:
            d_laux = torch.zeros((), device=dev)
        # decode backward: dL/d gate = <row, dy>, dL/d row = gate * dy (tutel_fast_dispatch.py:48-78)
        dgmax = o.dispatch_bwd_gate(idx, loc, dy, eo, cap)
        dout = o.dispatch_fwd(gmax, idx, loc, dy, E, cap)
        cnt = counts.view(-1)
        wbk = [o.pack_weights(w.detach().float().contiguous(), dt, False) for w in ws]
        dz = [torch.empty(E * cap, M, dtype=dt, device=dev) for _ in range(L - 1)]
        dxr = torch.empty(E * cap, M, dtype=dt, device=dev)
        skip_l = layer.skips[0] if layer.skips else None
        bl = [o.Layer(wbk[l], None, relu=2 if l > 0 else 0, mask=masks[l - 1] if l > 0 else None, save=dz[l - 1] if l > 0 else None)
              for l in range(L - 1, -1, -1)]
        o.mlp_chain(dout, bl, dxr, n_groups=E, n_wsets=E, group_stride=cap, group_rows=cnt, group_rows_clamp=cap,
                    y_add=dz[skip_l] if skip_l is not None else None, tag=2,
                    geometry=7 if (M == 256 and dt != torch.float32 and cap >= 256) else 1)
        dx = o.dispatch_bwd_data(None, idx, loc, dxr, cap)                                     # encode backward (no score)
        # expert weight / bias gradients, all layers in one launch (layer 0 reads its input rows through the permutation)
        dws = [torch.zeros(E, M, M, dtype=torch.float32, device=dev) for _ in range(L)]
        dbs = [torch.zeros(E, M, dtype=torch.float32, device=dev) for _ in range(L)]
        pv = perm.view(-1)
        items = [(xs if l == 0 else saves[l - 1], dout if l == L - 1 else dz[l], dws[l], dbs[l], pv if l == 0 else None, None)
                 for l in range(L)]
        for i0 in range(0, L, 8):
            o.wgrad_batched(items[i0:i0 + 8], n_groups=E, n_wsets=E, group_stride=cap, group_rows=cnt, group_rows_clamp=cap,
                            n_splits=max(1, min(256 // E, cap // 2048)), tag=1)
        # gate backward: softmax / fp32 router, including the load-balance loss term (tutel_fast_dispatch.py:141-150)
        d_wg = torch.zeros_like(wg32)
        coef = (d_laux.reshape(1).float() * (E / float(P * P))).contiguous()
        dg = o.gate_bwd(gs, None, None, wg32, gates, idx, dgmax, stats, counts, coef, P, d_wg, None, None)
        return (None, dx.to(ctx.x_dtype), dg.to(ctx.g_dtype), d_wg, None, *dws, *[b.view(E, 1, M) for b in dbs])


class _MoETopKFunction(torch.autograd.Function):
    """The layer with a top-k gate, k > 1 (extract_critical with top_k > 1, tutel_fast_dispatch.py:176-217; GatingEncoder / GatingDecoder
    looping over the choices, :17-78): every (token, choice) owns one row of the [E, capacity] row space (capacity = k * int(cf *
    ceil(P / E))), the choices' gates are normalised by their sum, l_aux comes from the first choice's mask.  Same kernels as the
    top-1 layer - the expert chains gather their rows through ONE permutation over all choices - plus swn_topk_select / swn_route_topk /
    swn_topk_gate_bwd and the accumulating forms of the sparse kernels."""

    @staticmethod
    def forward(ctx, layer, x, gate_in, wg, gate_noise, clean_noise, *wb):
        o, dt = ops, layer.dtype
        L, E, M, K = layer.layer_num, layer.n_experts, layer.model_dim, layer.top_k
        P = x.shape[0]
        xs = x.detach().to(dt).contiguous()
        gs = gate_in.detach().to(dt).contiguous()
        wg32 = wg.detach().float().contiguous()
        nscale = layer._noise_scale if gate_noise is not None else 0.0
        gates, _idx0, gmax, stats = o.gate_fwd(gs, None, None, wg32, noise=gate_noise, noise_scale=nscale)
        cap = K * int(layer.capacity_factor * ((P + E - 1) // E))                              # tutel_fast_dispatch.py:211
        if layer.moe_no_batch:
            cap = P                                                                            # (k distinct experts per token: <= P rows each)
        idx, _gsel, gn = o.topk_select(gates, K)                                               # :177-182, 204-206
        loc, counts, perm, _, group_rows, l_aux = o.route_topk(idx, gmax, gates, P, E, cap, layer.bpr)
        l_bal = l_aux.reshape(()) if layer.use_load_importance_loss else torch.zeros((), device=xs.device)    # (second output: extras only)
        li = ()
        if layer.use_load_importance_loss:
            # extract_critical_load_importance (:219-265): the layer's loss is load_importance_loss(softmax(logits), the k-th largest noisy
            # logit) (:232); the load-balance term only travels in the extras (compute_balance_loss)
            sigma = layer.gate_noise / E
            logits_w = o.gate_logits(gs, wg32, gate_noise, nscale)
            scores = gates if gate_noise is None else o.gate_fwd(gs, None, None, wg32, noise=clean_noise, noise_scale=1.0 / E if clean_noise is not None else 0.0)[0]
            l_imp, coef_li = o.load_importance_fwd(scores, logits_w, idx[K - 1].contiguous(), sigma)
            l_aux = l_imp
            li = (scores, logits_w, coef_li)
        need_grad = any(ctx.needs_input_grad[1:])
        rows = E * cap
        wf = [o.pack_weights(w.detach().float().contiguous(), dt, True) for w in wb[:L]]
        bias = [b.detach().float().reshape(E, M).contiguous() for b in wb[L:]]
        saves = [torch.empty(rows, M, dtype=dt, device=xs.device) for _ in range(L - 1)] if need_grad else [None] * (L - 1)
        nw = o.chain_mask_words(dt, E, cap, M)
        masks = [torch.empty(nw, dtype=torch.int32, device=xs.device) for _ in range(L - 1)] if need_grad else [None] * (L - 1)
        layers = [o.Layer(wf[l], bias[l], relu=1 if l < L - 1 else 0, skip=(l in layer.skips), save=saves[l] if l < L - 1 else None,
                          mask=masks[l] if l < L - 1 else None) for l in range(L)]
        eo = torch.empty(rows, M, dtype=dt, device=xs.device)
        geom = 7 if (M == 256 and dt != torch.float32 and cap >= 256) else 1
        o.mlp_chain(xs, layers, eo, n_groups=E, n_wsets=E, group_stride=cap, group_rows=group_rows, group_rows_clamp=cap,
                    x_gather=perm.view(-1), tag=1, geometry=geom)
        y = o.combine_fwd(gn[0], idx[0], loc[0], eo, cap, P, E, False)                         # decode, first choice (:59-62)
        for j in range(1, K):
            o.dispatch_bwd_data_more(gn[j], idx[j], loc[j], y, eo, cap)                        # ... `last_result + single_output`
        ctx.layer, ctx.cap, ctx.x_dtype, ctx.g_dtype = layer, cap, x.dtype, gate_in.dtype
        ctx.save_for_backward(xs, gs, wg32, gates, idx, gn, stats, loc, counts, perm, group_rows, eo, *wb[:L],
                              *[s for s in saves if s is not None], *[m for m in masks if m is not None], *li)
        ctx.mark_non_differentiable(idx)
        return y.to(x.dtype), l_aux.reshape(()), l_bal, idx

    @staticmethod
    def backward(ctx, dy, d_laux, d_bal, _d_idx):
        o, layer = ops, ctx.layer
        dt, L, E, M, K, cap = layer.dtype, layer.layer_num, layer.n_experts, layer.model_dim, layer.top_k, ctx.cap
        sv = ctx.saved_tensors
        d_logits_add = None
        if layer.use_load_importance_loss:
            scores, logits_w, coef_li = sv[-3:]
            sv = sv[:-3]
            if d_laux is not None:
                d_logits_add = o.load_importance_bwd(scores, logits_w, sv[4][K - 1].contiguous(), coef_li, d_laux.reshape(1).float().contiguous(),
                                                     layer.gate_noise / E)
            d_laux = d_bal                       # (the load-balance term's gradient arrives through the extras' tensor)
        elif d_bal is not None:
            d_laux = d_bal if d_laux is None else d_laux + d_bal
        xs, gs, wg32, gates, idx, gn, stats, loc, counts, perm, group_rows, eo = sv[:12]
        ws = sv[12:12 + L]
        saves = list(sv[12 + L:12 + L + (L - 1)])
        masks = list(sv[12 + L + (L - 1):])
        P = xs.shape[0]
        dev = xs.device
        dy = dy.to(dt).contiguous()
        if d_laux is None:
            d_laux = torch.zeros((), device=dev)
        # decode backward per choice (tutel_fast_dispatch.py:66-78): dL/d gate_j = <row_j, dy>, dL/d row_j = gate_j * dy
        dgn = torch.stack([o.dispatch_bwd_gate(idx[j], loc[j], dy, eo, cap) for j in range(K)])
        dout = o.dispatch_fwd(gn[0], idx[0], loc[0], dy, E, cap)
        for j in range(1, K):
            o.dispatch_fwd_more(gn[j], idx[j], loc[j], dy, dout, E, cap)
        wbk = [o.pack_weights(w.detach().float().contiguous(), dt, False) for w in ws]
        dz = [torch.empty(E * cap, M, dtype=dt, device=dev) for _ in range(L - 1)]
        dxr = torch.empty(E * cap, M, dtype=dt, device=dev)
        skip_l = layer.skips[0] if layer.skips else None
        bl = [o.Layer(wbk[l], None, relu=2 if l > 0 else 0, mask=masks[l - 1] if l > 0 else None, save=dz[l - 1] if l > 0 else None)
              for l in range(L - 1, -1, -1)]
        o.mlp_chain(dout, bl, dxr, n_groups=E, n_wsets=E, group_stride=cap, group_rows=group_rows, group_rows_clamp=cap,
                    y_add=dz[skip_l] if skip_l is not None else None, tag=2,
                    geometry=7 if (M == 256 and dt != torch.float32 and cap >= 256) else 1)
        dx = o.dispatch_bwd_data(None, idx[0], loc[0], dxr, cap)                               # encode backward (:34-37), summed over the choices
        for j in range(1, K):
            o.dispatch_bwd_data_more(None, idx[j], loc[j], dx, dxr, cap)
        dws = [torch.zeros(E, M, M, dtype=torch.float32, device=dev) for _ in range(L)]
        dbs = [torch.zeros(E, M, dtype=torch.float32, device=dev) for _ in range(L)]
        pv = perm.view(-1)
        items = [(xs if l == 0 else saves[l - 1], dout if l == L - 1 else dz[l], dws[l], dbs[l], pv if l == 0 else None, None)
                 for l in range(L)]
        for i0 in range(0, L, 8):
            o.wgrad_batched(items[i0:i0 + 8], n_groups=E, n_wsets=E, group_stride=cap, group_rows=group_rows, group_rows_clamp=cap,
                            n_splits=max(1, min(256 // E, cap // 2048)), tag=1)
        # gate backward: the normalisation (:204-206), the softmax / fp32 router, the load-balance term of the FIRST choice's mask (:184)
        d_probs = o.topk_gate_bwd(gates, idx, dgn)
        d_wg = torch.zeros_like(wg32)
        coef = (d_laux.reshape(1).float() * (E / float(P * P))).contiguous()
        dg = o.gate_bwd_dense(gs, None, None, wg32, gates, idx[0].contiguous(), None, d_probs, stats, counts[0].contiguous(), coef, P, d_wg,
                              None, None, d_logits_add=d_logits_add)
        return (None, dx.to(ctx.x_dtype), dg.to(ctx.g_dtype), d_wg, None, None, *dws, *[b.view(E, 1, M) for b in dbs])


class MoELayer(nn.Module):
    def __init__(self, gate_type: dict, model_dim: int, experts: dict, scan_expert_func=None, result_func=None, group=None,
                 seeds=None, a2a_ffn_overlap_degree=1, parallel_type="auto", pad_samples=False, moe_no_batch=False,
                 return_gates=False, return_gate_logits=False, dtype=torch.bfloat16):
        super().__init__()
        if gate_type.get("type", "top") != "top":
            raise NotImplementedError("gate type 'top' only (the one the reference's NeRFMoE builds)")
        self.top_k = int(gate_type.get("k", 1))                      # `k` of the model yaml's moe block (every shipped config: 1)
        assert self.top_k > 0, "Top-k value %d is not valid." % self.top_k                     # tutel_moe_layer_nobatch.py:59
        if experts.get("type", "expertmlp") != "expertmlp":
            raise NotImplementedError("expertmlp experts only (seqexperts checkpoints: checkpoint.to_expertmlp)")
        self.model_dim = int(model_dim)
        self.n_experts = int(experts["count_per_node"])
        if self.top_k > self.n_experts:
            raise ValueError("top-k gate: k = %d exceeds the %d experts" % (self.top_k, self.n_experts))
        self.layer_num = int(experts["layer_num"])
        self.skips = [int(s) for s in (experts.get("skips") or [])]
        assert int(experts.get("hidden_size_per_expert", model_dim)) == self.model_dim, "uniform-width expert MLP"
        assert len(self.skips) <= 1
        self.capacity_factor = float(gate_type.get("capacity_factor", 1.0))
        self.bpr = bool(gate_type.get("batch_prioritized_routing", False))
        self.gate_dim = int(gate_type.get("gate_dim", model_dim))
        # gate noise (--gate_noise, opts.py:208; <= 0 = off like the shipped configs' -1): in TRAINING the router's logits get
        # gate_noise * randn / E before the softmax (tutel_moe_layer_nobatch.py:119-122)
        self.gate_noise = float(gate_type.get("gate_noise", 0.0) or 0.0)
        # use_normal_noise (tutel_moe_layer_nobatch.py:116-117): in TRAINING the logits get randn / E - in front of the gate noise; both are
        # additive, so they reach the router kernel as ONE noise operand (swn_gate_fwd_noise)
        self.use_normal_noise = bool(gate_type.get("use_normal_noise", False))
        self._noise_scale = 0.0
        # --use_load_importance_loss (opts.py:210; extract_critical_load_importance, tutel_fast_dispatch.py:219-265): the layer's loss is the
        # load / importance loss of "Scaling Vision with Sparse MoE" instead of the load-balance loss; needs gate_noise > 0 (:154)
        self.use_load_importance_loss = bool(gate_type.get("use_load_importance_loss", False))
        self.compute_balance_loss = bool(gate_type.get("compute_balance_loss", False))
        if self.use_load_importance_loss:
            assert self.gate_noise > 0, "`gate_noise` must be > 0 for normalization in load_importance_loss()."
            if not 2 <= int(experts["count_per_node"]) <= 16:
                raise ValueError("use_load_importance_loss: 2 <= experts <= 16")
        elif self.compute_balance_loss:
            # the reference's layer dies here with a NameError (tutel_moe_layer_nobatch.py:134, :232: `l_balance_loss` only exists in the
            # load-importance branch); refuse at construction instead
            raise ValueError("compute_balance_loss needs use_load_importance_loss (tutel_moe_layer_nobatch.py:128-135, 231-232)")
        self.moe_no_batch, self.return_gates, self.dtype = bool(moe_no_batch), bool(return_gates), dtype
        gen = None
        if seeds is not None:                      # gate under seeds[0], experts under seeds[1] (tutel_moe_layer_nobatch.py:654-703)
            gen = torch.Generator().manual_seed(int(seeds[1]))
        self.gates = nn.ModuleList([_Gate(self.gate_dim, self.n_experts)])
        self.experts = nn.ModuleList([_ExpertParams(self.n_experts, self.model_dim, self.layer_num, gen)])

    def forward(self, input: torch.Tensor, gate_input: Optional[torch.Tensor] = None, gate_noise_draw: Optional[torch.Tensor] = None,
                normal_noise_draw: Optional[torch.Tensor] = None):
        """gate_noise_draw / normal_noise_draw: the [P, E] standard-normal tensors to use as the layer's noise draws (tests replay the
        reference's); None = drawn here (torch.randn on the device, the normal noise first like the reference) when the layer trains
        with gate_noise > 0 / use_normal_noise."""
        if not input.is_cuda:
            raise RuntimeError("MoELayer runs on the HIP library only (no CPU fallback)")
        gi = input if gate_input is None else gate_input
        shape = input.shape
        x = input.reshape(-1, self.model_dim)
        g = gi.reshape(-1, self.gate_dim)
        ex = self.experts[0]
        noise = clean = None
        E = self.n_experts
        draw = lambda given: (given.to(x.device, torch.float32).reshape(-1, E).contiguous() if given is not None
                              else torch.randn(x.shape[0], E, device=x.device, dtype=torch.float32))
        if self.training and self.use_normal_noise:            # logits + n1 / E (+ gate_noise * n2 / E) = logits + (n1 + gate_noise * n2) / E
            noise = clean = draw(normal_noise_draw)    # (clean: the part that belongs to `logits` in the load / importance loss, :116-117)
            if self.gate_noise > 0:
                noise = noise + self.gate_noise * draw(gate_noise_draw)
            self._noise_scale = 1.0 / E
        elif self.training and self.gate_noise > 0:
            noise = draw(gate_noise_draw)
            self._noise_scale = self.gate_noise / E
        l_bal = None
        if self.top_k == 1 and not self.use_load_importance_loss:
            y, l_aux, idx = _MoEFunction.apply(self, x, g, self.gates[0].wg.weight, noise, *ex.weights, *ex.bias)
        else:
            y, l_aux, l_bal, idx = _MoETopKFunction.apply(self, x, g, self.gates[0].wg.weight, noise, clean, *ex.weights, *ex.bias)
        y = y.view(shape)
        y.l_aux = l_aux                                                                         # :792-796
        extras = {}
        if self.return_gates:                                                                   # torch.topk(gates, k).indices, :229
            extras["gates"] = idx.long().view(-1, 1) if idx.dim() == 1 else idx.long().t().contiguous()
        if self.compute_balance_loss:                                                           # :231-232
            extras["balance_loss"] = l_bal
        if extras:
            y.gate_extras = extras
        return y


def moe_layer(*args, **kw):
    """The reference's factory name (tutel_moe_layer_nobatch.py: `moe_layer = MOELayer`)."""
    return MoELayer(*args, **kw)
