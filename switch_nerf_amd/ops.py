"""Torch-tensor wrappers over the C ABI (include/swn.h).  torch is used for device memory and streams only.

Every function enqueues HIP kernels from libswn_hip.so on torch's current stream.  There is no fallback path.
"""
from __future__ import annotations

import os

import ctypes as C
from typing import List, Optional, Sequence

import torch

from . import _lib
from ._lib import BF16, F16, F32, ChainDesc, WgradItem, WgradJob, call


def _code(dtype) -> int:
    """torch dtype -> dtype code of the C ABI.  The 16-bit type must be the one the selected build of the library computes in
    (_lib.use_half): bfloat16 -> libswn_hip.so, float16 -> libswn_hip_f16.so."""
    if dtype == torch.float32:
        return F32
    if dtype == torch.bfloat16:
        if _lib.half_kind() != "bf16":
            raise RuntimeError("bfloat16 tensor, but the fp16 build of the library is selected (_lib.use_half('bf16') first)")
        return BF16
    if dtype == torch.float16:
        if _lib.half_kind() != "f16":
            raise RuntimeError("float16 tensor, but the bf16 build of the library is selected (_lib.use_half('f16') first)")
        return F16
    raise TypeError(f"unsupported dtype {dtype}")


def _dt(t: torch.Tensor) -> int:
    return _code(t.dtype)


def _p(t: Optional[torch.Tensor]):
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), "tensors passed to the HIP library must be contiguous device tensors"
    return C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def torch_dtype(code: int):
    return {F32: torch.float32, BF16: torch.bfloat16, F16: torch.float16}[code]


def mfma_probe() -> torch.Tensor:
    out = torch.zeros(2048, dtype=torch.int32, device="cuda")
    call("swn_mfma_probe", _p(out), _stream())
    return out


def sample_pe(rays, t_steps, perturb_rand, perturb: float, n_samples: int, l_xyz: int, l_dir: int, dtype,
              pe_stride: int, dir_stride: int):
    """-> z [N,S] f32, pe_xyz [N*S, pe_stride] dtype, pe_dir [N, dir_stride] dtype"""
    n = rays.shape[0]
    z = torch.empty(n, n_samples, dtype=torch.float32, device=rays.device)
    pe = torch.empty(n * n_samples, pe_stride, dtype=dtype, device=rays.device)
    pd = torch.empty(n, dir_stride, dtype=dtype, device=rays.device)
    call("swn_sample_pe", _p(rays), _p(t_steps), _p(perturb_rand), float(perturb), n, n_samples, l_xyz, l_dir, _dt(pe),
         _p(z), _p(pe), pe_stride, _p(pd), dir_stride, _stream())
    return z, pe, pd


def pe_from_z(rays, z, l_xyz: int, dtype, pe_stride: int):
    n, S = z.shape
    pe = torch.empty(n * S, pe_stride, dtype=dtype, device=rays.device)
    call("swn_pe_from_z", _p(rays), _p(z), n, S, l_xyz, _dt(pe), _p(pe), pe_stride, _stream())
    return pe


def sample_pdf(z_coarse, weights, u, n_fine: int):
    n, S = z_coarse.shape
    zf = torch.empty(n, n_fine, dtype=torch.float32, device=z_coarse.device)
    call("swn_sample_pdf", _p(z_coarse), _p(weights), _p(u), n, S, n_fine, _p(zf), _stream())
    return zf


def merge_samples(z_fine, z_coarse, raw_fine, raw_coarse):
    n, F = z_fine.shape
    S = z_coarse.shape[1]
    dev = z_fine.device
    z = torch.empty(n, F + S, dtype=torch.float32, device=dev)
    order = torch.empty(n, F + S, dtype=torch.int32, device=dev)
    raw = torch.empty(n * (F + S), 4, dtype=torch.float32, device=dev)
    call("swn_merge_samples", _p(z_fine), _p(z_coarse), _p(raw_fine), _p(raw_coarse), n, F, S, _p(z), _p(order), _p(raw), _stream())
    return z, order, raw


def unmerge_grad(d_raw, order, n_fine: int, n_coarse: int):
    n = order.shape[0]
    dev = d_raw.device
    d_fine = torch.empty(n * n_fine, 4, dtype=torch.float32, device=dev)
    d_coarse = torch.empty(n * n_coarse, 4, dtype=torch.float32, device=dev)
    call("swn_unmerge_grad", _p(d_raw), _p(order), n, n_fine, n_coarse, _p(d_fine), _p(d_coarse), _stream())
    return d_fine, d_coarse


def sign_bits_pack(h):
    """[rows, F] 16-bit -> int32 [rows, F / 32]: bit j of word q = (h[:, 32 q + j] > 0) (include/swn.h swn_sign_bits_pack)."""
    assert h.dim() == 2 and h.is_contiguous() and h.element_size() == 2 and h.shape[1] % 32 == 0
    bits = torch.empty(h.shape[0], h.shape[1] // 32, dtype=torch.int32, device=h.device)
    call("swn_sign_bits_pack", _p(h), int(h.shape[0]), int(h.shape[1]), _p(bits), _stream())
    return bits


def sign_bits_unpack(bits, dtype):
    """int32 [rows, W] -> [rows, 32 W] of `dtype` (the library's 16-bit type): 1 where the bit is set, else 0."""
    assert bits.dim() == 2 and bits.is_contiguous() and bits.dtype == torch.int32
    h = torch.empty(bits.shape[0], bits.shape[1] * 32, dtype=dtype, device=bits.device)
    call("swn_sign_bits_unpack", _p(bits), int(bits.shape[0]), int(h.shape[1]), _p(h), _stream())
    return h


def gather_rows(src, index, out=None):
    """out[r] = src[index[r]] (zero rows for index < 0); src [*, C] row-major, index int32 [R]."""
    R, Cc = index.numel(), src.shape[1]
    if out is None:
        out = torch.empty(R, Cc, dtype=src.dtype, device=src.device)
    if R == 0:
        return out
    call("swn_gather_rows", _p(src), _p(index), R, Cc * src.element_size(), _p(out), _stream())
    return out


def scatter_rows(src, index, out):
    """out[index[r]] = src[r] (index < 0: nowhere; no index twice); src [R, C] row-major, index int32 [R], out [*, C] (include/swn.h
    swn_scatter_rows)."""
    R = index.numel()
    assert src.is_contiguous() and out.is_contiguous() and index.dtype == torch.int32 and src.shape[0] >= R and src.dtype == out.dtype
    rb = (src.numel() // max(1, src.shape[0])) * src.element_size()
    assert rb == (out.numel() // max(1, out.shape[0])) * out.element_size()
    if R:
        call("swn_scatter_rows", _p(src), _p(index), R, rb, _p(out), _stream())
    return out


def owner_aux(gate, noise, index, rows_per_ray: int, ray_base: int, out, zero_gate: bool = False):
    """out[r] = (gate[t], bits of (t // rows_per_ray + ray_base), noise[t] or 0, 0), t = index[r] (include/swn.h swn_owner_aux)."""
    n = index.numel()
    assert out.dtype == torch.float32 and out.is_contiguous() and out.shape[0] >= n and out.shape[1] == 4 and index.dtype == torch.int32
    if n:
        call("swn_owner_aux", _p(gate), _p(noise), _p(index), n, int(rows_per_ray), int(ray_base), int(bool(zero_gate)), _p(out), _stream())
    return out


def owner_aux_split(aux, gate, ray, noise=None):
    """aux [n, 4] -> gate [n] f32, ray [n] int32, noise [n] f32 (or None) (include/swn.h swn_owner_aux_split)."""
    n = aux.shape[0]
    assert aux.dtype == torch.float32 and aux.is_contiguous() and gate.numel() >= n and ray.numel() >= n and ray.dtype == torch.int32
    if n:
        call("swn_owner_aux_split", _p(aux), n, _p(gate), _p(ray), _p(noise), _stream())


def ray_bias_grad_bits(bits, raw, d_raw, w_color, rows_per_ray: int):
    """-> dc_ray [P / rows_per_ray, 32 W] f32: the per-ray sums of dh2 from the sign bits of h2 (include/swn.h swn_ray_bias_grad_bits)."""
    P, W = bits.shape
    assert bits.dtype == torch.int32 and raw.shape == (P, 4) and d_raw.shape == (P, 4) and P % rows_per_ray == 0 and w_color.shape == (3, 32 * W)
    out = torch.empty(P // rows_per_ray, 32 * W, dtype=torch.float32, device=bits.device)
    call("swn_ray_bias_grad_bits", _p(bits), _p(raw), _p(d_raw), _p(w_color), P // rows_per_ray, int(rows_per_ray), 32 * W, _p(out), _stream())
    return out


def gate_fwd(g, ln_w, ln_b, wg, noise=None, noise_scale: float = 0.0):
    """LayerNorm + fp32 router + softmax + top-1 -> (gates [P, E], idx, gmax, stats).  noise [P, E] fp32: logits += noise_scale * noise
    before the softmax (the gate-noise branch of a training forward, swn_gate_fwd_noise)."""
    P, G = g.shape
    E = wg.shape[0]
    dev = g.device
    gates = torch.empty(P, E, dtype=torch.float32, device=dev)
    idx = torch.empty(P, dtype=torch.int32, device=dev)
    gmax = torch.empty(P, dtype=torch.float32, device=dev)
    stats = torch.empty(P, 2, dtype=torch.float32, device=dev)
    if noise is not None:
        assert noise.shape == (P, E) and noise.dtype == torch.float32
        call("swn_gate_fwd_noise", _p(g), _dt(g), _p(ln_w), _p(ln_b), _p(wg), _p(noise.contiguous()), float(noise_scale), P, G, E, _p(gates),
             _p(idx), _p(gmax), _p(stats), _stream())
    else:
        call("swn_gate_fwd", _p(g), _dt(g), _p(ln_w), _p(ln_b), _p(wg), P, G, E, _p(gates), _p(idx), _p(gmax), _p(stats), _stream())
    return gates, idx, gmax, stats


def gate_bwd(g, ln_w, ln_b, wg, gates, idx, d_gmax, stats, counts, laux_coef, seg_tokens, d_wg, d_ln_w, d_ln_b):
    """Accumulates into d_wg / d_ln_w / d_ln_b (fp32), returns dg."""
    P, G = g.shape
    E = wg.shape[0]
    dg = torch.empty_like(g)
    dlogits = torch.empty(int(_lib.load().swn_gate_bwd_scratch_floats(P, G, E)), dtype=torch.float32, device=g.device)
    call("swn_gate_bwd", _p(g), _dt(g), _p(ln_w), _p(ln_b), _p(wg), _p(gates), _p(idx), _p(d_gmax), _p(stats), _p(counts),
         _p(laux_coef), int(seg_tokens), P, G, E, _p(dg), _p(dlogits), _p(d_wg), _p(d_ln_w), _p(d_ln_b), _stream())
    return dg


def gate_bwd_dense(g, ln_w, ln_b, wg, gates, idx, d_gmax, d_probs, stats, counts, laux_coef, seg_tokens, d_wg, d_ln_w, d_ln_b,
                   d_logits_add=None):
    """gate_bwd with dense operands on top (swn_gate_bwd_dense): d_probs [P, E] w.r.t. the probabilities (top-k layers), d_logits_add
    [P, E] w.r.t. the logits, added behind the softmax backward (load / importance loss)."""
    P, G = g.shape
    E = wg.shape[0]
    for t in (d_probs, d_logits_add):
        assert t is None or (t.shape == (P, E) and t.dtype == torch.float32 and t.is_contiguous())
    dg = torch.empty_like(g)
    dlogits = torch.empty(int(_lib.load().swn_gate_bwd_scratch_floats(P, G, E)), dtype=torch.float32, device=g.device)
    call("swn_gate_bwd_dense", _p(g), _dt(g), _p(ln_w), _p(ln_b), _p(wg), _p(gates), _p(idx), _p(d_gmax), _p(d_probs), _p(d_logits_add),
         _p(stats), _p(counts), _p(laux_coef), int(seg_tokens), P, G, E, _p(dg), _p(dlogits), _p(d_wg), _p(d_ln_w), _p(d_ln_b), _stream())
    return dg


def gate_logits(g, wg, noise=None, noise_scale: float = 0.0):
    """The fp32 router's logits g @ wg^T (+ noise_scale * noise) [P, E] (swn_gate_logits; no LayerNorm)."""
    P, G = g.shape
    E = wg.shape[0]
    logits = torch.empty(P, E, dtype=torch.float32, device=g.device)
    call("swn_gate_logits", _p(g), _dt(g), _p(wg), _p(noise), float(noise_scale), P, G, E, _p(logits), _stream())
    return logits


def load_importance_fwd(scores_wo_noise, logits_w_noise, idx_last, sigma: float):
    """load_importance_loss (tutel_fast_dispatch.py:152-174) -> (l_loss [1], coef [2 E] for the backward)."""
    P, E = scores_wo_noise.shape
    dev = scores_wo_noise.device
    l = torch.empty(1, dtype=torch.float32, device=dev)
    coef = torch.empty(2 * E, dtype=torch.float32, device=dev)
    ws = torch.empty(int(_lib.load().swn_load_importance_workspace_floats(P, E)), dtype=torch.float32, device=dev)
    call("swn_load_importance_fwd", _p(scores_wo_noise), _p(logits_w_noise), _p(idx_last), float(sigma), P, E, _p(l), _p(coef), _p(ws), _stream())
    return l, coef


def load_importance_bwd(scores_wo_noise, logits_w_noise, idx_last, coef, d_l, sigma: float):
    """-> d_logits [P, E] of the loss, times the device scalar d_l [1]."""
    P, E = scores_wo_noise.shape
    d_logits = torch.empty(P, E, dtype=torch.float32, device=scores_wo_noise.device)
    call("swn_load_importance_bwd", _p(scores_wo_noise), _p(logits_w_noise), _p(idx_last), _p(coef), _p(d_l), float(sigma), P, E, _p(d_logits),
         _stream())
    return d_logits


def topk_select(gates, k: int):
    """torch.topk(gates, k, dim=1) + the normalised gates of a top-k layer (tutel_fast_dispatch.py:177-182, 204-206)
    -> (idx int32 [k, P], gsel fp32 [k, P], gnorm fp32 [k, P])."""
    P, E = gates.shape
    dev = gates.device
    idx = torch.empty(k, P, dtype=torch.int32, device=dev)
    gsel = torch.empty(k, P, dtype=torch.float32, device=dev)
    gnorm = torch.empty(k, P, dtype=torch.float32, device=dev)
    call("swn_topk_select", _p(gates), P, E, int(k), _p(idx), _p(gsel), _p(gnorm), _stream())
    return idx, gsel, gnorm


def topk_gate_bwd(gates, idx, d_gnorm):
    """Backward of topk_select's normalisation: d_gnorm [k, P] -> d_probs [P, E]."""
    P, E = gates.shape
    k = idx.shape[0]
    d_probs = torch.empty(P, E, dtype=torch.float32, device=gates.device)
    call("swn_topk_gate_bwd", _p(gates), _p(idx), _p(d_gnorm.contiguous()), P, E, int(k), _p(d_probs), _stream())
    return d_probs


def route_topk(idx, gmax, gates, seg_tokens: int, n_experts: int, capacity: int, bpr: bool, want_tok2row=False):
    """Capacity assignment of a top-k routing (swn_route_topk; idx [k, P] from topk_select, gmax [P] the tokens' top-1 gates)
    -> (loc [k, P], counts [k, n_seg, E], perm [n_seg, E * capacity], tok2row [k, P] or None, group_rows [n_seg * E], l_aux [n_seg])."""
    k, P = idx.shape
    n_seg = P // seg_tokens
    dev = idx.device
    loc = torch.empty(k, P, dtype=torch.int32, device=dev)
    counts = torch.empty(k, n_seg, n_experts, dtype=torch.int32, device=dev)
    perm = torch.empty(n_seg, n_experts * capacity, dtype=torch.int32, device=dev)
    tok2row = torch.empty(k, P, dtype=torch.int32, device=dev) if want_tok2row else None
    group_rows = torch.empty(n_seg * n_experts, dtype=torch.int32, device=dev)
    l_aux = torch.empty(n_seg, dtype=torch.float32, device=dev) if gates is not None else None
    nbytes = _lib.load().swn_route_workspace_bytes(P, n_seg, n_experts)
    key = (dev, nbytes)
    ws = _route_ws.get(key)
    if ws is None:
        ws = _route_ws[key] = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    call("swn_route_topk", _p(idx), _p(gmax), _p(gates), P, int(seg_tokens), int(n_experts), int(capacity), int(bool(bpr)), int(k),
         _p(loc), _p(counts), _p(perm), _p(tok2row), _p(group_rows), _p(l_aux), _p(ws), nbytes, _stream())
    return loc, counts, perm, tok2row, group_rows, l_aux


_route_ws = {}


_route_sync = {}


def route_sync(dev):
    """The synchronisation words of the one-launch routing (swn_route_top1x): zero at creation, left zero by every launch; one per
    (device, stream) - routings on one stream are ordered, routings that may overlap get their own.  Never freed (graphs)."""
    k = (dev, torch.cuda.current_stream(dev).cuda_stream if dev.type == "cuda" else 0)
    t = _route_sync.get(k)
    if t is None:
        t = _route_sync[k] = torch.zeros(int(_lib.load().swn_route_sync_bytes()) // 4, dtype=torch.int32, device=dev)
    return t


_ROUTE_MODE = int(os.environ.get("SWN_ROUTE_MODE", "0"))            # (experiment build only: see route_top1)
CHAINQ_STATIC = os.environ.get("SWN_CHAINQ_STATIC") is not None       # persistent chains without tile queues (tests assign it)


def route_top1(idx, gmax, gates, seg_tokens: int, n_experts: int, capacity: int, bpr: bool, want_perm=True, want_drops=False, multi=False,
               mode=None):
    """Top-1 capacity assignment of every routing segment (swn_route_top1x) -> (loc, counts, perm, tok2row, l_aux)
    [+ (drop_begin, dropped) with want_drops: the tokens no expert kept, swn_route_dropped's lists, from the same call].
    mode (include/swn.h): 0 = the 20 per-phase kernels (the product library's only mode); 1 / 2 = the fused forms of round 5, in the
    experiment build only (scripts/experiments/build_route_one.sh; SWN_ROUTE_MODE, read once at import, selects them there).
    multi=True: the round 1-4 entry points themselves (swn_route_top1 + swn_route_dropped) - the twin the tests compare with."""
    P = idx.shape[0]
    n_seg = P // seg_tokens
    dev = idx.device
    loc = torch.empty(P, dtype=torch.int32, device=dev)
    counts = torch.empty(n_seg, n_experts, dtype=torch.int32, device=dev)
    perm = torch.empty(n_seg, n_experts * capacity, dtype=torch.int32, device=dev) if want_perm else None
    tok2row = torch.empty(P, dtype=torch.int32, device=dev)
    l_aux = torch.empty(n_seg, dtype=torch.float32, device=dev) if gates is not None else None
    drop_begin = torch.empty(n_seg * n_experts + 1, dtype=torch.int32, device=dev) if want_drops else None
    dropped = torch.empty(P, dtype=torch.int32, device=dev) if want_drops else None
    nbytes = _lib.load().swn_route_workspace_bytes(P, n_seg, n_experts)
    key = (dev, nbytes)
    ws = _route_ws.get(key)
    if ws is None:      # one workspace per size, kept for good: a captured hipGraph may hold its address (coarse / fine passes alternate
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)      # two sizes; a size follows the batch shape: a handful per process)
        _route_ws[key] = ws
    if multi:
        call("swn_route_top1", _p(idx), _p(gmax), _p(gates), P, int(seg_tokens), n_experts, int(capacity), int(bool(bpr)),
             _p(loc), _p(counts), _p(perm), _p(tok2row), _p(l_aux), _p(ws), nbytes, _stream())
        if want_drops:
            call("swn_route_dropped", _p(idx), _p(loc), _p(counts), P, int(seg_tokens), int(n_experts), int(capacity), _p(drop_begin),
                 _p(dropped), _stream())
    else:
        call("swn_route_top1x", _p(idx), _p(gmax), _p(gates), P, int(seg_tokens), n_experts, int(capacity), int(bool(bpr)),
             _p(loc), _p(counts), _p(perm), _p(tok2row), _p(l_aux), _p(drop_begin), _p(dropped),
             _p(route_sync(dev)), int(_ROUTE_MODE if mode is None else mode), _p(ws), nbytes, _stream())
    if want_drops:
        return loc, counts, perm, tok2row, l_aux, drop_begin, dropped
    return loc, counts, perm, tok2row, l_aux


def route_dropped(idx, loc, counts, seg_tokens: int, n_experts: int, capacity: int):
    """-> (drop_begin int32 [n_groups + 1], dropped int32 [P]): the tokens no expert kept, per (segment, expert) in location order;
    drop_begin[-1] = their number (a device scalar: nothing is read on the host).  Input of the fused tail (mlp_chain(tail=...))."""
    P = idx.shape[0]
    n_groups = (P // seg_tokens) * n_experts
    drop_begin = torch.empty(n_groups + 1, dtype=torch.int32, device=idx.device)
    dropped = torch.empty(P, dtype=torch.int32, device=idx.device)
    call("swn_route_dropped", _p(idx), _p(loc), _p(counts), P, int(seg_tokens), int(n_experts), int(capacity), _p(drop_begin), _p(dropped),
         _stream())
    return drop_begin, dropped


def dispatch_fwd(gates, indices, locations, x, n_experts: int, capacity: int):
    S, H = x.shape
    d = torch.empty(n_experts * capacity, H, dtype=x.dtype, device=x.device)
    call("swn_dispatch_fwd", _p(gates), _p(indices), _p(locations), _p(x), _p(d), _dt(x), S, H, capacity, n_experts, _stream())
    return d


def dispatch_bwd_data(gates, indices, locations, dispatched, capacity: int):
    S = indices.shape[0]
    H = dispatched.shape[1]
    out = torch.empty(S, H, dtype=dispatched.dtype, device=dispatched.device)
    call("swn_dispatch_bwd_data", _p(gates), _p(indices), _p(locations), _p(out), _p(dispatched), _dt(out), S, H, capacity, _stream())
    return out


def dispatch_fwd_more(gates, indices, locations, x, dispatched, n_experts: int, capacity: int):
    """A further choice's rows into an existing dispatched buffer (top-k: tutel_fast_dispatch.py:26-27, later iterations)."""
    S, H = x.shape
    call("swn_dispatch_fwd_more", _p(gates), _p(indices), _p(locations), _p(x), _p(dispatched), _dt(x), S, H, capacity, n_experts, _stream())
    return dispatched


def dispatch_bwd_data_more(gates, indices, locations, out, dispatched, capacity: int):
    """out[i] += g * dispatched[row] (top-k: `last_result + ...`, tutel_fast_dispatch.py:37, :62)."""
    S = indices.shape[0]
    H = dispatched.shape[1]
    call("swn_dispatch_bwd_data_more", _p(gates), _p(indices), _p(locations), _p(out), _p(dispatched), _dt(out), S, H, capacity, _stream())
    return out


def dispatch_bwd_gate(indices, locations, x, dispatched, capacity: int):
    S, H = x.shape
    gg = torch.empty(S, dtype=torch.float32, device=x.device)
    call("swn_dispatch_bwd_gate", _p(gg), _p(indices), _p(locations), _p(x), _p(dispatched), _dt(x), S, H, capacity, _stream())
    return gg


def dispatch_nobatch_fwd(gates, indices, locations, expert_locations_begin, x, dispatched_rows: int, capacity: int = 0):
    """The no-batch encode kernel (tutel_fast_dispatch_nobatch.py:36): D[begin[idx] + loc] = g * x, rows packed per expert."""
    S, H = x.shape
    E = expert_locations_begin.numel()
    d = torch.empty(int(dispatched_rows), H, dtype=x.dtype, device=x.device)
    call("swn_dispatch_nobatch_fwd", _p(gates), _p(indices), _p(locations), _p(expert_locations_begin), _p(x), _p(d), _dt(x), S, H,
         int(capacity), E, int(dispatched_rows), _stream())
    return d


def dispatch_nobatch_bwd_data(gates, indices, locations, expert_locations_begin, dispatched, capacity: int = 0):
    """... :47 / :73 (also the decode forward): out[i] = g * D[begin[idx] + loc], zero rows for idx < 0."""
    S, H = indices.shape[0], dispatched.shape[1]
    out = torch.empty(S, H, dtype=dispatched.dtype, device=dispatched.device)
    call("swn_dispatch_nobatch_bwd_data", _p(gates), _p(indices), _p(locations), _p(expert_locations_begin), _p(out), _p(dispatched),
         _dt(out), S, H, int(capacity), expert_locations_begin.numel(), _stream())
    return out


def dispatch_nobatch_bwd_gate(indices, locations, expert_locations_begin, x, dispatched, capacity: int = 0):
    """... :53 / :93: dgate[i] = <D[begin[idx] + loc], x[i]>."""
    S, H = x.shape
    gg = torch.empty(S, dtype=torch.float32, device=x.device)
    call("swn_dispatch_nobatch_bwd_gate", _p(gg), _p(indices), _p(locations), _p(expert_locations_begin), _p(x), _p(dispatched), _dt(x),
         S, H, int(capacity), expert_locations_begin.numel(), _stream())
    return gg


def route_pack(idx, loc, counts, seg_tokens: int, n_experts: int):
    """Packed (no-batch) row space of a routing -> begin [n_seg * E] (expert_locations_begin per segment), perm [P] row -> token,
    tok2row [P] token -> row."""
    P = idx.shape[0]
    n_groups = (P // seg_tokens) * n_experts
    begin = torch.empty(n_groups, dtype=torch.int32, device=idx.device)
    perm = torch.empty(P, dtype=torch.int32, device=idx.device)
    tok2row = torch.empty(P, dtype=torch.int32, device=idx.device)
    call("swn_route_pack", _p(idx), _p(loc), _p(counts), P, int(seg_tokens), int(n_experts), _p(begin), _p(perm), _p(tok2row), _stream())
    return begin, perm, tok2row


def combine_fwd(gates, indices, locations, expert_out, capacity: int, seg_tokens: int, n_experts: int, relu: bool):
    S = indices.shape[0]
    H = expert_out.shape[1]
    y = torch.empty(S, H, dtype=expert_out.dtype, device=expert_out.device)
    call("swn_combine_fwd", _p(gates), _p(indices), _p(locations), _p(y), _p(expert_out), _dt(y), S, H, capacity,
         int(seg_tokens), n_experts, int(bool(relu)), _stream())
    return y


def combine_bwd(dy_in, y, dsig, wsig, gate):
    S, H = y.shape
    dout = torch.empty_like(y)
    dgate = torch.empty(S, dtype=torch.float32, device=y.device)
    call("swn_combine_bwd", _p(dy_in), _p(y), _p(dsig), _p(wsig), _p(gate), _dt(y), S, H, _p(dout), _p(dgate), _stream())
    return dout, dgate


def heads_fwd(y, h2, w_sigma, b_sigma, w_color, b_color, sigma_noise):
    P, M = y.shape
    H2 = h2.shape[1]
    raw = torch.empty(P, 4, dtype=torch.float32, device=y.device)
    call("swn_heads_fwd", _p(y), _p(h2), _dt(y), _p(w_sigma), _p(b_sigma), _p(w_color), _p(b_color), _p(sigma_noise), P, M, H2,
         _p(raw), _stream())
    return raw


def heads_bwd(y, h2, w_color, raw, d_raw, d_w_sigma, d_b_sigma, d_w_color, d_b_color, rows_per_group: int = 0):
    """-> (dh2, dsig) or, with rows_per_group > 0 (the samples per ray, dividing P), (dh2, dsig, colsum [P / rows_per_group, H2] f32 =
    group_colsum(dh2, rows_per_group) from the same launch)."""
    P, H2 = h2.shape
    M = d_w_sigma.numel()          # (y may be None: the sigma weight gradient comes from the fused backward chain, mlp_chain(combine=(..., dws)))
    dh2 = torch.empty_like(h2)
    dsig = torch.empty(P, dtype=torch.float32, device=h2.device)
    nb = int(_lib.load().swn_heads_bwd_workspace_bytes(int(P), int(M), int(H2)))
    ws = torch.empty(max(nb, 4) // 4, dtype=torch.float32, device=h2.device)     # block partial sums (added in a fixed order)
    cs = torch.empty(P // rows_per_group, H2, dtype=torch.float32, device=h2.device) if rows_per_group else None
    call("swn_heads_bwd", _p(y), _p(h2), _dt(h2), _p(w_color), _p(raw), _p(d_raw), P, M, H2, _p(dh2), _p(dsig), _p(d_w_sigma),
         _p(d_b_sigma), _p(d_w_color), _p(d_b_color), int(rows_per_group), _p(cs), _p(ws), nb, _stream())
    return (dh2, dsig, cs) if rows_per_group else (dh2, dsig)


def group_colsum(x, rows_per_group: int):
    R, Cc = x.shape
    g = R // rows_per_group
    out = torch.empty(g, Cc, dtype=torch.float32, device=x.device)
    call("swn_group_colsum", _p(x), _dt(x), g, rows_per_group, Cc, _p(out), _stream())
    return out


def _idx_arg(image_indices):
    assert image_indices.dtype in (torch.int32, torch.int64) and image_indices.is_contiguous()
    return _p(image_indices), int(image_indices.dtype == torch.int64)


def ray_feat_fwd(pe_dir, in_dir: int, emb, image_indices, w2r, b2):
    """-> feat [N, in_dir + app_dim] f32, c_ray [N, h2] f32 (include/swn.h swn_ray_feat_fwd)."""
    N, h2, app = pe_dir.shape[0], w2r.shape[1], emb.shape[1]
    feat = torch.empty(N, in_dir + app, dtype=torch.float32, device=pe_dir.device)
    c_ray = torch.empty(N, h2, dtype=torch.float32, device=pe_dir.device)
    ip, i64 = _idx_arg(image_indices)
    call("swn_ray_feat_fwd", _p(pe_dir), _dt(pe_dir), pe_dir.shape[1], int(in_dir), _p(emb), app, ip, i64, _p(w2r), _p(b2), N, h2,
         _p(feat), _p(c_ray), _stream())
    return feat, c_ray


def emb_grad(d_feat, image_indices, d_emb):
    """d_emb[image_indices[n]] += d_feat[n], rays in ascending order with a fixed association (deterministic nn.Embedding backward)."""
    assert d_feat.dtype == torch.float32 and d_emb.dtype == torch.float32 and d_feat.stride(1) == 1 and d_emb.is_contiguous()
    ip, i64 = _idx_arg(image_indices)
    call("swn_emb_grad", _p(d_feat), d_feat.stride(0), ip, i64, d_feat.shape[0], d_feat.shape[1], d_emb.shape[0], _p(d_emb), _stream())


def ray_feat_wgrad(feat, dc_ray, d_w2r, d_b2):
    """d_w2r [F, H2] += feat^T dc_ray, d_b2 [H2] += dc_ray.sum(0) (fp32): block partial sums added in a fixed order, one launch + reduce."""
    N, F = feat.shape
    H2 = dc_ray.shape[1]
    assert all(t.dtype == torch.float32 and t.is_contiguous() for t in (feat, dc_ray, d_w2r, d_b2)) and dc_ray.shape[0] == N
    assert d_w2r.numel() == F * H2 and d_b2.numel() == H2
    nb = int(_lib.load().swn_ray_feat_wgrad_workspace_bytes(int(N), int(F), int(H2)))
    ws = torch.empty(max(nb, 4) // 4, dtype=torch.float32, device=feat.device)
    call("swn_ray_feat_wgrad", _p(feat), _p(dc_ray), N, F, H2, _p(d_w2r), _p(d_b2), _p(ws), nb, _stream())


def step_loss(rgb, target, l_aux_a, l_aux_b, wt: float, loss_scale_dev=None):
    """-> (out4 = [photo, gate_loss, loss, psnr] on the device, d_rgb, d_l_aux_a, d_l_aux_b or None)."""
    dev = rgb.device
    d_rgb = torch.empty_like(rgb)
    d_a = torch.empty_like(l_aux_a)
    d_b = torch.empty_like(l_aux_b) if l_aux_b is not None else None
    out4 = torch.empty(4, dtype=torch.float32, device=dev)
    call("swn_step_loss", _p(rgb), _p(target), rgb.numel(), _p(l_aux_a), l_aux_a.numel(), _p(l_aux_b), 0 if l_aux_b is None else l_aux_b.numel(),
         float(wt), _p(loss_scale_dev), _p(d_rgb), _p(d_a), _p(d_b), _p(out4), _stream())
    return out4, d_rgb, d_a, d_b


def sample_z(rays, t_steps, perturb_rand, perturb: float, n_samples: int):
    n = rays.shape[0]
    z = torch.empty(n, n_samples, dtype=torch.float32, device=rays.device)
    call("swn_sample_z", _p(rays), _p(t_steps), _p(perturb_rand), float(perturb), n, n_samples, _p(z), _stream())
    return z


def _hash_cfg(hc: dict):
    c = _lib.HashCfg()
    c.n_levels, c.log2_table, c.base_res = int(hc["n_levels"]), int(hc["log2_table"]), int(hc["base_res"])
    c.per_level_scale = float(hc["per_level_scale"])
    for i in range(3):
        c.aabb_lo[i], c.aabb_hi[i] = float(hc["aabb_lo"][i]), float(hc["aabb_hi"][i])
    return c


def hash_encode_fwd(rays, z, table, hc: dict, dtype, out_stride: int):
    """Multiresolution hash-grid encoding of the points o + d z (include/swn.h swn_hash_encode_fwd) -> [N * S, out_stride]."""
    n, S = z.shape
    out = torch.empty(n * S, out_stride, dtype=dtype, device=rays.device)
    call("swn_hash_encode_fwd", _p(rays), _p(z), n, S, C.byref(_hash_cfg(hc)), _p(table), _code(dtype),
         _p(out), int(out_stride), _stream())
    return out


_hash_xcd_tables = {}
HASH_XCD_COPIES = 8          # XCDs of an MI300X / MI355X (the kernel picks its copy by HW_REG_XCC_ID & 7: hashgrid.hip)


HASH_XCD_MB = int(os.environ.get("SWN_HASH_XCD_MB", "1024"))      # (read once; tests assign ops.HASH_XCD_MB)


def hash_xcd_budget_bytes() -> int:
    """Upper bound on the per-XCD gradient-table workspace of hash_encode_bwd (SWN_HASH_XCD_MB, default 1024 MiB; 0 = never)."""
    return HASH_XCD_MB << 20


HASH_BWD_MODE = os.environ.get("SWN_HASH_BWD", "binned")      # "binned" (default) | "atomic": read once; tests assign ops.HASH_BWD_MODE
_hash_bin_ws = {}


def hash_encode_bwd(rays, z, d_out, hc: dict, d_table):
    """d_table [L, T, 2] fp32 += the table gradient for dL/d encoding d_out [N * S, stride].

    Default ("binned", swn_hash_encode_bwd_binned): no global float atomics - the contributions are binned by table tile into a workspace
    (12 bytes per (point, level, corner): 3.2 GB at 2M points x 16 levels, kept per size: a captured hipGraph holds its address) and added
    per tile in fixed point in LDS: bit-deterministic, ~4 x faster than the atomics at the recipe's size (profiles/r06_experiments.md 3).
    Tables of more than 2^22 entries per level, or HASH_BWD_MODE = "atomic": the atomic kernels - one private fp32 copy of the gradient
    table per XCD (HASH_XCD_COPIES x the table: 512 MiB at 16 levels x 2^19, zeroed once, left zero by every launch, never freed) so
    that the atomics of an XCD stay in its own L2; a table whose copies would exceed hash_xcd_budget_bytes() takes the plain path:
    atomics straight into d_table."""
    n, S = z.shape
    if HASH_BWD_MODE == "binned" and int(hc["log2_table"]) <= 22 and n * S * 8 * int(hc["n_levels"]) < (1 << 32):
        cfg = _hash_cfg(hc)
        nbytes = int(_lib.load().swn_hash_bwd_workspace_bytes(n * S, C.byref(cfg)))
        key = (d_table.device, nbytes)
        ws = _hash_bin_ws.get(key)
        if ws is None:
            ws = _hash_bin_ws[key] = torch.empty(nbytes, dtype=torch.uint8, device=d_table.device)
        call("swn_hash_encode_bwd_binned", _p(rays), _p(z), n, S, C.byref(cfg), _p(d_out), _dt(d_out), d_out.shape[1], _p(d_table), _p(ws),
             nbytes, _stream())
        return
    ws_bytes = HASH_XCD_COPIES * d_table.numel() * 4
    if ws_bytes > hash_xcd_budget_bytes():
        call("swn_hash_encode_bwd", _p(rays), _p(z), n, S, C.byref(_hash_cfg(hc)), _p(d_out), _dt(d_out), d_out.shape[1], _p(d_table), _stream())
        return
    key = (d_table.device, d_table.numel())
    ws = _hash_xcd_tables.get(key)
    if ws is None:
        ws = _hash_xcd_tables[key] = torch.zeros(HASH_XCD_COPIES * d_table.numel(), dtype=torch.float32, device=d_table.device)
    call("swn_hash_encode_bwd_xcd", _p(rays), _p(z), n, S, C.byref(_hash_cfg(hc)), _p(d_out), _dt(d_out), d_out.shape[1], _p(d_table), _p(ws),
         _stream())


def mip_encode(rays, radii, z, l_xyz: int, dtype, pe_stride: int):
    """-> integrated positional encoding of the n_edges - 1 frustums per ray: [N * (S - 1), pe_stride] dtype"""
    n, S = z.shape
    pe = torch.empty(n * (S - 1), pe_stride, dtype=dtype, device=rays.device)
    call("swn_mip_encode", _p(rays), _p(radii), _p(z), n, S, l_xyz, _dt(pe), _p(pe), pe_stride, _stream())
    return pe


def mip_resample(z, weights, u_rand, n_fine: int, padding: float = 0.01):
    n, S = z.shape
    zf = torch.empty(n, n_fine, dtype=torch.float32, device=z.device)
    call("swn_mip_resample", _p(z), _p(weights), _p(u_rand), float(padding), n, S, n_fine, _p(zf), _stream())
    return zf


def composite_fwd(raw, z, last_delta=1e10, want_weights=False, rgb_padding=0.0):
    N, S = z.shape
    dev = z.device
    rgb = torch.empty(N, 3, dtype=torch.float32, device=dev)
    depth = torch.empty(N, dtype=torch.float32, device=dev)
    dvar = torch.empty(N, dtype=torch.float32, device=dev)
    w = torch.empty(N, S, dtype=torch.float32, device=dev) if want_weights else None
    call("swn_composite_fwd", _p(raw), _p(z), float(last_delta), float(rgb_padding), N, S, _p(rgb), _p(depth), _p(dvar), _p(w), _stream())
    return rgb, depth, dvar, w


def composite_bwd(raw, z, d_rgb, last_delta=1e10, rgb_padding=0.0):
    N, S = z.shape
    d_raw = torch.empty(N * S, 4, dtype=torch.float32, device=z.device)
    call("swn_composite_bwd", _p(raw), _p(z), float(last_delta), float(rgb_padding), _p(d_rgb), N, S, _p(d_raw), _stream())
    return d_raw


def _host3(v):
    """3 floats in host memory (sphere centre / radii) as a ctypes array; None stays None."""
    if v is None:
        return None
    arr = v.detach().cpu().numpy() if torch.is_tensor(v) else v
    vals = [float(x) for x in arr]
    assert len(vals) == 3
    return (C.c_float * 3)(*vals)


def fg_bounds(rays, center, radius):
    """render_rays' foreground bound (rendering.py:32-44, 497-518) -> rays with the clipped far plane, fg_far [N],
    last_delta [N] (fg_far for rays that continue into the background, 1e10 otherwise), has_bg [N] int32.
    Raises like the reference when a ray's closest approach to the centre is outside the bound."""
    N = rays.shape[0]
    dev = rays.device
    rays_fg = torch.empty_like(rays)
    fg_far = torch.empty(N, dtype=torch.float32, device=dev)
    last_delta = torch.empty(N, dtype=torch.float32, device=dev)
    has_bg = torch.empty(N, dtype=torch.int32, device=dev)
    n_out = torch.zeros(1, dtype=torch.int32, device=dev)
    call("swn_fg_bounds", _p(rays), _host3(center), _host3(radius), N, _p(rays_fg), _p(fg_far), _p(last_delta), _p(has_bg), _p(n_out),
         _stream())
    return rays_fg, fg_far, last_delta, has_bg, n_out


def bg_sample_pe(rays, center, radius, n_samples, l_xyz, dtype, pe_stride, perturb_rand=None, perturb=0.0, z_in=None, pe_out=None):
    """Background samples of the (gathered) rays: see swn_bg_sample_pe in include/swn.h.  -> z [N,S], depth_real [N,S], pe."""
    N = rays.shape[0]
    dev = rays.device
    S = n_samples if z_in is None else z_in.shape[1]
    z = torch.empty(N, S, dtype=torch.float32, device=dev) if z_in is None else z_in
    dreal = torch.empty(N, S, dtype=torch.float32, device=dev)
    pe = pe_out if pe_out is not None else torch.empty(N * S, pe_stride, dtype=dtype, device=dev)
    t_steps = torch.linspace(0, 1, S, dtype=torch.float32).to(dev) if z_in is None else None
    call("swn_bg_sample_pe", _p(rays), _host3(center), _host3(radius), _p(t_steps), _p(perturb_rand), float(perturb), N, S, int(l_xyz),
         _code(dtype), _p(z_in), _p(z) if z_in is None else None, _p(dreal), _p(pe), int(pe_stride), _stream())
    return z, dreal, pe


def composite_bounded_fwd(raw, z, last_delta=None, flip=False, depth_real=None, want_weights=False, want_bg_lambda=False):
    """Compositing with a per-ray last delta / descending depths / metric depth source / leftover transmittance
    (rendering.py:435-494 with last_delta, flip, depth_real, get_bg_lambda) -> rgb, depth, depth_variance, weights, bg_lambda."""
    N, S = z.shape
    dev = z.device
    rgb = torch.empty(N, 3, dtype=torch.float32, device=dev)
    depth = torch.empty(N, dtype=torch.float32, device=dev)
    dvar = torch.empty(N, dtype=torch.float32, device=dev)
    w = torch.empty(N, S, dtype=torch.float32, device=dev) if want_weights else None
    lam = torch.empty(N, dtype=torch.float32, device=dev) if want_bg_lambda else None
    call("swn_composite_bounded_fwd", _p(raw), _p(z), _p(last_delta), int(bool(flip)), _p(depth_real), N, S, _p(rgb), _p(depth),
         _p(dvar), _p(w), _p(lam), _stream())
    return rgb, depth, dvar, w, lam


def composite_bounded_bwd(raw, z, d_rgb, last_delta=None, flip=False, d_bg_lambda=None):
    N, S = z.shape
    d_raw = torch.empty(N * S, 4, dtype=torch.float32, device=z.device)
    call("swn_composite_bounded_bwd", _p(raw), _p(z), _p(last_delta), int(bool(flip)), _p(d_rgb), _p(d_bg_lambda), N, S, _p(d_raw),
         _stream())
    return d_raw


def pack_weights(master, dtype, transpose: bool):
    """master [n_wsets, in, out] fp32 -> packed compute copy for mlp_chain, tagged with its logical (n, k)."""
    assert master.dim() == 3 and master.dtype == torch.float32
    ws, i, o_ = master.shape
    out = torch.empty(ws * i * o_, dtype=dtype, device=master.device)
    call("swn_pack_weights", _p(master), _p(out), _dt(out), ws, i, o_, int(bool(transpose)), _stream())
    out.swn_nk = (o_, i) if transpose else (i, o_)
    return out


def repack_weights(master, packed, transpose: bool):
    ws, i, o_ = master.shape
    call("swn_pack_weights", _p(master), _p(packed), _dt(packed), ws, i, o_, int(bool(transpose)), _stream())
    return packed


def repack_weights_batched(pairs):
    """pairs: (master [ws, in, out] f32, packed, transpose[, in_padded[, out_padded]]) of one compute dtype -> one launch per 32 weights.
    in_padded / out_padded: the packed copy's in / out dimension when larger than the master's (zero-padded: pack_weights_padded)."""
    from ._lib import PackItem
    for i0 in range(0, len(pairs), 32):
        chunk = pairs[i0:i0 + 32]
        arr = (PackItem * len(chunk))()
        for it, pr in zip(arr, chunk):
            master, packed, transpose = pr[:3]
            ws, i, o_ = master.shape
            ipad = int(pr[3]) if len(pr) > 3 and pr[3] else i
            opad = int(pr[4]) if len(pr) > 4 and pr[4] else o_
            it.master, it.out, it.n_wsets, it.in_dim, it.out_dim, it.transpose = _p(master), _p(packed), ws, ipad, opad, int(bool(transpose))
            it.in_rows, it.out_cols = (i if ipad != i else 0), (o_ if opad != o_ else 0)
        call("swn_pack_weights_batched", arr, len(chunk), _dt(chunk[0][1]), _stream())


def pack_weights_padded(master, dtype, transpose: bool, in_padded: int = 0, out_padded: int = 0):
    """pack_weights with the master's in / out dimension zero-padded: a 128-feature first layer for the K = 256 kernels of chain
    geometries 6 / 7 (mlp_chain(..., x_features=128)) - forward (transpose): in 128 -> 256; backward-data of a 128-output layer: out 128 ->
    256.  Refresh with repack_weights_batched([(master, packed, transpose, in_padded, out_padded)])."""
    assert master.dim() == 3 and master.dtype == torch.float32
    ws, i, o_ = master.shape
    ipad, opad = in_padded or i, out_padded or o_
    assert ipad >= i and opad >= o_
    out = torch.empty(ws * ipad * opad, dtype=dtype, device=master.device)
    out.swn_nk = (opad, ipad) if transpose else (ipad, opad)
    repack_weights_batched([(master, out, transpose, ipad, opad)])
    return out


class Layer:
    """One Linear of a chain: w = pack_weights(...) output (carries .swn_nk = (N, K)), b [n_wsets, N] f32 or None."""

    def __init__(self, w, b=None, relu=0, skip=False, save=None, mask=None, rowbias=None, rows_per_bias=0):
        self.w, self.b, self.relu, self.skip, self.save, self.mask = w, b, int(relu), int(skip), save, mask      # skip: 0 / 1 residual / 2 concat half
        self.rowbias, self.rows_per_bias = rowbias, int(rows_per_bias)


def chain_tile_rows(dtype) -> int:
    return _lib.load().swn_chain_tile_rows(_code(dtype))


def chain_mask_words(dtype, n_groups: int, group_stride: int, max_width: int = 256) -> int:
    """uint32 words per layer mask buffer for a chain launch with this geometry (1 bit per row x feature of the kernel's tile;
    max_width = the widest layer of the chain: > 256 selects the 512-feature kernels)."""
    return int(_lib.load().swn_chain_mask_words(_code(dtype), int(n_groups), int(group_stride), int(max_width)))


def mlp_chain(x, layers: Sequence[Layer], y, n_groups=1, n_wsets=1, group_stride=None, group_rows=None,
              group_rows_clamp=None, x_gather=None, x_save=None, y_add=None, y_add_gather=None, tag=0, x_scale=None,
              x_relu=False, geometry=0, group_begin=None, combine=None, heads=None, sched=None, x_features=0, tail=None, head=None):
    """geometry: 0 / 1 the 64-row tile kernels, 2 - 5 the chain_big.hip geometries (include/swn.h).  group_begin: first row of every
    group (packed / no-batch layout) instead of g * group_stride.  combine = (y_fwd, dsig, wsig, gate, dgate_out): the combine backward
    (ops.combine_bwd) fused into the write-out of the last layer.  heads = (w_sigma, b_sigma, w_color, b_color, sigma_noise or None, raw):
    the sigma / colour heads (ops.heads_fwd) fused into the tail forward chain (tag 4) - y may then be None (nothing but raw is written).
    sched: int32 [16] zero-initialised tile-queue counters of the persistent geometries 6 / 7 (chain_sched(); left zero by the kernel).
    tail = (tail_first, gate [P] f32, drop_begin, dropped, y_features[, bias_row int32 [P]]): the dense tail folded into the expert forward chain (include/swn.h,
    tail_first: geometry 7, tag 7) - layers[tail_first:] are shared layers, the saves from layer tail_first - 1 on, y and the heads'
    raw are in token order (P rows), x_gather maps rows to tokens.
    head = (head_layers, drop_begin, dropped): the mirror image for the backward pass (include/swn.h, head_layers: geometry 7, tag 8) -
    x = dh2 in token order, layers[:head_layers] shared, `combine` (token order) applied behind them, the expert backward layers after."""
    d = ChainDesc()
    d.dtype = _dt(x)
    d.tag = int(tag)
    d.geometry = int(geometry)
    d.n_layers = len(layers)
    d.n_groups, d.n_wsets = int(n_groups), int(n_wsets)
    d.group_stride = int(group_stride if group_stride is not None else (y if y is not None else heads[5]).shape[0])
    if tail is not None:
        t_first, t_gate, t_begin, t_dropped, t_yf = tail[:5]
        if len(tail) > 5 and tail[5] is not None:      # token -> row of the last layer's rowbias (a token space that is not in ray order)
            assert tail[5].dtype == torch.int32 and tail[5].is_contiguous() and tail[5].numel() >= t_gate.numel()
            d.tail_bias_row = _p(tail[5])
        assert t_gate.dtype == torch.float32 and t_begin.dtype == torch.int32 and t_dropped.dtype == torch.int32 and x_gather is not None
        d.tail_first, d.y_features, d.tail_gate, d.tail_dropped = int(t_first), int(t_yf), _p(t_gate), _p(t_dropped)
        d.tail_n_dropped = t_begin.data_ptr() + 4 * (t_begin.numel() - 1)
        d.tail_dropped_max, d.tail_tokens = int(t_dropped.numel()), int(t_gate.numel())
    if head is not None:
        h_layers, h_begin, h_dropped = head
        assert h_begin.dtype == torch.int32 and h_dropped.dtype == torch.int32 and x_gather is not None and combine is not None
        d.head_layers, d.tail_dropped = int(h_layers), _p(h_dropped)
        d.tail_n_dropped = h_begin.data_ptr() + 4 * (h_begin.numel() - 1)
        d.tail_dropped_max, d.tail_tokens = int(h_dropped.numel()), int(x.shape[0])
    d.group_rows = _p(group_rows)
    d.group_rows_clamp = int(group_rows_clamp if group_rows_clamp is not None else d.group_stride)
    d.group_begin = _p(group_begin)
    d.x, d.x_gather, d.x_save, d.y = _p(x), _p(x_gather), _p(x_save), _p(y)
    d.x_scale, d.x_relu = _p(x_scale), int(bool(x_relu))
    d.y_add, d.y_add_gather = _p(y_add), _p(y_add_gather)
    d.x_features = int(x_features)                 # (geometry 6 / 7: 128-feature rows under a zero-padded K = 256 first layer)
    if sched is None and d.geometry >= 6 and not CHAINQ_STATIC:
        sched = chain_sched(x.device, d.tag)       # (launches of one role on one stream are ordered: they can share the counters)
    if sched is not None:
        assert sched.dtype == torch.int32 and sched.numel() >= 16
        d.sched = _p(sched)
    if combine is not None:
        cy, cds, cws, cg, cdg = combine[:5]
        assert cy.dtype == x.dtype and (cy.shape == y.shape or head is not None) and cg.dtype == torch.float32 and cdg.dtype == torch.float32
        d.comb_y, d.comb_dsig, d.comb_wsig, d.comb_gate, d.comb_dgate = _p(cy), _p(cds), _p(cws), _p(cg), _p(cdg)
        if len(combine) > 5 and combine[5] is not None:      # the sigma head's weight gradient += from the same pass (fused backward only)
            dws = combine[5]
            assert head is not None and dws.dtype == torch.float32 and dws.numel() == 256 and dws.is_contiguous()
            nb = int(_lib.load().swn_chain_dwsig_workspace_bytes(int(n_groups), int(d.group_rows_clamp)))
            ws = _dwsig_ws.get((x.device, nb))
            if ws is None:           # per-wave partial sums (zeroed and added up in a fixed order by the launch); never freed (graphs)
                ws = _dwsig_ws[(x.device, nb)] = torch.empty(nb // 4, dtype=torch.float32, device=x.device)
            d.comb_dwsig, d.comb_dwsig_ws = _p(dws), _p(ws)
    if heads is not None:
        hws, hbs, hwc, hbc, hnoise, hraw = heads
        assert all(t.dtype == torch.float32 and t.is_contiguous() for t in (hws, hbs, hwc, hbc, hraw)) and hraw.shape[1] == 4
        assert hnoise is None or (hnoise.dtype == torch.float32 and hnoise.is_contiguous())
        d.heads_ws, d.heads_bs, d.heads_wc, d.heads_bc, d.heads_noise, d.heads_raw = _p(hws), _p(hbs), _p(hwc), _p(hbc), _p(hnoise), _p(hraw)
    for i, ly in enumerate(layers):
        L = d.layers[i]
        assert ly.w.dtype == x.dtype and hasattr(ly.w, "swn_nk"), "weights must come from ops.pack_weights (compute dtype)"
        L.w, L.b, L.save, L.mask = _p(ly.w), _p(ly.b), _p(ly.save), _p(ly.mask)
        L.rowbias, L.rows_per_bias = _p(ly.rowbias), ly.rows_per_bias
        L.n, L.k = ly.w.swn_nk
        L.relu, L.skip = ly.relu, int(ly.skip)
    call("swn_mlp_chain", C.byref(d), _stream())
    return y


_chain_sched = {}
_dwsig_ws = {}


def chain_sched(dev, key):
    """The tile-queue counters of a persistent chain launch (geometry 6 / 7): int32 [16], zero at creation and left zero by every launch.
    One tensor per (device, stream, key): launches that share one are ordered on their stream; give launches that may overlap on
    different streams different keys.  Never freed (a captured hipGraph holds the address)."""
    k = (dev, torch.cuda.current_stream(dev).cuda_stream if dev.type == "cuda" else 0, key)
    t = _chain_sched.get(k)
    if t is None:
        t = _chain_sched[k] = torch.zeros(16, dtype=torch.int32, device=dev)
    return t


_wgrad_ws = {}


def _wgrad_workspace(dev, nbytes: int):
    """One workspace per (device, stream) - launches on different streams may overlap -, grown to the largest request."""
    key = (dev, torch.cuda.current_stream().cuda_stream)
    held = _wgrad_ws.setdefault(key, [])
    for ws in held:
        if ws.numel() >= nbytes:
            return ws
    # a larger request: a NEW buffer next to the old ones, which are NEVER freed (a captured hipGraph may hold an old one's address
    # for as long as the process lives; a handful of growth steps at most - the sizes depend on the job count and the weight sets only)
    ws = torch.empty(int(nbytes), dtype=torch.uint8, device=dev)
    held.insert(0, ws)
    return ws


def _multi_ws_bytes(n_jobs: int, n_wsets: int) -> int:
    return int(_lib.load().swn_wgrad_multi_workspace_bytes(int(n_jobs), int(n_wsets)))


def wgrad(a, b, dw, db=None, n_groups=1, n_wsets=1, group_stride=None, group_rows=None, group_rows_clamp=None, n_splits=8,
          tag=0, use_workspace=True, a_gather=None, b_gather=None):
    """dw [n_wsets, m_dim, n_dim] f32 += a^T b per group; db [n_wsets, n_dim] += colsum(b).
    a_gather / b_gather: read the rows of a / b through an index (the routing permutation) instead of a dispatched copy.
    With a workspace (default) the launch is the balanced stream kernel (swn_wgrad_multi: n_splits is ignored) whenever
    n_groups % n_wsets == 0; use_workspace=False adds the partial tiles with fp32 atomics (the row-split kernel)."""
    m_dim, n_dim = a.shape[1], b.shape[1]
    assert group_stride is not None or (a_gather is None and b_gather is None)
    if m_dim > 256 or n_dim > 256:      # wider than one 256 x 256 tile: column blocks of the operands, one launch
        return wgrad_batched([(a, b, dw, db, a_gather, b_gather)], n_groups, n_wsets, group_stride, group_rows, group_rows_clamp,
                             n_splits, tag)
    gs = int(group_stride if group_stride is not None else a.shape[0])
    ws, ws_bytes = None, 0
    if use_workspace:
        ws = _wgrad_workspace(a.device, max(int(n_groups) * int(n_splits) * (m_dim * n_dim + n_dim) * 4, _multi_ws_bytes(1, n_wsets)))
        ws_bytes = ws.numel()
    call("swn_wgrad", _p(a), _p(b), _p(a_gather), _p(b_gather), _dt(a), m_dim, n_dim, int(n_groups), int(n_wsets), gs, _p(group_rows),
         int(group_rows_clamp if group_rows_clamp is not None else gs), _p(dw), _p(db), int(n_splits), int(tag), _p(ws), ws_bytes, _stream())


def wgrad_multi(jobs, n_groups=1, n_wsets=1, group_stride=None, group_rows=None, group_rows_clamp=None, tag=0, group_begin=None):
    """jobs: tuples (a, b, dw, db, a_gather, b_gather) over ONE row grouping, each with its own widths (multiples of 32):
    the balanced stream launch (swn_wgrad_multi, include/swn.h) - up to 8 GEMMs per launch, the work cut into equal shares of the
    valid rows, deterministic reduction.  dw [n_wsets, m, n] f32 (accumulated into), db [n_wsets, n] or None.  Operands wider than
    256 features (the Mission Bay widths) are cut into 256-column blocks - one GEMM per (row block of dw, column block of dw) through
    the per-job leading dimensions - so packed row spaces (group_begin: the rows an expert-parallel rank received) work at any width."""
    a0 = jobs[0][0]
    gs = int(group_stride if group_stride is not None else a0.shape[0])
    esz = a0.element_size()
    gemms = []              # (a ptr, b ptr, a_gather, b_gather, dw ptr, db ptr, m, n, lda, ldb, ldw, dw_set_stride, db_set_stride)
    for (a, b, dw, db, ag, bg) in jobs:
        m, n = a.shape[1], b.shape[1]
        assert a.dtype == a0.dtype and b.dtype == a0.dtype and dw.shape[-2:] == (m, n) and dw.dtype == torch.float32
        bm, bn = min(m, 256), min(n, 256)
        assert m % bm == 0 and n % bn == 0, "operand widths above 256 must be multiples of 256"
        for i in range(0, m, bm):
            for j in range(0, n, bn):
                gemms.append((a.data_ptr() + i * esz, b.data_ptr() + j * esz, _p(ag), _p(bg), dw.data_ptr() + (i * n + j) * 4,
                              (db.data_ptr() + j * 4) if (db is not None and i == 0) else None, bm, bn, m, n, n, m * n, n))
    for i0 in range(0, len(gemms), 8):
        chunk = gemms[i0:i0 + 8]
        arr = (WgradJob * len(chunk))()
        for jb, (pa, pb, ag, bg, pdw, pdb, m, n, lda, ldb, ldw, dws, dbs) in zip(arr, chunk):
            jb.a, jb.b, jb.a_gather, jb.b_gather, jb.dw, jb.db = pa, pb, ag, bg, pdw, pdb
            jb.m_dim, jb.n_dim, jb.lda, jb.ldb, jb.ldw = m, n, lda, ldb, ldw
            jb.dw_set_stride, jb.db_set_stride = dws, dbs
        ws = _wgrad_workspace(a0.device, _multi_ws_bytes(len(chunk), n_wsets))
        call("swn_wgrad_multi", arr, len(chunk), _dt(a0), int(n_groups), int(n_wsets), gs, _p(group_rows),
             int(group_rows_clamp if group_rows_clamp is not None else gs), _p(group_begin), int(tag), _p(ws), ws.numel(), _stream())


def wgrad_batched(items, n_groups=1, n_wsets=1, group_stride=None, group_rows=None, group_rows_clamp=None, n_splits=8, tag=0):
    """items: tuples (a, b, dw, db, a_gather, b_gather) of identical shapes / grouping -> launches of up to 8 GEMM blocks
    (see wgrad).  Operands wider than 256 features are cut into 256-column blocks (swn_wgrad_blocks)."""
    a0, b0 = items[0][0], items[0][1]
    M, N = a0.shape[1], b0.shape[1]
    bm, bn = min(M, 256), min(N, 256)
    esz = a0.element_size()
    gs = int(group_stride if group_stride is not None else a0.shape[0])
    blocks = []
    for (a, b, dw, db, ag, bg) in items:
        assert a.shape[1] == M and b.shape[1] == N and a.dtype == a0.dtype and dw.shape[-2:] == (M, N)
        for i in range(0, M, bm):
            for j in range(0, N, bn):
                blocks.append((a.data_ptr() + i * esz, b.data_ptr() + j * esz, _p(ag), _p(bg), dw.data_ptr() + (i * N + j) * 4,
                               (db.data_ptr() + j * 4) if (db is not None and i == 0) else None))
    per = int(n_groups) * int(n_splits) * (bm * bn + bn) * 4
    ws = _wgrad_workspace(a0.device, max(8 * per, _multi_ws_bytes(8, n_wsets)))
    for i0 in range(0, len(blocks), 8):
        chunk = blocks[i0:i0 + 8]
        arr = (WgradItem * len(chunk))()
        for i, (pa, pb, ag, bg, pdw, pdb) in enumerate(chunk):
            arr[i].a, arr[i].b, arr[i].a_gather, arr[i].b_gather, arr[i].dw, arr[i].db = pa, pb, ag, bg, pdw, pdb
        call("swn_wgrad_blocks", arr, len(chunk), _dt(a0), bm, bn, M, N, N, M * N, N, int(n_groups), int(n_wsets), gs, _p(group_rows),
             int(group_rows_clamp if group_rows_clamp is not None else gs), int(n_splits), int(tag), _p(ws), ws.numel(), _stream())


def adam_step(param, grad, m, v, shadow, step: int, lr: float, beta1=0.9, beta2=0.999, eps=1e-8, grad_scale=1.0):
    n = param.numel()
    call("swn_adam_step", _p(param), _p(grad), _p(m), _p(v), _p(shadow), _dt(shadow) if shadow is not None else F32, n,
         float(lr), float(beta1), float(beta2), float(eps), int(step), float(grad_scale), _stream())


def cast(src, dst):
    call("swn_cast", _p(src), _p(dst), _dt(dst), src.numel(), _stream())
    return dst


def cast_transpose(src, dst):
    """src [B, R, C] f32 -> dst [B, C, R] (dst dtype)"""
    B, R, Cc = src.shape
    call("swn_cast_transpose", _p(src), _p(dst), _dt(dst), B, R, Cc, _stream())
    return dst
