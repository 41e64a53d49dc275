"""Expert parallelism with the dense tail on the EXPERT's rank (round 6; VERDICT round 5 item 4, missing 3).

The reference exchanges the dispatched rows around the experts and runs everything else where the tokens live
(/root/reference/switch_nerf/modules/tutel_moe_ext/tutel_moe_layer_nobatch.py:157-185, 220-225).  parallel.ExpertParallel's default mode
does the same with kept rows only; the dense tail (gate scaling + ReLU, Linear "1", Linear "2" + per-ray bias, sigma / colour heads) then
cannot ride the expert launches (chain_big.hip tags 7 / 8: the tokens' gate values, per-ray biases and head outputs live on the source
rank) and four 512-byte rows per kept token cross the links.

Here the OWNER of a token's expert runs the whole fused launch on a RECEIVED TOKEN SPACE:

    source rank                                              owner rank (of the token's routed expert)
    front chain, router, routing (rank-local, bit-equal to data parallel)
    kept rows x (512 B)  +  per kept token:
      gate value, global ray id, sigma noise (16 B)  ---->  token space = [received kept rows in (source, local expert, slot) order |
    per-ray half of layer "2" (c_ray)  -- all-gather -->                   the rank's OWN dropped tokens (their tail needs no expert: they stay)]
                                                             tag 7: experts + tail + heads on it (kept rows through an identity gather, the
                                                             dropped tokens as its dropped-token tiles, per-token bias row c_ray[ray])
    raw[token]  <---- raw (16 B) + sign bits of h2 (16 B)    saves y / h1 / h2 / expert activations stay on the owner
    compositing, loss, d_raw
    per-ray bias gradient: dc_ray = per-ray sums of dh2, and dh2 = (h2 > 0) * (colour-head gradient of d_raw, raw) needs h2's SIGN only:
      swn_ray_bias_grad_bits forms it from the returned bits, raw and d_raw (48 bytes per token)
    d_raw[token] (16 B)  -------------------------------->   heads backward, tag 8 (tail backward + combine backward + expert backward),
                                                             tail / head / expert weight gradients (dense ones: summed by the all-reduce)
    front backward, router backward  <---- dx (512 B) + gate gradient (4 B) per kept row

1092 bytes per kept token instead of 2048 (0.53 x; dropped tokens cost nothing), and both fused launches back.  Every row's arithmetic is the
local fused launch's: raw, dx and the gate gradient are bit-identical to the data-parallel step, weight gradients are the same sums in
another order.  Every payload travels as ONE unequal-split all-to-all over all segments (7 collectives per step + the counts + the all-gather
of the per-ray terms, instead of 4 per segment: xGMI wants few, large transfers), then ONE launch per direction runs (no overlap of the
exchange with the experts yet); the split sizes are read on the host once per forward like the kept-rows mode; eager (not captured).
The per-token bias row of layer "2" is found through swn_chain_desc.tail_bias_row (token -> global ray); the index work around the
exchanges is swn_gather_rows / swn_scatter_rows / swn_owner_aux (no torch indexing in the step).
"""
from __future__ import annotations

import torch
import torch.distributed as dist

from . import ops


def eligible(m, c, no_batch, row_range) -> bool:
    ep = m.ep
    esz = 4 if m.dtype == torch.float32 else 2
    return (ep is not None and getattr(ep, "owner_tail", False) and m._tail_fused() and c["geom"] == 7 and row_range is None and not no_batch
            and "l2r.w" in m.p and ep.world * c["P"] * m.M * esz < (1 << 32) - 64)


def _xchg(ep, pl, send, recv, back=False):
    """ONE unequal-split all-to-all of a payload's rows for ALL segments (blocking): out = source -> owner (splits ks -> kr), back = home.
    No process group: the caller built `send` inside `recv` (nothing moves)."""
    if ep.local:
        if recv.data_ptr() != send.data_ptr():
            recv.copy_(send)
        return recv
    a, b = (pl["kr"], pl["ks"]) if back else (pl["ks"], pl["kr"])
    ep.all_to_all_v(send, a, recv, b, None)()
    return recv


def _all_gather(ep, t):
    if ep.local:
        return t
    out = torch.empty((ep.world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out, t.contiguous(), group=ep.group)
    return out


def _excl(t):
    """exclusive prefix sum of a flattened int tensor, in its shape"""
    f = t.reshape(-1)
    return (torch.cumsum(f, 0, dtype=torch.int32) - f.to(torch.int32)).view(t.shape)


def _plan(m, c):
    """Counts exchange + ONE host read.  Every payload travels as ONE all-to-all over all segments: this rank's kept rows are sent in
    (destination rank, segment, local expert, slot) order and arrive in (source rank, segment, local expert, slot) order; the expert kernels
    find the group (segment, source, local expert) through its first row (swn_chain_desc.group_begin: any order will do)."""
    ep = m.ep
    W, El, n_seg, cap = ep.world, ep.El, c["n_seg"], c["cap"]
    kept = c["counts"].clamp(max=cap)                              # [n_seg, E]: rows this rank sends per (segment, expert)
    rk = ep.exchange_counts(kept, cap, None)()                     # [n_seg, W * El]: kept rows of every received group
    len_s = kept.view(n_seg, W, El).permute(1, 0, 2).contiguous()  # [destination, segment, local expert]
    len_r = rk.view(n_seg, W, El).permute(1, 0, 2).contiguous()    # [source, segment, local expert]
    host = torch.cat([len_s.sum((1, 2)), len_r.sum((1, 2)), (c["counts"] - kept).sum().view(1)]).to("cpu", torch.int64).tolist()   # the one sync
    ks, kr, n_drop = host[:W], host[W:2 * W], host[2 * W]
    grp_begin = _excl(len_r).permute(1, 0, 2).contiguous().view(-1)      # first received row of group (segment, source, local expert)
    return dict(ks=ks, kr=kr, rk=rk, kept=kept, len_s=len_s, n_kept=sum(ks), Rk=sum(kr), n_drop=n_drop, grp_begin=grp_begin)


def _send_order(m, c, pl):
    """(perm, row_of_tok): the token of every kept row in SEND order (destination, segment, local expert, slot), and its inverse (-1: the token
    was dropped).  One rank: swn_route_pack's order (segment, expert, slot) is the send order."""
    ep, dev = m.ep, m.dev
    W, El, n_seg, E, P = ep.world, ep.El, c["n_seg"], m.E, c["P"]
    n_kept = pl["n_kept"]
    idx_kept = torch.where(c["loc"] < c["cap"], c["idx"], torch.full_like(c["idx"], -1))
    _gb, perm_p, row_of_tok = ops.route_pack(idx_kept, c["loc"], pl["kept"], c["seg_tokens"], E)      # packed (segment, expert, slot) order
    perm_p = perm_p[:n_kept]
    if W == 1:
        return perm_p.contiguous(), row_of_tok
    start_a = _excl(pl["kept"]).view(n_seg, W, El).permute(1, 0, 2)                     # where a run starts in pack order, runs in send order
    shift = (start_a - _excl(pl["len_s"])).reshape(-1)
    src = torch.arange(n_kept, dtype=torch.int32, device=dev) + torch.repeat_interleave(shift, pl["len_s"].reshape(-1), output_size=n_kept)
    perm = perm_p[src.long()].contiguous()
    row_of_tok = torch.full((P, 1), -1, dtype=torch.int32, device=dev)
    ops.scatter_rows(torch.arange(n_kept, dtype=torch.int32, device=dev).view(-1, 1), perm, row_of_tok)
    return perm, row_of_tok.view(-1)


def forward(m, c, pe_dir, image_indices, sigma_noise, sv):
    """The expert-parallel branch of SwitchNeRF._net_forward_rows behind the routing: fills c["raw"] (token order) and what backward_a needs."""
    o, ep, dev, dt = ops, m.ep, m.dev, m.dtype
    W, El, E, M, H2, L = ep.world, ep.El, m.E, m.M, m.H2, m.L
    P, S, N, n_seg, cap, tag = c["P"], c["S"], c["N"], c["n_seg"], c["cap"], c["tag"]
    _b = lambda name, shape, dtype: m._buf(tag + ":ot_" + name, shape, dtype)
    c["ray_feat"], c["c_ray"] = o.ray_feat_fwd(pe_dir, m.in_dir, m.p["emb"], image_indices.contiguous(), m.p["l2r.w"], m.p["l2.b"])
    c_ray_all = _all_gather(ep, c["c_ray"])                   # the per-ray half of layer "2" of EVERY rank's rays (512 B per ray)
    pl = _plan(m, c)
    n_kept, Rk, n_drop = pl["n_kept"], pl["Rk"], pl["n_drop"]
    assert n_kept + n_drop == P, "every token is either kept or dropped"
    T = Rk + n_drop                                           # this rank's token space: [received kept rows | its OWN dropped tokens]
    perm, c["row_of_tok"] = _send_order(m, c, pl)
    dropped = c["dropped"][:max(n_drop, 1)]                   # this rank's dropped tokens (they never travel: their tail runs here)
    rows_b = n_seg * E * cap                                  # bound of the received kept rows (capacity of the local experts over all sources)
    TB = rows_b + P                                           # bound of the token space
    mine = lambda buf, name, shape, dtype: buf if ep.local else _b(name, shape, dtype)[:n_kept]      # (one rank: built where it is read)
    # ---- out: the kept tokens' x rows and one 16-byte record each (gate value, global ray, sigma noise) ----
    xr = _b("xr", (rows_b, M), dt)
    x_send = mine(xr[:n_kept], "x_send", (P, M), dt)
    o.gather_rows(c["h0"], perm, x_send)
    _xchg(ep, pl, x_send, xr[:Rk])
    aux_t = _b("aux", (TB, 4), torch.float32)[:T]
    aux_send = mine(aux_t[:n_kept], "aux_send", (P, 4), torch.float32)
    o.owner_aux(c["gmax"], sigma_noise, perm, S, ep.rank * N, aux_send)
    _xchg(ep, pl, aux_send, aux_t[:Rk])
    if n_drop:
        o.owner_aux(c["gmax"], sigma_noise, dropped[:n_drop], S, ep.rank * N, aux_t[Rk:], zero_gate=True)
    gmax_t = _b("gmax", (TB,), torch.float32)[:T]
    ray_t = _b("ray", (TB,), torch.int32)[:T]                 # token -> row of c_ray_all
    noise_t = _b("noise", (TB,), torch.float32)[:T] if sigma_noise is not None else None
    o.owner_aux_split(aux_t, gmax_t, ray_t, noise_t)
    # ---- the fused launch on the token space ----
    ngs = n_seg * W * El
    grp_rows = pl["rk"].reshape(-1).contiguous()
    ep_begin = pl["grp_begin"]
    ident = m._bufs.get(("ot_ident", rows_b))                # the identity gather: a received row IS its token
    if ident is None:
        ident = m._bufs[("ot_ident", rows_b)] = torch.arange(rows_b, dtype=torch.int32, device=dev)
    dropped_t = torch.arange(Rk, Rk + max(n_drop, 1), dtype=torch.int32, device=dev)      # (one spare entry: the list must not be empty)
    drop_begin_t = torch.tensor([0, n_drop], dtype=torch.int32, device=dev)
    saves, masks = c["saves"], c["masks"]                     # (the context's expert buffers: n_seg * E * cap rows, one mask word set per tile)
    skips = set(m.cfg["skips"])
    y_t = _b("y", (TB, M), dt)[:T] if sv else None
    h1_t = _b("h1", (TB, M), dt)[:T] if sv else None
    h2_t = _b("h2", (TB, H2), dt)[:T] if sv else None
    raw_t = _b("raw", (TB, 4), torch.float32)[:T]
    if "l2h_pad" not in m.wf:
        m.wf["l2h_pad"] = o.pack_weights_padded(m.p["l2h.w"].unsqueeze(0), dt, True, 0, 256)
    lys = [o.Layer(m._local_experts(m.wf[f"exp{l}"]), m._local_experts(m.p[f"exp{l}.b"]), relu=1 if l < L - 1 else 0, skip=(l in skips),
                   save=saves[l] if (sv and l < L - 1) else None, mask=masks[l] if (sv and l < L - 1) else None) for l in range(L)]
    lys[-1].save = y_t
    lys += [o.Layer(m.wf["l1"], m.p["l1.b"].view(1, M), save=h1_t),
            o.Layer(m.wf["l2h_pad"], None, relu=1, rowbias=c_ray_all, rows_per_bias=0)]      # (the bias row of token t: ray_t[t])
    heads = (m.p["sigma.w"], m.p["sigma.b"], m.p["color.w"], m.p["color.b"], noise_t, raw_t)
    with m._timed("expert_fwd"):
        o.mlp_chain(xr, lys, h2_t, n_groups=ngs, n_wsets=El, group_stride=cap, group_rows=grp_rows, group_rows_clamp=cap, x_gather=ident,
                    tag=7, geometry=7, heads=heads, group_begin=ep_begin, tail=(L, gmax_t, drop_begin_t, dropped_t, H2, ray_t))
    # ---- home: raw (+ the sign bits of h2: what the per-ray bias gradient needs of it), back into token order ----
    c["raw"] = torch.empty(P, 4, dtype=torch.float32, device=dev)
    raw_home = raw_t[:n_kept] if ep.local else torch.empty(n_kept, 4, dtype=torch.float32, device=dev)
    o.scatter_rows(_xchg(ep, pl, raw_t[:Rk], raw_home, back=True), perm, c["raw"])
    if n_drop:
        o.scatter_rows(raw_t[Rk:], dropped[:n_drop], c["raw"])
    if sv:
        bits_t = o.sign_bits_pack(h2_t)                       # bit j of word q = (h2[:, 32 q + j] > 0)
        c["h2_bits"] = torch.empty(P, H2 // 32, dtype=torch.int32, device=dev)
        bits_home = bits_t[:n_kept] if ep.local else torch.empty(n_kept, H2 // 32, dtype=torch.int32, device=dev)
        o.scatter_rows(_xchg(ep, pl, bits_t[:Rk], bits_home, back=True), perm, c["h2_bits"])
        if n_drop:
            o.scatter_rows(bits_t[Rk:], dropped[:n_drop], c["h2_bits"])
    c["ep_owner"] = dict(pl=pl, perm=perm, dropped_src=dropped, xr=xr, saves=saves, masks=masks, y=y_t, h1=h1_t, h2=h2_t, raw=raw_t,
                         gmax=gmax_t, ident=ident, dropped=dropped_t, drop_begin=drop_begin_t, grp_rows=grp_rows, ep_begin=ep_begin, ngs=ngs,
                         T=T, Rk=Rk, rows_b=rows_b, TB=TB)
    c["ep_counts"], c["ep_padded"], c["tail_fused"] = pl["rk"], False, True
    m._kernel_sel.update(ep_owner_tail=True, tail_fused=True)
    return c


def backward_a(m, c, d_raw, d_laux):
    """First half of the backward for a context of forward() (replaces SwitchNeRF.backward_net_a); returns backward_net_b's state."""
    o, ep, dev, dt = ops, m.ep, m.dev, m.dtype
    W, El, E, M, H2, L = ep.world, ep.El, m.E, m.M, m.H2, m.L
    P, S, N, cap, tag = c["P"], c["S"], c["N"], c["cap"], c["tag"]
    g, q = m.g, c["ep_owner"]
    pl, T, Rk, perm = q["pl"], q["T"], q["Rk"], q["perm"]
    n_kept, n_drop = pl["n_kept"], pl["n_drop"]
    _b = lambda name, shape, dtype: m._buf(tag + ":ot_" + name, shape, dtype)
    # ---- source: the per-ray bias gradient from d_raw, raw and the sign of h2 (dh2 = (h2 > 0) * the colour heads' input gradient) ----
    dc_ray = o.ray_bias_grad_bits(c["h2_bits"], c["raw"], d_raw.contiguous(), m.p["color.w"], S)
    if dc_ray.shape[1] in (64, 128, 256) and c["ray_feat"].shape[1] <= 256:
        o.ray_feat_wgrad(c["ray_feat"], dc_ray, g["l2r.w"], g["l2.b"])
    else:
        g["l2r.w"].addmm_(c["ray_feat"].t(), dc_ray)
        g["l2.b"].add_(dc_ray.sum(0))
    o.emb_grad(dc_ray @ m.p["l2r.w"][m.in_dir:].t(), c["image_indices"].contiguous(), g["emb"])
    # ---- d_raw to the owners ----
    d_raw = d_raw.contiguous()
    d_raw_t = _b("d_raw", (q["TB"], 4), torch.float32)[:T]
    d_send = d_raw_t[:n_kept] if ep.local else _b("d_send", (P, 4), torch.float32)[:n_kept]
    o.gather_rows(d_raw, perm, d_send)
    _xchg(ep, pl, d_send, d_raw_t[:Rk])
    if n_drop:
        o.gather_rows(d_raw, q["dropped_src"][:n_drop], d_raw_t[Rk:])
    # ---- owner: heads backward, the fused backward launch, weight gradients ----
    fused_dws = M == 256 and m.sw["fused_dwsig"]
    dh2_t, dsig_t = o.heads_bwd(None if fused_dws else q["y"], q["h2"], m.p["color.w"], q["raw"], d_raw_t, g["sigma.w"], g["sigma.b"],
                                g["color.w"], g["color.b"])
    m._kernel_sel.update(fused_backward=True, comb_dwsig=fused_dws)
    TB, rows_b, ngs = q["TB"], q["rows_b"], q["ngs"]
    dh1_t = _b("dh1", (TB, M), dt)[:T]
    dgmax_t = _b("dgmax", (TB,), torch.float32)[:T]
    dz = [_b(f"dz{l}", (rows_b, M), dt) for l in range(L - 1)]
    dz_last = _b("dz_last", (rows_b, M), dt)
    dx = m._buf(tag + ":dx", (c["rows"], M), dt)
    dx_r = dx if ep.local else _b("dx_r", (rows_b, M), dt)      # (no process group: the input gradients are written where the front backward reads)
    if "l2h_pad" not in m.wb:
        m.wb["l2h_pad"] = o.pack_weights_padded(m.p["l2h.w"].unsqueeze(0), dt, False, 0, 256)
    skip_l = list(m.cfg["skips"])[0] if len(m.cfg["skips"]) else None
    bl = []
    for i in range(L):
        l = L - 1 - i
        bl.append(o.Layer(m._local_experts(m.wb[f"exp{l}"]), None, relu=2 if l > 0 else 0, mask=q["masks"][l - 1] if l > 0 else None,
                          save=dz[l - 1] if l > 0 else None))
    with m._timed("expert_bwd"):
        o.mlp_chain(dh2_t, [o.Layer(m.wb["l2h_pad"], None, save=dh1_t), o.Layer(m.wb["l1"], None, save=dz_last)] + bl, dx_r, n_groups=ngs,
                    n_wsets=El, group_stride=cap, group_rows=q["grp_rows"], group_rows_clamp=cap, x_gather=q["ident"],
                    y_add=dz[skip_l] if skip_l is not None else None, tag=8, geometry=7, x_features=H2,
                    combine=(q["y"], dsig_t, m.p["sigma.w"], q["gmax"], dgmax_t, g["sigma.w"].view(-1) if fused_dws else None),
                    head=(2, q["drop_begin"], q["dropped"]), group_begin=q["ep_begin"])
    nsp = max(1, min(256, T // 1024))
    m._dense_wgrads([(q["h1"], dh2_t, g["l2h.w"].view(1, M, H2), None), (q["y"], dh1_t, g["l1.w"].view(1, M, M), g["l1.b"].view(1, M))], nsp)
    items = []
    for l in range(L):
        a = q["xr"] if l == 0 else q["saves"][l - 1]
        bz = dz_last if l == L - 1 else dz[l]
        items.append((a, bz, m._local_experts(g[f"exp{l}.w"]), m._local_experts(g[f"exp{l}.b"]), None, None))
    with m._timed("expert_wgrad"):
        o.wgrad_multi(items, n_groups=ngs, n_wsets=El, group_stride=cap, group_rows=q["grp_rows"], group_rows_clamp=cap, tag=1,
                      group_begin=q["ep_begin"])
    # ---- home: the experts' input gradient (512 B) and the gate gradient (4 B) of every kept row ----
    _xchg(ep, pl, dx_r[:Rk], dx[:n_kept], back=True)         # (in send order: c["row_of_tok"] is the front backward's gather)
    dg_t = dgmax_t[:Rk].view(Rk, 1)
    dg_home = dg_t if ep.local else torch.empty(n_kept, 1, dtype=torch.float32, device=dev)
    _xchg(ep, pl, dg_t, dg_home, back=True)
    dgmax = m._buf(tag + ":dgmax", (P,), torch.float32)
    dgmax.zero_()                                             # (a dropped token's gate gradient is zero: no expert saw it)
    o.scatter_rows(dg_home, perm, dgmax.view(P, 1))
    return dict(c=c, d_laux=d_laux, dgmax=dgmax, dx=dx, dout=None, returns=[], tail_jobs=[], nsp=max(1, min(256, P // 1024)), side_done=None,
                keep=(dc_ray, dh2_t, dsig_t))
