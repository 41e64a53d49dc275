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
      swn_heads_bwd on a 0 / 1 stand-in built from the returned bits (its weight-gradient outputs go to scratch)
    d_raw[token] (16 B)  -------------------------------->   heads backward, tag 8 (tail backward + combine backward + expert backward),
                                                             tail / head / expert weight gradients (dense ones: summed by the all-reduce)
    front backward, router backward  <---- dx (512 B) + gate gradient (4 B) per kept row

1092 bytes per kept token instead of 2048 (0.53 x; dropped tokens cost nothing), and both fused launches back.  Every row's arithmetic is the
local fused launch's: raw, dx and the gate gradient are bit-identical to the data-parallel step, weight gradients are the same sums in
another order.  This first version exchanges all segments, then runs ONE launch per direction (no overlap of the exchange with the
experts yet), reads the split sizes on the host once per forward like the kept-rows mode, and is eager (not captured).
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


def _xchg(ep, send, in_splits, recv, out_splits):
    """One unequal-split all-to-all of packed rows (blocking); no process group: a copy."""
    if ep.local:
        recv.copy_(send)
        return
    ep.all_to_all_v(send, in_splits, recv, out_splits, None)()


def _all_gather(ep, t):
    if ep.local:
        return t
    out = torch.empty((ep.world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out, t.contiguous(), group=ep.group)
    return out


def _plan(m, c):
    """Counts exchange + ONE host read: the split sizes of every segment's exchanges (all of them carry KEPT rows: same splits)."""
    ep = m.ep
    W, El, n_seg, cap = ep.world, ep.El, c["n_seg"], c["cap"]
    kept = c["counts"].clamp(max=cap)
    rk = ep.exchange_counts(kept, cap, None)()                     # [n_seg, W * El]: kept rows of every received group
    per = lambda t: t.view(n_seg, W, El).sum(2)
    host = torch.cat([per(kept), per(rk), (c["counts"] - kept).sum().view(1, 1).expand(n_seg, 1)], 1).to("cpu", torch.int64).tolist()   # the one sync
    ks, kr = [r[:W] for r in host], [r[W:2 * W] for r in host]
    cum = lambda rows: [0] + [sum(sum(r) for r in rows[:i + 1]) for i in range(len(rows))]
    return dict(ks=ks, kr=kr, rk=rk, kept=kept, n_kept=sum(map(sum, ks)), Rk=sum(map(sum, kr)), n_drop=host[0][2 * W], so=cum(ks), ro=cum(kr))


def _exchange_rows(ep, pl, send, recv, back=False):
    """Per segment: the kept rows' records, packed (destination rank, local expert, slot) -> (source rank, local expert, slot); back = home."""
    for s in range(len(pl["ks"])):
        if back:
            _xchg(ep, send[pl["ro"][s]:pl["ro"][s + 1]], pl["kr"][s], recv[pl["so"][s]:pl["so"][s + 1]], pl["ks"][s])
        else:
            _xchg(ep, send[pl["so"][s]:pl["so"][s + 1]], pl["ks"][s], recv[pl["ro"][s]:pl["ro"][s + 1]], pl["kr"][s])


def forward(m, c, pe_dir, image_indices, sigma_noise, sv):
    """The expert-parallel branch of SwitchNeRF._net_forward_rows behind the routing: fills c["raw"] (token order) and what backward_a needs."""
    o, ep, dev, dt = ops, m.ep, m.dev, m.dtype
    W, El, E, M, H2, L = ep.world, ep.El, m.E, m.M, m.H2, m.L
    P, S, N, n_seg, cap, seg_tokens, tag = c["P"], c["S"], c["N"], c["n_seg"], c["cap"], c["seg_tokens"], c["tag"]
    _b = lambda name, shape, dtype: m._buf(tag + ":ot_" + name, shape, dtype)
    c["ray_feat"], c["c_ray"] = o.ray_feat_fwd(pe_dir, m.in_dir, m.p["emb"], image_indices.contiguous(), m.p["l2r.w"], m.p["l2.b"])
    c_ray_all = _all_gather(ep, c["c_ray"])
    pl = _plan(m, c)
    n_kept, Rk, n_drop = pl["n_kept"], pl["Rk"], pl["n_drop"]
    assert n_kept + n_drop == P, "every token is either kept or dropped"
    T = Rk + n_drop                                           # this rank's token space: [received kept rows | its OWN dropped tokens]
    idx_kept = torch.where(c["loc"] < cap, c["idx"], torch.full_like(c["idx"], -1))
    _gb, perm_p, c["row_of_tok"] = o.route_pack(idx_kept, c["loc"], pl["kept"], seg_tokens, E)      # packed row space of this rank's kept rows
    tok_k = perm_p[:n_kept].long()                            # token of every kept row this rank sends
    tok_d = c["dropped"][:n_drop].long()                      # ... and its dropped tokens (they never travel: their tail runs here)
    # ---- out: the kept tokens' x rows and one 16-byte record each (gate value, global ray, sigma noise) ----
    rows_b = n_seg * E * cap                                  # bound of the received kept rows (capacity of the local experts over all sources)
    TB = rows_b + P                                           # bound of the token space
    xr = _b("xr", (rows_b, M), dt)
    if ep.local:                                              # (no process group: the packed rows ARE the received rows)
        o.gather_rows(c["h0"], perm_p[:n_kept], xr[:n_kept])
    else:
        x_send = _b("x_send", (P, M), dt)[:n_kept]
        o.gather_rows(c["h0"], perm_p[:n_kept], x_send)
        _exchange_rows(ep, pl, x_send, xr[:Rk])
    ray_of = lambda tok: (tok // S + ep.rank * N).to(torch.int32)
    aux_send = torch.zeros(n_kept, 4, dtype=torch.float32, device=dev)
    aux_send[:, 0] = c["gmax"][tok_k]
    aux_send[:, 1] = ray_of(tok_k).view(torch.float32)
    if sigma_noise is not None:
        aux_send[:, 2] = sigma_noise[tok_k]
    aux_recv = torch.empty(Rk, 4, dtype=torch.float32, device=dev)
    _exchange_rows(ep, pl, aux_send, aux_recv)
    gmax_t = torch.cat([aux_recv[:, 0], torch.zeros(n_drop, dtype=torch.float32, device=dev)])
    ray_t = torch.cat([aux_recv[:, 1].contiguous().view(torch.int32), ray_of(tok_d)]).long()
    noise_t = torch.cat([aux_recv[:, 2], sigma_noise[tok_d]]) if sigma_noise is not None else None
    c_row = _b("c_row", (TB, H2), torch.float32)[:T]
    torch.index_select(c_ray_all, 0, ray_t, out=c_row)
    # ---- the fused launch on the token space ----
    ngs = n_seg * W * El
    grp_rows = pl["rk"].reshape(-1).contiguous()
    ep_begin = (torch.cumsum(grp_rows, 0, dtype=torch.int32) - grp_rows).contiguous()
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
            o.Layer(m.wf["l2h_pad"], None, relu=1, rowbias=c_row, rows_per_bias=1)]
    heads = (m.p["sigma.w"], m.p["sigma.b"], m.p["color.w"], m.p["color.b"], noise_t, raw_t)
    with m._timed("expert_fwd"):
        o.mlp_chain(xr, lys, h2_t, n_groups=ngs, n_wsets=El, group_stride=cap, group_rows=grp_rows, group_rows_clamp=cap, x_gather=ident,
                    tag=7, geometry=7, heads=heads, group_begin=ep_begin, tail=(L, gmax_t, drop_begin_t, dropped_t, H2))
    # ---- home: raw (+ the sign bits of h2: what the per-ray bias gradient needs of it) ----
    ncol = 8 if sv else 4
    ret_t = torch.empty(T, ncol, dtype=torch.float32, device=dev)
    ret_t[:, :4] = raw_t
    if sv:
        ret_t[:, 4:] = o.sign_bits_pack(h2_t).view(torch.float32)      # bit j of word q = (h2[:, 32 q + j] > 0)
    ret_recv = torch.empty(n_kept, ncol, dtype=torch.float32, device=dev)
    _exchange_rows(ep, pl, ret_t[:Rk], ret_recv, back=True)
    c["raw"] = torch.empty(P, 4, dtype=torch.float32, device=dev)
    c["raw"][tok_k] = ret_recv[:, :4]
    c["raw"][tok_d] = ret_t[Rk:, :4]
    if sv:
        c["h2_bits"] = torch.empty(P, 4, dtype=torch.int32, device=dev)
        c["h2_bits"][tok_k] = ret_recv[:, 4:].contiguous().view(torch.int32)
        c["h2_bits"][tok_d] = ret_t[Rk:, 4:].contiguous().view(torch.int32)
    c["ep_owner"] = dict(pl=pl, tok_k=tok_k, tok_d=tok_d, perm_p=perm_p, xr=xr, saves=saves, masks=masks, y=y_t, h1=h1_t, h2=h2_t, raw=raw_t,
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
    pl, T, Rk = q["pl"], q["T"], q["Rk"]
    n_kept = pl["n_kept"]
    _b = lambda name, shape, dtype: m._buf(tag + ":ot_" + name, shape, dtype)
    # ---- source: the per-ray bias gradient from d_raw, raw and the sign of h2 (dh2 = (h2 > 0) * the colour heads' input gradient) ----
    h2_sign = o.sign_bits_unpack(c["h2_bits"], dt)
    scr = m._bufs.get("_ot_scratch")
    if scr is None:
        scr = m._bufs["_ot_scratch"] = [torch.zeros(n, dtype=torch.float32, device=dev) for n in (M, 1, 3 * H2, 3)]
    _dh2, _dsig, dc_ray = o.heads_bwd(None, h2_sign, m.p["color.w"], c["raw"], d_raw, scr[0], scr[1], scr[2].view(3, H2), scr[3], rows_per_group=S)
    if dc_ray.shape[1] in (64, 128, 256) and c["ray_feat"].shape[1] <= 256:
        o.ray_feat_wgrad(c["ray_feat"], dc_ray, g["l2r.w"], g["l2.b"])
    else:
        g["l2r.w"].addmm_(c["ray_feat"].t(), dc_ray)
        g["l2.b"].add_(dc_ray.sum(0))
    o.emb_grad(dc_ray @ m.p["l2r.w"][m.in_dir:].t(), c["image_indices"].contiguous(), g["emb"])
    # ---- d_raw to the owners ----
    d_recv = torch.empty(Rk, 4, dtype=torch.float32, device=dev)
    _exchange_rows(ep, pl, d_raw[q["tok_k"]], d_recv)
    d_raw_t = torch.cat([d_recv, d_raw[q["tok_d"]]])
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
    if not ep.local:
        _exchange_rows(ep, pl, dx_r[:Rk], dx[:n_kept], back=True)
    dg_send = dgmax_t[:Rk].contiguous().view(Rk, 1)
    dg_recv = torch.empty(n_kept, 1, dtype=torch.float32, device=dev)
    _exchange_rows(ep, pl, dg_send, dg_recv, back=True)
    dgmax = m._buf(tag + ":dgmax", (P,), torch.float32)
    dgmax.zero_()                                             # (a dropped token's gate gradient is zero: no expert saw it)
    dgmax[q["perm_p"][:n_kept].long()] = dg_recv[:, 0]
    return dict(c=c, d_laux=d_laux, dgmax=dgmax, dx=dx, dout=None, returns=[], tail_jobs=[], nsp=max(1, min(256, P // 1024)), side_done=None,
                keep=(dc_ray, dh2_t, dsig_t))
