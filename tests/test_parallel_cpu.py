"""World-size-2 gloo test of the data-parallel host logic (ray sharding + flat-gradient all-reduce), CPU only."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from switch_nerf_amd import parallel
    r, w = parallel.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    b, e = parallel.shard_rays(8192, rank, world)
    flat = torch.full((1000,), float(rank + 1))
    flat[rank] = 100.0
    scale = parallel.make_grad_allreduce()(flat)
    out[rank] = (b, e, scale, flat[:3].tolist(), flat[5].item())
    dist.destroy_process_group()


def test_dp_allreduce_and_sharding_gloo():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, 29531 + os.getpid() % 200, out), nprocs=world, join=True)
    assert out[0][:2] == (0, 4096) and out[1][:2] == (4096, 8192)
    for r in range(world):
        b, e, scale, head, mid = out[r]
        assert scale == 0.5
        assert head == [102.0, 101.0, 3.0] and mid == 3.0     # identical summed buffer on both ranks


def _overlap_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from switch_nerf_amd import parallel
    parallel.init_from_env(backend="gloo")
    g = torch.Generator().manual_seed(5 + rank)
    flat = torch.randn(1000, generator=g)
    ref = flat.clone()
    f = parallel.make_grad_allreduce()
    s0 = f(ref)                                          # one bucket
    n_dense = 300                                        # overlapped form: the tail ("expert block") first, the prefix behind it
    f.begin(flat[n_dense:])
    flat[:n_dense] += 0.0                                # (the second half of the backward would run here)
    s1 = f.finish(flat[:n_dense])
    out[rank] = (s0, s1, bool(torch.equal(flat, ref)), ref[:4].tolist())
    try:
        f.finish(flat[:n_dense])
        out[rank] += (False,)
    except AssertionError:
        out[rank] += (True,)
    dist.destroy_process_group()


def test_overlapped_two_part_allreduce_equals_one_bucket_gloo():
    """GradAllReduce.begin / finish (the expert block travels between the two backward graphs, the dense prefix behind the second:
    graph.GraphedTrainStep) sums the same elements over the same ranks as the one-bucket call: bit-identical buffers on every rank."""
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_overlap_worker, args=(world, 29331 + os.getpid() % 200, out), nprocs=world, join=True)
    assert out[0][3] == out[1][3]
    for r in range(world):
        s0, s1, same, _head, guarded = out[r]
        assert s0 == s1 == 0.5 and same and guarded


def _route(rng, n_seg, seg_tokens, E, cap):
    """Random top-1 routing with capacity in the native layout of swn_route_top1: perm, counts, tok2row."""
    import numpy as np
    P = n_seg * seg_tokens
    idx = rng.integers(0, E, P)
    perm = np.full((n_seg, E, cap), -1, np.int32)
    counts = np.zeros((n_seg, E), np.int32)
    tok2row = np.full(P, -1, np.int32)
    for t in rng.permutation(P):
        s, e = t // seg_tokens, idx[t]
        counts[s, e] += 1                       # like the reference, counts include the dropped tokens
        if counts[s, e] <= cap:
            l = counts[s, e] - 1
            perm[s, e, l] = t
            tok2row[t] = (s * E + e) * cap + l
    return idx, perm.reshape(-1), counts, tok2row


def _ep_worker(rank, world, port, out):
    import numpy as np
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from switch_nerf_amd import parallel
    parallel.init_from_env(backend="gloo")
    E, n_seg, seg_tokens, cap, M = 4, 3, 40, 10, 8
    ep = parallel.ExpertParallel(rank, world, E)
    rng = np.random.default_rng(100 + rank)
    idx, perm, counts, tok2row = _route(rng, n_seg, seg_tokens, E, cap)
    P = n_seg * seg_tokens
    x = torch.from_numpy(rng.standard_normal((P, M)).astype(np.float32)) + 10.0 * rank
    perm_t, counts_t, t2r = torch.from_numpy(perm), torch.from_numpy(counts), torch.from_numpy(tok2row)
    # per routing segment: the dispatched rows in native order (expert, slot) are the payload (destination rank, local expert, slot)
    crecv = ep.exchange_counts(counts_t, cap)()                 # [n_seg, world * E_local] valid rows of every received group
    seg_rows = E * cap
    back = torch.zeros(n_seg * seg_rows, M)
    for s_ in range(n_seg):
        ps = perm_t[s_ * seg_rows:(s_ + 1) * seg_rows].long()
        send = torch.where((ps >= 0)[:, None], x[ps.clamp(min=0)], torch.zeros(1, M))
        recv, wait = ep.all_to_all(send)
        wait()
        # mock expert: y = x * (e + 1) + e on the valid rows of every received group; the group's local expert is g % E_local
        y = torch.zeros_like(recv)
        for g in range(world * ep.El):
            e = rank * ep.El + g % ep.El
            n = int(crecv[s_, g])
            y[g * cap: g * cap + n] = recv[g * cap: g * cap + n] * (e + 1) + e
        _, bwait = ep.all_to_all(y, out=back[s_ * seg_rows:(s_ + 1) * seg_rows])
        bwait()
    rows = t2r.long()                                           # the returned rows sit in the native row space: no remapping
    got = torch.where((rows >= 0)[:, None], back[rows.clamp(min=0)], torch.zeros(1, M))
    ref = torch.where(t2r[:, None] >= 0, x * (torch.from_numpy(idx)[:, None] + 1) + torch.from_numpy(idx)[:, None], torch.zeros(1, M))
    out[rank] = (bool(torch.equal(got, ref)), int((t2r >= 0).sum()), int(crecv.sum()))
    dist.destroy_process_group()


def test_expert_parallel_exchange_gloo():
    """Tokens routed on 2 ranks reach the rank that owns their expert (per-segment payload order = the expert kernels' group order),
    are processed by the right local expert, and come back to the native row the combine gathers from."""
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_ep_worker, args=(world, 29331 + os.getpid() % 200, out), nprocs=world, join=True)
    assert out[0][0] and out[1][0]
    assert out[0][1] + out[1][1] == out[0][2] + out[1][2]      # every kept token is processed exactly once, somewhere


def _ragged_worker(rank, world, port, out):
    import numpy as np
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from switch_nerf_amd import parallel
    parallel.init_from_env(backend="gloo")
    E, M = 4, 6
    ep = parallel.ExpertParallel(rank, world, E)
    rng = np.random.default_rng(300 + rank)
    counts = torch.from_numpy(rng.integers(0, 9, E).astype(np.int32))               # tokens of this rank per (global) expert
    counts[rank] = 0                                                                 # an empty group on the way
    n = int(counts.sum())
    # packed rows, expert-major: value = 1000 * source rank + 100 * expert + slot (so the receiver can check who sent what)
    rows = torch.cat([torch.full((int(c), M), 1000.0 * rank + 100.0 * e) + torch.arange(int(c), dtype=torch.float32)[:, None]
                      for e, c in enumerate(counts)]) if n else torch.zeros(0, M)
    recv, rc = ep.all_to_all_ragged(rows, counts)
    ok = recv.shape[0] == int(rc.sum())
    pos = 0
    for g in range(world * ep.El):                                                   # groups arrive as (source rank, local expert)
        src, e = g // ep.El, rank * ep.El + g % ep.El
        c = int(rc[g])
        want = torch.full((c, M), 1000.0 * src + 100.0 * e) + torch.arange(c, dtype=torch.float32)[:, None]
        ok = ok and torch.equal(recv[pos:pos + c], want)
        pos += c
    back, _ = ep.all_to_all_ragged(recv * 2.0, rc, recv_counts=counts)               # the way back: every row returns to its slot
    out[rank] = bool(ok and torch.equal(back, rows * 2.0))
    dist.destroy_process_group()


def test_expert_parallel_ragged_exchange_gloo():
    """The evaluation path's unequal-split exchange (the reference's list_all_to_all): packed rows reach the owner of their expert
    grouped by (source rank, local expert) and come back to their own slot."""
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_ragged_worker, args=(world, 29731 + os.getpid() % 200, out), nprocs=world, join=True)
    assert out[0] and out[1]


def test_expert_parallel_single_rank_is_identity():
    sys.path.insert(0, ROOT)
    from switch_nerf_amd import parallel
    ep = parallel.ExpertParallel(0, 1, 8)
    counts = torch.tensor([[3, 9, 0, 8, 1, 2, 8, 30], [8, 8, 8, 8, 8, 8, 8, 8]], dtype=torch.int32)
    assert torch.equal(ep.exchange_counts(counts, 8)(), counts.clamp(max=8))
    s = torch.zeros(4, 2)
    r, w = ep.all_to_all(s, out=s)
    w()
    assert r is s


def _ep_packed_worker(rank, world, port, out):
    """Round 3 protocol: kept rows only, packed per (segment, expert), unequal-split exchange planned from ONE host read of the counts."""
    import numpy as np
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from switch_nerf_amd import parallel
    parallel.init_from_env(backend="gloo")
    E, n_seg, seg_tokens, cap, M = 4, 3, 40, 10, 8
    ep = parallel.ExpertParallel(rank, world, E)
    rng = np.random.default_rng(300 + rank)
    idx, perm, counts, tok2row = _route(rng, n_seg, seg_tokens, E, cap)
    P = n_seg * seg_tokens
    x = torch.from_numpy(rng.standard_normal((P, M)).astype(np.float32)) + 10.0 * rank
    counts_t = torch.from_numpy(counts)
    kept = counts_t.clamp(max=cap)                                                  # [n_seg, E]
    # the packed row space of swn_route_pack on the kept rows: group (segment, expert) starts at the exclusive prefix sum of `kept`
    begin = (torch.cumsum(kept.reshape(-1), 0) - kept.reshape(-1)).long()
    perm_c = torch.from_numpy(perm).view(n_seg, E, cap)
    packed_perm = torch.cat([perm_c[s_, e, : int(kept[s_, e])] for s_ in range(n_seg) for e in range(E)]).long()      # packed row -> token
    recv_counts = ep.exchange_counts(counts_t, cap)()                               # [n_seg, W * E_local]
    pl = ep.plan(kept, recv_counts)
    so, ro = pl["send_off"], pl["recv_off"]
    assert so[-1] == int(kept.sum()) and ro[-1] == int(recv_counts.sum())
    send = x[packed_perm]                                                           # swn_gather_rows through the packed permutation
    xr = torch.zeros(ro[-1], M)
    back = torch.zeros(so[-1], M)
    rbeg = (torch.cumsum(recv_counts.reshape(-1), 0) - recv_counts.reshape(-1)).long()
    ngs = world * ep.El
    for s_ in range(n_seg):
        ep.all_to_all_v(send[so[s_]:so[s_ + 1]], pl["in_splits"][s_], xr[ro[s_]:ro[s_ + 1]], pl["out_splits"][s_])()
        y = torch.zeros(ro[s_ + 1] - ro[s_], M)
        for g in range(ngs):                        # received groups in (source rank, local expert) order, first rows from a prefix sum
            e = rank * ep.El + g % ep.El
            b0, n = int(rbeg[s_ * ngs + g]), int(recv_counts[s_, g])
            y[b0 - ro[s_]: b0 - ro[s_] + n] = xr[b0: b0 + n] * (e + 1) + e
        ep.all_to_all_v(y, pl["out_splits"][s_], back[so[s_]:so[s_ + 1]], pl["in_splits"][s_])()
    # the returned rows sit in the packed row space: token -> begin[(segment, expert)] + slot
    t2r = torch.from_numpy(tok2row).long()
    slot = t2r % cap
    grp = t2r // cap
    rows = begin[grp.clamp(min=0)] + slot
    keep = t2r >= 0
    got = torch.where(keep[:, None], back[rows.clamp(0, max(so[-1] - 1, 0))], torch.zeros(1, M))
    ie = torch.from_numpy(idx)[:, None]
    ref = torch.where(keep[:, None], x * (ie + 1) + ie, torch.zeros(1, M))
    sent_rows = sum(sum(r) - r[rank] for r in pl["in_splits"])
    out[rank] = (bool(torch.equal(got, ref)), int(keep.sum()), ro[-1], sent_rows, n_seg * E * cap)
    dist.destroy_process_group()


def test_expert_parallel_kept_rows_only_exchange_gloo():
    """The round-3 exchange on 2 ranks: packed kept rows, unequal splits from one host read of the counts, received groups addressed by
    a prefix sum, the way back with swapped splits - every kept token is processed by its expert's owner and returns to its packed row;
    no padding row travels (rows leaving a rank <= its kept rows < the capacity-padded payload)."""
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_ep_packed_worker, args=(world, 29431 + os.getpid() % 200, out), nprocs=world, join=True)
    assert out[0][0] and out[1][0]
    assert out[0][1] + out[1][1] == out[0][2] + out[1][2]
    for r in range(world):
        assert out[r][3] <= out[r][1] < out[r][4]


def _loopback_worker(rank, world, port, out):
    """ONE rank, gloo, loopback: every collective of the W > 1 code path is issued (the mode tests/test_rccl_gpu.py runs over RCCL)."""
    import warnings
    sys.path.insert(0, ROOT)
    os.environ.update(RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from switch_nerf_amd import parallel
    assert parallel.init_from_env(backend="gloo", loopback=True) == (0, 1) and dist.is_initialized() and dist.get_world_size() == 1
    ep = parallel.ExpertParallel(0, 1, 8, loopback=True)
    local = parallel.ExpertParallel(0, 1, 8)
    res = dict(local_flags=(ep.local, local.local))
    x = torch.arange(40.0).view(10, 4)
    recv = torch.zeros(10, 4)
    ep.all_to_all_v(x[:7], [7], recv[:7], [7])()
    res["a2a_v"] = bool(torch.equal(recv[:7], x[:7]) and recv[7:].abs().sum() == 0)
    r2, wait = ep.all_to_all(x)
    wait()
    res["a2a"] = bool(r2.data_ptr() != x.data_ptr() and torch.equal(r2, x))
    counts = torch.tensor([[3, 300, 0, 7, 256, 1, 2, 9]], dtype=torch.int32)
    res["counts"] = bool(torch.equal(ep.exchange_counts(counts, 256)(), counts.clamp(max=256)))
    gc = torch.tensor([2, 0, 3, 1, 0, 0, 4, 0], dtype=torch.int32)
    rows = torch.randn(10, 4)
    back, rc = ep.all_to_all_ragged(rows, gc)
    res["ragged"] = bool(torch.equal(back, rows) and torch.equal(rc, gc))
    ar = parallel.make_grad_allreduce(loopback=True)
    g = torch.randn(1000)
    ref = g.clone()
    ar.begin(g[300:])
    s = ar.finish(g[:300])
    res["allreduce"] = bool(ar.active and s == 1.0 and torch.equal(g, ref))
    # a begin() without its finish(): the stale part is drained with a warning and counted
    ar.begin(g[300:])
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        ar.begin(g[300:])
        res["stale_warned"] = bool(any("begin() without finish()" in str(x_.message) for x_ in w) and ar.stale_drains == 1)
    ar.finish(g[:300])
    parallel.shutdown(timeout_s=30.0)
    res["down"] = not dist.is_initialized()
    out[0] = res


def test_loopback_mode_issues_every_collective_gloo_world1():
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_loopback_worker, args=(1, 29731 + os.getpid() % 200, out), nprocs=1, join=True)
    r = out[0]
    assert r["local_flags"] == (False, True)
    assert all(r[k] for k in ("a2a_v", "a2a", "counts", "ragged", "allreduce", "stale_warned", "down")), r
