"""Multi-GPU plumbing for the hot path: one process per GPU, RCCL (backend "nccl" on ROCm) over xGMI.

Default: data parallel (below).  Optional: expert parallel (class ExpertParallel at the end of the file).

The reference's multi-GPU mode is DDP over rays (runner.py:203-207, :575: per-rank batch = batch_size // world_size,
SURVEY.md F4: expert parallelism is disabled as shipped).  Rays are independent and routing is rank-local, so ranks
never exchange activations; the only collective is one all-reduce of the flat fp32 gradient buffer per step, after which
every rank applies the same Adam update with grad_scale = 1 / world_size.
"""
from __future__ import annotations

import os
from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None, device: Optional[torch.device] = None, loopback: bool = False):
    """Initialise torch.distributed from torchrun's environment (RANK / WORLD_SIZE / MASTER_*).  Returns (rank, world).
    loopback: initialise the process group even for ONE rank (a world-1 RCCL communicator: every collective of the multi-GPU step is
    issued for real - same API calls, streams and graph captures as at W > 1 - and moves its payload inside the GPU)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if (world > 1 or loopback) and not dist.is_initialized():
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this driver
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = {}
        if backend == "nccl" and device is not None:
            kw["device_id"] = device
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world


def shutdown(timeout_s: float = 20.0, exit_code: int = 0):
    """destroy_process_group() with a watchdog.  After a hipGraph with captured RCCL all-to-alls has been replayed, the teardown of the
    communicator does not return on RCCL 2.26.6 / torch 2.10 + ROCm 7.0 (profiles/r06_experiments.md 2; results are complete at that
    point): the process then exits through os._exit(exit_code) after `timeout_s`.  Drop graph objects before calling this."""
    if not dist.is_initialized():
        return
    import gc
    import sys
    import threading
    gc.collect()
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    sys.stdout.flush()
    sys.stderr.flush()
    dog = threading.Timer(timeout_s, lambda: os._exit(exit_code))
    dog.daemon = True
    dog.start()
    dist.destroy_process_group()
    dog.cancel()


def shard_rays(n_rays: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous per-rank slice [begin, end) of a global ray batch (runner.py:575: batch_size // world_size each;
    the remainder, if any, is dropped like DataLoader(drop_last) would)."""
    per = n_rays // world
    return rank * per, (rank + 1) * per


class GradAllReduce:
    """The data-parallel gradient all-reduce (RCCL over xGMI with backend "nccl").

        f = GradAllReduce(group);  scale = f(flat_grad)          # one bucket, in place; returns the 1 / world factor Adam applies

    Overlapped form - what DDP's buckets do in the reference (runner.py:203-207: gradients are reduced while the backward still
    runs) - for a backward pass that is cut in two (SwitchNeRF.backward_net_a / _b, graph.GraphedTrainStep with two backward graphs):

        f.begin(grad[n_dense:], side_stream)     # after the first half: the expert block (93 % of the bytes) is final and travels on
                                                 # the side stream while the second half (router, front chain, dense weight
                                                 # gradients) runs on the launch stream
        scale = f.finish(grad[:n_dense])         # after the second half: the dense prefix, then the launch stream joins the side stream

    Both forms sum the same elements over the same ranks: results are bit-identical.  profile = True records HIP events around the
    collectives and the launch stream's waits (report())."""

    def __init__(self, group=None, loopback: bool = False):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # loopback: issue the collectives even with ONE rank (init_from_env(loopback=True)): the sums are the identity, the calls,
        # streams and events are the W > 1 ones
        self.active = self.world > 1 or (bool(loopback) and dist.is_initialized())
        self.profile = False
        self._pending = None
        self._ev = []            # (kind, start, end)
        self.stale_drains = 0    # begin() / __call__ found a part whose finish() never ran (see _drain)

    def _timed(self, kind, fn):
        if not self.profile or not torch.cuda.is_available():
            return fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        out = fn()
        b.record()
        self._ev.append((kind, a, b))
        return out

    def _drain(self):
        """A pending early part whose finish() never ran (an exception between the two backward halves): wait for it and forget it, so
        that the next step starts clean instead of dying on a stale handle."""
        if self._pending is not None:
            work, _part = self._pending
            self._pending = None
            self.stale_drains += 1
            import warnings
            warnings.warn("GradAllReduce: begin() without finish() - the pending part of the previous step was dropped; if that step's "
                          "optimizer update ran, its dense gradient block was never reduced", RuntimeWarning, stacklevel=3)
            if work is not None:
                work.wait()

    def __call__(self, flat: torch.Tensor) -> float:
        self._drain()
        if self.active:
            self._timed("coll", lambda: dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group))
        return 1.0 / self.world

    def begin(self, part: torch.Tensor, stream=None):
        """Start the all-reduce of `part` (final already) on `stream`, ordered after the work queued on the current stream."""
        self._drain()
        if not self.active:
            self._pending = (None, part)
            return
        if stream is None or not part.is_cuda:
            self._pending = (dist.all_reduce(part, op=dist.ReduceOp.SUM, group=self.group, async_op=True), part)
            return
        ready = torch.cuda.Event()
        ready.record()
        with torch.cuda.stream(stream):
            stream.wait_event(ready)

            def issue():
                w = dist.all_reduce(part, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                if self.profile:
                    w.wait()         # (orders `stream` after the collective so that the closing event times it)
                return w
            work = self._timed("coll", issue)
        self._pending = (work, part)

    def finish(self, rest: torch.Tensor) -> float:
        """All-reduce `rest` on the current stream, then make the current stream wait for the part begin() sent."""
        assert self._pending is not None, "GradAllReduce.finish without begin"
        work, part = self._pending
        self._pending = None
        if self.active:
            if rest.numel():
                self._timed("coll", lambda: dist.all_reduce(rest, op=dist.ReduceOp.SUM, group=self.group))
            self._timed("wait", work.wait)
            if part.is_cuda:
                part.record_stream(torch.cuda.current_stream())
        return 1.0 / self.world

    def report(self):
        """(after a synchronize) per profiled step set: total time of the collectives, time the launch stream waited for the early part,
        and the hidden fraction 1 - wait / (time of the early collective)."""
        coll = sum(a.elapsed_time(b) for k, a, b in self._ev if k == "coll")
        wait = sum(a.elapsed_time(b) for k, a, b in self._ev if k == "wait")
        n = sum(1 for k, _a, _b in self._ev if k == "coll")
        self._ev = []
        return dict(collectives=n, allreduce_ms=coll, wait_ms=wait, hidden_fraction=(1.0 - wait / coll) if coll > 0 else None)


def make_grad_allreduce(group=None, loopback: bool = False) -> GradAllReduce:
    """Returns f(flat_grad) -> grad_scale: sums the flat gradient buffer over ranks in place (one bucket) and returns
    the 1/world factor that swn_adam_step applies; f.begin / f.finish: the overlapped two-part form (GradAllReduce)."""
    return GradAllReduce(group, loopback)



class ExpertParallel:
    """Expert-parallel token exchange: BASELINE.json configs[2] / the reference's optional EP mode
    (/root/reference/switch_nerf/modules/tutel_moe_ext/tutel_moe_layer_nobatch.py:157-185: all-to-all of the dispatched
    [W, E_local, C, M] payload before the experts and back after them; runner.py:97-101: E_local = E // W).

    Rank r owns experts [r * E_local, (r + 1) * E_local).  Every rank still routes ITS OWN points (capacity, ranking and
    l_aux stay rank-local exactly as in the data-parallel mode, so routing indices do not depend on the mode); only the
    rows that enter the expert MLP travel.

    The exchange is PER ROUTING SEGMENT (= the reference's per-model-chunk MoE call) and carries KEPT ROWS ONLY: the rows of a
    segment that fit their expert's capacity, PACKED in (expert, capacity slot) order (swn_route_pack's layout: group g = (segment,
    expert) starts at the exclusive prefix sum of the kept counts), are the payload in (destination rank, local expert, slot) order;
    the send buffer is one swn_gather_rows through the packed permutation and the returned rows land in the packed row space the
    combine gathers from - no index remapping.  all_to_all_single with UNEQUAL splits (the reference pads to capacity,
    tutel_moe_layer_nobatch.py:157; padding rows cost 20 % of the bytes at 80 % kept rows) delivers (source rank, local expert, slot)
    = the group order of the expert kernels with group % E_local = local expert (no kernel knows about ranks; the groups' first rows
    come from a device prefix sum, swn_chain_desc.group_begin / swn_wgrad_multi's group_begin).  The valid-row counts of ALL segments
    travel once, ahead of the rows, and are read on the host ONCE per forward pass (plan()): the split sizes of every segment's
    exchange, forward and backward.
    Segments are pipelined: the exchange of segment s + 1 runs on a side HIP stream while the experts work on segment s, and the
    return of segment s overlaps both (model.py).  RCCL over xGMI is point-to-point: an all-to-all sends (W - 1) / W of the payload
    over the 7 links in parallel, there is nothing to gain from ring-style chunking.
    world == 1 degenerates to the identity (no process group needed): the single-GPU parity test runs this path.
    """

    def __init__(self, rank: int, world: int, n_experts: int, group=None, padded=False, loopback: bool = False, owner_tail: bool = False):
        """padded: False = kept rows only, unequal splits sized on the host (one device-to-host read per forward pass: the step cannot
        be captured into a hipGraph); True = the reference's own layout (tutel_moe_layer_nobatch.py:157: every (expert, capacity slot)
        travels, empty slots as zero rows) with EQUAL splits - nothing is read on the host, so the whole step, collectives included,
        can be replayed from a graph (graph.GraphedTrainStep); "auto" = padded when one segment's payload is at most PAD_AUTO_BYTES
        (the small per-GPU batches of a strong-scaling run, where ~150 eager launches cost more than the padding)."""
        if n_experts % world:
            raise ValueError(f"expert parallelism needs world ({world}) to divide the expert count ({n_experts})")
        self.rank, self.world, self.E, self.El, self.group = rank, world, n_experts, n_experts // world, group
        self.padded = padded
        # owner_tail: the dense tail runs on the EXPERT's rank (ep_owner.py): both fused launches stay, 0.53 x the bytes of the default mode
        # (kept rows out, raw + 16 bytes back; d_raw out, dx + the gate gradient back); eager, host-sized splits; 16-bit 256-feature models
        # (others fall back to the default exchange)
        self.owner_tail = bool(owner_tail)
        # local: ONE rank and no process group - nothing moves, the send buffers ARE the receive buffers.  loopback (one rank WITH a
        # process group, init_from_env(loopback=True)): every collective is issued like at W > 1 - separate send / receive buffers,
        # side stream, split sizes - and RCCL moves the payload inside the GPU: the W > 1 code path on a one-GPU box
        self.local = world == 1 and not (loopback and dist.is_initialized())

    PAD_AUTO_BYTES = 64 << 20

    def use_padded(self, segment_payload_bytes: int) -> bool:
        """Whether a step whose per-segment exchange carries this many capacity-padded bytes runs in the padded (host-free) mode."""
        if self.padded == "auto":
            return segment_payload_bytes <= self.PAD_AUTO_BYTES
        return bool(self.padded)

    @property
    def capturable(self) -> bool:
        """True when no forward pass of this mode reads split sizes on the host (see `padded`; "auto": the caller asks use_padded)."""
        return self.padded is True or self.padded == "auto"

    def exchange_counts(self, counts: torch.Tensor, cap: int, stream=None):
        """counts [n_seg, E] (tokens routed per expert, before capacity) -> (recv, wait): recv [n_seg, W * E_local] int32 = valid
        rows of every received group of every segment, in the expert kernels' group order (source rank, local expert)."""
        n_seg = counts.shape[0]
        send = counts.clamp(max=cap).view(n_seg, self.world, self.El).permute(1, 0, 2).contiguous()      # [dest rank, seg, el]
        recv, wait = self.all_to_all(send, stream)
        out = {}

        def wait_and_view():
            wait()
            out["v"] = recv.view(self.world, n_seg, self.El).permute(1, 0, 2).contiguous().view(n_seg, self.world * self.El)
            return out["v"]
        return wait_and_view

    def owner_of(self, expert: int) -> int:
        return expert // self.El

    # ---- evaluation without token dropping: unequal splits ----
    def all_to_all_ragged(self, rows: torch.Tensor, group_counts: torch.Tensor, recv_counts: Optional[torch.Tensor] = None):
        """The reference's list_all_to_all (tutel_communicate_nobatch.py:18-51, used at tutel_moe_layer_nobatch.py:308-335): the
        no-batch rows are packed, so every peer gets a different number of them.
          rows          [sum(group_counts), M]: groups in (destination rank, local expert) order, each contiguous;
          group_counts  int32 [W * E_local]: rows per group.
        Forward direction (recv_counts None): the counts travel first (equal split), the split sizes are read on the host (the
        reference's `.tolist()`, :312-313), then one all_to_all_single with unequal splits.  Returns (recv_rows, recv_counts): the rows
        of the (source rank, local expert) groups, packed, and their sizes.  Way back: pass the counts received on the way in as
        `recv_counts` - the split sizes are swapped and `group_counts` must be what this rank holds per (source rank, local expert)."""
        W, El = self.world, self.El
        if self.local:
            return rows, group_counts
        if recv_counts is None:
            recv_counts = torch.empty_like(group_counts)
            dist.all_to_all_single(recv_counts, group_counts.contiguous(), group=self.group)
        in_splits = group_counts.view(W, El).sum(1).tolist()
        out_splits = recv_counts.view(W, El).sum(1).tolist()
        out = torch.empty((int(sum(out_splits)),) + tuple(rows.shape[1:]), dtype=rows.dtype, device=rows.device)
        dist.all_to_all_single(out, rows.contiguous(), output_split_sizes=out_splits, input_split_sizes=in_splits, group=self.group)
        return out, recv_counts

    # ---- the training exchange: kept rows only ----
    def plan(self, kept: torch.Tensor, recv_counts: torch.Tensor):
        """ONE host read per forward pass (not one per segment): kept [n_seg, E] = rows this rank sends per (segment, expert),
        recv_counts [n_seg, W * E_local] = rows it receives per (segment, source rank, local expert).  Returns the per-segment split
        sizes and row offsets of the packed send / receive spaces: dict(in_splits, out_splits [n_seg][W], send_off, recv_off [n_seg + 1])."""
        n_seg = kept.shape[0]
        both = torch.cat([kept.view(n_seg, self.world, self.El).sum(2), recv_counts.view(n_seg, self.world, self.El).sum(2)], 1)
        host = both.to("cpu", torch.int64).tolist()          # the one synchronisation point of the expert-parallel step
        ins = [row[: self.world] for row in host]
        outs = [row[self.world:] for row in host]
        so, ro = [0], [0]
        for s_ in range(n_seg):
            so.append(so[-1] + sum(ins[s_]))
            ro.append(ro[-1] + sum(outs[s_]))
        return dict(in_splits=ins, out_splits=outs, send_off=so, recv_off=ro)

    profile = False          # bench.py --parallelism ep: record HIP events around the collectives and the waits for them

    @staticmethod
    def _capturing() -> bool:
        """Inside a hipGraph capture the all-to-all goes out on the CAPTURING stream, blocking form.  Measured on RCCL 2.26.6 / torch
        2.10 + ROCm 7.0 (scripts/experiments/rccl_capture_probe.py, profiles/r06_experiments.md 2): all_reduce / all_gather /
        reduce_scatter and a blocking all_to_all_single on the capturing stream capture and replay correctly; an all-to-all issued with
        async_op=True, or on a stream forked from the capturing one, crashes hipStreamEndCapture (SIGSEGV).  The captured (padded) step
        therefore serialises its exchanges with the expert launches; the eager step keeps the side-stream overlap."""
        return torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()

    def all_to_all_v(self, send: torch.Tensor, in_splits, recv: torch.Tensor, out_splits, stream=None):
        """Unequal-split all-to-all of packed rows (dim 0): chunk r of `send` (in_splits[r] rows) goes to rank r, `recv` receives
        out_splits[w] rows from rank w.  Only rows that exist travel (20 % fewer bytes than the capacity-padded payload at 80 % kept
        rows).  Same stream semantics as all_to_all; returns wait()."""
        if self.local:
            assert recv.data_ptr() == send.data_ptr()
            return lambda: None
        row_bytes = (send.numel() // max(1, send.shape[0])) * send.element_size()
        self.bytes_sent = getattr(self, "bytes_sent", 0) + (sum(in_splits) - in_splits[self.rank]) * row_bytes     # rows that leave this GPU
        if send.is_cuda and self._capturing():
            dist.all_to_all_single(recv, send, output_split_sizes=list(out_splits), input_split_sizes=list(in_splits), group=self.group)
            return lambda: None
        if stream is None or not send.is_cuda:
            bev = None
            if self.profile and send.is_cuda:      # (the blocking form on the launch stream - the owner-tail mode: nothing of it is hidden)
                bev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                bev[0].record()
            work = dist.all_to_all_single(recv, send, output_split_sizes=list(out_splits), input_split_sizes=list(in_splits), group=self.group,
                                          async_op=True)
            if bev is None:
                return work.wait

            def wait_blocking():
                work.wait()
                bev[1].record()
                self.__dict__.setdefault("ev_coll", []).append(bev)
                self.__dict__.setdefault("ev_wait", []).append(bev)
            return wait_blocking
        ready = torch.cuda.Event()
        ready.record()
        with torch.cuda.stream(stream):
            stream.wait_event(ready)
            ev = None
            if self.profile:
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record()
            work = dist.all_to_all_single(recv, send, output_split_sizes=list(out_splits), input_split_sizes=list(in_splits), group=self.group,
                                          async_op=True)
            if ev is not None:
                work.wait()
                ev[1].record()
                self.__dict__.setdefault("ev_coll", []).append(ev)

        def wait():
            wv = None
            if self.profile:
                wv = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                wv[0].record()
            work.wait()                       # orders the consumer's (current) stream after the collective
            if wv is not None:
                wv[1].record()
                self.__dict__.setdefault("ev_wait", []).append(wv)
            send.record_stream(torch.cuda.current_stream())
        return wait

    def overlap_report(self):
        """(after a synchronize) total time of the profiled collectives on the side stream, total time the consumer stream spent
        waiting for them, and the fraction of the exchange that was hidden behind compute: 1 - wait / collective."""
        coll = sum(a.elapsed_time(b) for a, b in self.__dict__.get("ev_coll", []))
        wait = sum(a.elapsed_time(b) for a, b in self.__dict__.get("ev_wait", []))
        n = len(self.__dict__.get("ev_coll", []))
        self.__dict__["ev_coll"], self.__dict__["ev_wait"] = [], []
        return dict(collectives=n, collective_ms=coll, wait_ms=wait, hidden_fraction=(1.0 - wait / coll) if coll > 0 else None)

    # ---- the collective ----
    def all_to_all(self, send: torch.Tensor, stream=None, out: Optional[torch.Tensor] = None):
        """Equal-split all-to-all over dim 0 (world chunks).  Returns (recv, wait): call wait() on the stream that consumes
        recv.  With `stream` (a side HIP stream) the collective is ordered after the work already queued on the current
        stream and runs concurrently with what the caller queues next.  out: receive buffer (same shape; world == 1: must be
        `send` itself or None - nothing moves)."""
        if self.local:
            assert out is None or out.data_ptr() == send.data_ptr()
            return send, (lambda: None)
        recv = torch.empty_like(send) if out is None else out
        if send.is_cuda and self._capturing():
            dist.all_to_all_single(recv, send, group=self.group)
            return recv, (lambda: None)
        if stream is None or not send.is_cuda:
            work = dist.all_to_all_single(recv, send, group=self.group, async_op=True)
            return recv, work.wait
        ready = torch.cuda.Event()
        ready.record()
        with torch.cuda.stream(stream):
            stream.wait_event(ready)
            work = dist.all_to_all_single(recv, send, group=self.group, async_op=True)

        def wait():
            work.wait()                       # orders the consumer's (current) stream after the collective
            send.record_stream(torch.cuda.current_stream())
        return recv, wait
