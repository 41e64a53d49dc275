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


def init_from_env(backend: Optional[str] = None, device: Optional[torch.device] = None):
    """Initialise torch.distributed from torchrun's environment (RANK / WORLD_SIZE / MASTER_*).  Returns (rank, world)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this driver
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = {}
        if backend == "nccl" and device is not None:
            kw["device_id"] = device
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world


def shard_rays(n_rays: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous per-rank slice [begin, end) of a global ray batch (runner.py:575: batch_size // world_size each;
    the remainder, if any, is dropped like DataLoader(drop_last) would)."""
    per = n_rays // world
    return rank * per, (rank + 1) * per


def make_grad_allreduce(group=None) -> Callable[[torch.Tensor], float]:
    """Returns f(flat_grad) -> grad_scale: sums the flat gradient buffer over ranks in place (one bucket) and returns
    the 1/world factor that swn_adam_step applies."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1

    def f(flat: torch.Tensor) -> float:
        if world > 1:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        return 1.0 / world
    return f



class ExpertParallel:
    """Expert-parallel token exchange: BASELINE.json configs[2] / the reference's optional EP mode
    (/root/reference/switch_nerf/modules/tutel_moe_ext/tutel_moe_layer_nobatch.py:157-185: all-to-all of the dispatched
    [W, E_local, C, M] payload before the experts and back after them; runner.py:97-101: E_local = E // W).

    Rank r owns experts [r * E_local, (r + 1) * E_local).  Every rank still routes ITS OWN points (capacity, ranking and
    l_aux stay rank-local exactly as in the data-parallel mode, so routing indices do not depend on the mode); only the
    rows that enter the expert MLP travel.

    The exchange is PER ROUTING SEGMENT (= the reference's per-model-chunk MoE call): the dispatched rows of one segment in their
    native order (expert, capacity slot) ARE the payload order (destination rank, local expert, slot), so the send buffer is one
    swn_gather_rows through the segment's routing permutation and the returned rows land in the native row space the combine
    gathers from - no index remapping.  all_to_all_single (equal splits: capacity-padded like the reference's batched path)
    delivers (source rank, local expert, slot) = the group order of the expert kernels with group % E_local = local expert (no
    kernel knows about ranks).  The valid-row counts of all segments travel once, ahead of the rows.
    Segments are pipelined: the exchange of segment s + 1 runs on a side HIP stream while the experts work on segment s, and the
    return of segment s overlaps both (model.py).  RCCL over xGMI is point-to-point: an all-to-all sends (W - 1) / W of the payload
    over the 7 links in parallel, there is nothing to gain from ring-style chunking.
    world == 1 degenerates to the identity (no process group needed): the single-GPU parity test runs this path.
    """

    def __init__(self, rank: int, world: int, n_experts: int, group=None):
        if n_experts % world:
            raise ValueError(f"expert parallelism needs world ({world}) to divide the expert count ({n_experts})")
        self.rank, self.world, self.E, self.El, self.group = rank, world, n_experts, n_experts // world, group

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
        if W == 1:
            return rows, group_counts
        if recv_counts is None:
            recv_counts = torch.empty_like(group_counts)
            dist.all_to_all_single(recv_counts, group_counts.contiguous(), group=self.group)
        in_splits = group_counts.view(W, El).sum(1).tolist()
        out_splits = recv_counts.view(W, El).sum(1).tolist()
        out = torch.empty((int(sum(out_splits)),) + tuple(rows.shape[1:]), dtype=rows.dtype, device=rows.device)
        dist.all_to_all_single(out, rows.contiguous(), output_split_sizes=out_splits, input_split_sizes=in_splits, group=self.group)
        return out, recv_counts

    # ---- the collective ----
    def all_to_all(self, send: torch.Tensor, stream=None, out: Optional[torch.Tensor] = None):
        """Equal-split all-to-all over dim 0 (world chunks).  Returns (recv, wait): call wait() on the stream that consumes
        recv.  With `stream` (a side HIP stream) the collective is ordered after the work already queued on the current
        stream and runs concurrently with what the caller queues next.  out: receive buffer (same shape; world == 1: must be
        `send` itself or None - nothing moves)."""
        if self.world == 1:
            assert out is None or out.data_ptr() == send.data_ptr()
            return send, (lambda: None)
        recv = torch.empty_like(send) if out is None else out
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
