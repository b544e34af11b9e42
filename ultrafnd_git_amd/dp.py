"""Data parallelism for the fusion train step: one process per GPU, `torch.distributed`
(`nccl` backend == RCCL over xGMI on ROCm; `gloo` on CPU for tests).

The reference is single-process (forensic_trainer.py:232-234 is a plain DataLoader).  Samples
are independent through the whole path (no BatchNorm, CE is a batch mean, gnn_Z is a constant table), so the
global batch is sharded across ranks and the ONLY exchange is the sum of the flat gradient arena (12.75 M fp32 =
51 MB; the encoders are frozen and exchange nothing).  Ranks sum; the 1/world factor is folded into `grad_scale`
of the device step state, so the clip (which must see the reduced gradient, forensic_trainer.py:292-297) and AdamW
read the mean without another pass.

Overlap with backward.  The arena is laid out in gradient-ready order [classifier | fuse_mlp | co-attention |
projections] and backward runs in two phases (ufnd_fusion_backward_phase): as soon as the fuse_mlp weight
gradients are written -- 36 MB of the 51, two thirds of them `fuse_mlp.0.weight` -- bucket 0 = [classifier |
fuse_mlp] starts reducing on the process group's own stream while the rest of backward (co-attention, stacked
q/k/v, projections) runs; bucket 1 = the remainder follows.  Both are asynchronous collectives issued from the
step's stream (torch makes the group's stream wait for the work enqueued so far); `finish()` makes the step's
stream wait for them before the clip.  With the encoders inside the step, the next batch's frozen encoder
forwards are already enqueued and hide what is left.
"""
from __future__ import annotations

import os
from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def init_process_group(device: torch.device) -> None:
    """`nccl` (= RCCL) process group whose internal stream is HIGH priority: the gradient exchange sits on the serial
    head -> exchange -> optimizer chain while two encoder graphs keep every CU busy; at normal priority its
    kernels queue for CUs behind whole-CU GEMM blocks."""
    opts = None
    try:
        opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
    except Exception:          # (older torch without the option: default priority)
        opts = None
    if opts is not None:
        dist.init_process_group("nccl", device_id=device, pg_options=opts)
    else:
        dist.init_process_group("nccl", device_id=device)


def world_info(group=None):
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(group), dist.get_rank(group)
    return 1, 0


class GradReducer:
    """Bucketed sum-all-reduce of a flat gradient buffer, overlapped with the producer of the later buckets.

    `bounds` are the bucket boundaries in elements (ascending, inside the buffer): buckets are
    [0, bounds[0]), [bounds[0], bounds[1]), ..., [bounds[-1], n).  `start(k)` begins reducing bucket k -- its
    gradients must be complete on the current stream; `finish()` makes the current stream wait for every started
    bucket.  Nothing happens at world size 1 unless UFND_FORCE_REDUCE=1 (runs the collectives anyway: exercises the
    RCCL path on one GPU)."""

    def __init__(self, grad: torch.Tensor, group=None, bounds: Sequence[int] = ()):
        self.grad, self.group = grad, group
        self.world, self.rank = world_info(group)
        n = grad.numel()
        cuts = [0] + [int(b) for b in bounds] + [n]
        if any(b <= a for a, b in zip(cuts, cuts[1:])):
            raise ValueError(f"bucket bounds {list(bounds)} must be ascending inside (0, {n})")
        self.buckets: List[Tuple[int, int]] = list(zip(cuts[:-1], cuts[1:]))
        self.force = os.environ.get("UFND_FORCE_REDUCE", "0") == "1" and dist.is_available() and dist.is_initialized()
        self._pending: list = []
        # a process group that cannot reduce device memory (gloo without device support) goes through pinned host
        # memory -- a test seam (two ranks sharing one GPU), never the product path (nccl = RCCL)
        self._via_host: Optional[bool] = None
        self._host: Optional[torch.Tensor] = None

    @property
    def active(self) -> bool:
        return self.world > 1 or self.force

    @property
    def grad_scale(self) -> float:
        return 1.0 / self.world

    def _reduce(self, t: torch.Tensor):
        if t.device.type == "cuda" and dist.get_backend(self.group) != "nccl":
            if self._via_host is None:
                try:
                    probe = torch.zeros(4, device=t.device)
                    dist.all_reduce(probe, group=self.group)
                    self._via_host = False
                except Exception:
                    self._via_host = True
            if self._via_host:
                if self._host is None:
                    self._host = torch.empty(self.grad.numel(), dtype=self.grad.dtype, pin_memory=True)
                lo = t.storage_offset() - self.grad.storage_offset()
                h = self._host[lo:lo + t.numel()]
                h.copy_(t)                       # (synchronises with the current stream)
                dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
                t.copy_(h, non_blocking=True)
                return None
        return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def start(self, k: Optional[int] = None) -> None:
        """Begin reducing bucket k (None: every bucket, in order)."""
        if not self.active:
            return
        for lo, hi in (self.buckets if k is None else [self.buckets[k]]):
            w = self._reduce(self.grad[lo:hi])
            if w is not None:
                self._pending.append(w)

    def finish(self) -> None:
        """Make the current stream wait for the buckets started so far."""
        for w in self._pending:
            w.wait()
        self._pending = []


def _needs_host(t: torch.Tensor, group=None) -> bool:
    return t.device.type == "cuda" and dist.get_backend(group) != "nccl"


def shard_indices(n: int, world: int, rank: int, perm: Optional[torch.Tensor] = None, pad: bool = True) -> torch.Tensor:
    """DistributedSampler-style split of the (permuted) index list: every world-th index starting at `rank`.
    pad=True (training: every rank must take the same number of steps) first wraps the list to a multiple of
    `world`, so every rank gets ceil(n/world); pad=False (validation / test: no per-step collective) leaves the
    shards uneven, so that no sample is counted twice in the epoch metrics."""
    idx = perm if perm is not None else torch.arange(n)
    if world == 1:
        return idx
    if pad:
        total = (n + world - 1) // world * world
        if total > n:
            idx = idx.repeat((total + n - 1) // n)[:total]
        return idx[rank:total:world]
    return idx[rank::world]


def gather_rows(t: torch.Tensor, group=None) -> torch.Tensor:
    """all-gather of per-rank rows for epoch metrics (AUC is not decomposable); row counts may differ per rank."""
    world, _ = world_info(group)
    if world == 1:
        return t
    if _needs_host(t, group):           # (test seam: a gloo group carrying device tensors)
        return gather_rows(t.cpu(), group).to(t.device)
    t = t.contiguous()
    n = torch.tensor([t.shape[0]], dtype=torch.int64, device=t.device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n, group=group)
    counts = [int(c.item()) for c in counts]
    mx = max(counts)
    if t.shape[0] < mx:
        t = torch.cat([t, t.new_zeros((mx - t.shape[0],) + tuple(t.shape[1:]))], 0)
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t, group=group)
    return torch.cat([o[:c] for o, c in zip(out, counts)], 0)


def gather_epoch_outputs(y: torch.Tensor, p1: torch.Tensor, forensic: torch.Tensor, loss_sum: torch.Tensor, n_batches: int,
                         group=None):
    """Every rank's (labels, P(class 1), forensic (3, rows)) rows and the mean of the per-batch losses over ALL ranks'
    batches -- what the epoch metrics of forensic_trainer.py:316-328 are computed from."""
    world, _ = world_info(group)
    if world == 1:
        return y, p1, forensic, loss_sum / max(1, n_batches)
    y = gather_rows(y, group)
    p1 = gather_rows(p1, group)
    forensic = gather_rows(forensic.t().contiguous(), group).t()
    if _needs_host(loss_sum, group):
        loss_sum = loss_sum.cpu()
    acc = torch.stack([loss_sum.reshape(()).to(torch.float64), torch.tensor(float(n_batches), dtype=torch.float64, device=loss_sum.device)])
    dist.all_reduce(acc, group=group)
    return y, p1, forensic, (acc[0] / acc[1].clamp_min(1.0)).to(torch.float32).to(y.device)


def save_checkpoint(obj: dict, path: str, group=None) -> None:
    """Rank 0 writes `path` atomically (temporary file + rename); every rank returns only once it is complete."""
    world, rank = world_info(group)
    if rank == 0:
        tmp = f"{path}.tmp.{os.getpid()}"
        torch.save(obj, tmp)
        os.replace(tmp, path)
    if world > 1:
        dist.barrier(group=group)


def broadcast_from_rank0(t: torch.Tensor, group=None) -> None:
    """Every rank continues with rank 0's values (after rank 0 loaded a checkpoint)."""
    world, _ = world_info(group)
    if world > 1:
        src = dist.get_global_rank(group, 0) if group is not None else 0
        if t.device.type == "cuda" and dist.get_backend(group) != "nccl":
            h = t.cpu()
            dist.broadcast(h, src=src, group=group)
            t.copy_(h)
        else:
            dist.broadcast(t, src=src, group=group)
