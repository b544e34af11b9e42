"""Data parallelism for the fusion train step: one process per GPU, `torch.distributed`
(`nccl` backend == RCCL over xGMI on ROCm; `gloo` on CPU for tests).

The reference is single-process.  Samples are independent through the whole
path (no BatchNorm, CE is a batch mean, gnn_Z is a constant table), so the global batch is
sharded across ranks and the ONLY exchange is one all-reduce per step over the flat gradient
arena (12.75 M fp32 = 51 MB, encoders are frozen and exchange nothing).  Ranks sum; the 1/world
factor is folded into `grad_scale` of the device step state, so the clip (which must see the
reduced gradient, forensic_trainer.py:292-297) and AdamW read the mean without another pass.

Overlap: the exchange is issued from the (high-priority) stream of the head -> exchange -> optimizer
chain; torch's process group runs it on its own internal stream and makes the issuing stream wait,
which is all that chain needs (a second user stream for it only added one more contender for the four
hardware queues: 14.85k vs 15.2k samples/s in the one-GPU rehearsal).  The next batch's frozen encoder
forwards (~2 ms of MFMA work on their own streams) are already enqueued and hide the exchange.
"""
from __future__ import annotations

import os
from typing import Optional

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
    """Sum-all-reduce of a flat gradient buffer on a side stream (device) or inline (CPU)."""

    def __init__(self, grad: torch.Tensor, group=None, buckets: Optional[list] = None):
        self.grad, self.group = grad, group
        self.world, self.rank = world_info(group)
        self.buckets = buckets or [(0, grad.numel())]
        # UFND_FORCE_REDUCE=1 runs the collective even at world size 1 (exercises the RCCL path on one GPU)
        self.force = os.environ.get("UFND_FORCE_REDUCE", "0") == "1" and dist.is_available() and dist.is_initialized()
        self.stream = None       # the collective is issued from the caller's stream (module docstring)
        self._pending = False

    @property
    def grad_scale(self) -> float:
        return 1.0 / self.world

    def start(self) -> None:
        """Begin reducing (asynchronously on a device); gradients must be complete on the
        current stream."""
        if self.world == 1 and not self.force:
            return
        if self.stream is None:
            for lo, hi in self.buckets:
                dist.all_reduce(self.grad[lo:hi], op=dist.ReduceOp.SUM, group=self.group)
            return
        self.stream.wait_stream(torch.cuda.current_stream(self.grad.device))
        with torch.cuda.stream(self.stream):
            for lo, hi in self.buckets:
                dist.all_reduce(self.grad[lo:hi], op=dist.ReduceOp.SUM, group=self.group)
        self._pending = True

    def finish(self) -> None:
        """Make the current stream wait for the reduce started by start()."""
        if self._pending:
            torch.cuda.current_stream(self.grad.device).wait_stream(self.stream)
            self._pending = False


def shard_indices(n: int, world: int, rank: int, perm: Optional[torch.Tensor] = None) -> torch.Tensor:
    """DistributedSampler-style split: pad the (permuted) index list by wrapping to a multiple of
    `world`, then take every world-th index starting at `rank`.  Every rank gets ceil(n/world)."""
    idx = perm if perm is not None else torch.arange(n)
    if world == 1:
        return idx
    total = (n + world - 1) // world * world
    if total > n:
        idx = idx.repeat((total + n - 1) // n)[:total]
    return idx[rank:total:world]


def gather_rows(t: torch.Tensor, group=None) -> torch.Tensor:
    """all-gather of per-rank rows (equal counts) for epoch metrics -- AUC is not decomposable."""
    world, _ = world_info(group)
    if world == 1:
        return t
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t.contiguous(), group=group)
    return torch.cat(out, 0)
