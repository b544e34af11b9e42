"""Data parallelism for the fusion train step: one process per GPU, `torch.distributed`
(`nccl` backend == RCCL over xGMI on ROCm; `gloo` on CPU for tests).

The reference is single-process (forensic_trainer.py:232-234 is a plain DataLoader).  Samples
are independent through the whole path (no BatchNorm, CE is a batch mean, gnn_Z is a constant table).  Every rank
takes `TrainConfig.batch_size` rows per step from its own shard of the (shuffled) split -- WEAK scaling: the global
batch is world x batch_size at an unchanged learning rate, so a reference configuration run at world > 1 does not retrace
the single-process optimisation trajectory (divide batch_size by the world size for that) -- and the ONLY exchange is the
sum of the flat gradient arena (12.75 M fp32 = 51 MB; the frozen encoders exchange nothing).  Ranks sum; the 1/world
factor is folded into `grad_scale` of the device step state, so the clip (which must see the reduced gradient,
forensic_trainer.py:292-297) and AdamW read the mean without another pass.

Overlap with backward.  The arena is laid out in gradient-ready order [classifier | fuse_mlp | co-attention |
projections] and backward runs in two phases (ufnd_fusion_backward_phase): as soon as the fuse_mlp weight
gradients are written -- 36 MB of the 51, two thirds of them `fuse_mlp.0.weight` -- bucket 0 = [classifier |
fuse_mlp] starts reducing on the process group's own stream while the rest of backward (co-attention, stacked
q/k/v, projections) runs; bucket 1 = the remainder follows.  Both are asynchronous collectives issued from the
step's stream (torch makes the group's stream wait for the work enqueued so far); `finish()` makes the step's
stream wait for them before the clip.  With the encoders inside the step, the next batch's frozen encoder
forwards are already enqueued and hide what is left.

Exchange variants (`GradReducer(payload=, algorithm=)`, `TrainConfig.grad_payload / grad_exchange`):
  payload "fp32" (default) sums the gradient arena in place; "bf16" rounds each bucket to bf16 (25.5 MB on the wire), sums
  that, and widens the sum back into the fp32 arena -- the fp32 master weights, the moments and the clip are untouched, the
  exchanged sum carries 8 significant bits per rank's addend;
  algorithm "all_reduce" (default) leaves the schedule to RCCL; "rs_ag" issues reduce-scatter + all-gather per bucket (each
  rank reduces 1/world of the bucket over all of its xGMI links, then gathers the reduced shards: the direct form of
  SURVEY.md 8e for a point-to-point fabric).

  algorithm "factors" (`FactorExchange`, SURVEY.md 5: the head-only step is communication-bound -- a 51 MB ring all-reduce
  against a 0.25 ms step) does not move the head's Linear gradients at all: every one of them is dW = dY^T X over the batch
  rows, so each rank all-gathers its FACTOR panels (2.5 MB at B = 32) and forms the summed dW / db over all ranks' rows
  locally, in rank order (ufnd_head_linear_grads_from_factors); the 21 k floats of other gradients are all-reduced as before.
  `.grad` keeps its meaning (the sum over ranks), the wire carries 20 x fewer bytes per rank; the summation order of a dW
  element differs from the all-reduce's (one chain over world x B rows instead of a sum of per-rank chains).

Every collective goes through a `Collectives` object, which applies torch.distributed to tensors WHERE THEY LIVE: device
tensors need a backend that reduces device memory (nccl).  There is no host staging in this package.
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


class Collectives:
    """The torch.distributed calls of the trainer, on one process group.  Tensors are reduced / gathered / broadcast where
    they live; a subclass may route them differently (tests do, to run two ranks on one GPU over gloo)."""

    def __init__(self, group=None):
        self.group = group
        on = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if on else 1
        self.rank = dist.get_rank(group) if on else 0
        self.initialized = on

    @property
    def backend(self) -> Optional[str]:
        return dist.get_backend(self.group) if self.initialized else None

    def all_reduce_async(self, t: torch.Tensor):
        """Start summing `t` in place over the ranks; returns an object with .wait() (stream-ordered for device tensors)."""
        return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def reduce_scatter_async(self, out: torch.Tensor, t: torch.Tensor):
        return dist.reduce_scatter_tensor(out, t, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def all_gather_into_async(self, out: torch.Tensor, t: torch.Tensor):
        return dist.all_gather_into_tensor(out, t, group=self.group, async_op=True)

    def all_reduce(self, t: torch.Tensor, op=None) -> torch.Tensor:
        dist.all_reduce(t, op=op if op is not None else dist.ReduceOp.SUM, group=self.group)
        return t

    def all_gather(self, t: torch.Tensor) -> List[torch.Tensor]:
        out = [torch.empty_like(t) for _ in range(self.world)]
        dist.all_gather(out, t, group=self.group)
        return out

    def broadcast(self, t: torch.Tensor, src: int = 0) -> None:
        dist.broadcast(t, src=dist.get_global_rank(self.group, src) if self.group is not None else src, group=self.group)

    def barrier(self) -> None:
        dist.barrier(group=self.group)


def as_comm(group=None) -> Collectives:
    """`group` arguments of this package take a process group (None = the default one) or a Collectives."""
    return group if isinstance(group, Collectives) else Collectives(group)


def world_info(group=None):
    c = as_comm(group)
    return c.world, c.rank


class _Joined:
    """Several stream-ordered steps of one bucket's exchange behind one .wait()."""

    def __init__(self, works, after=None):
        self.works, self.after = [w for w in works if w is not None], after

    def wait(self):
        for w in self.works:
            w.wait()
        if self.after is not None:
            self.after()


class GradReducer:
    """Bucketed sum of a flat gradient buffer over the ranks, overlapped with the producer of the later buckets.

    `bounds` are the bucket boundaries in elements (ascending, inside the buffer): buckets are
    [0, bounds[0]), [bounds[0], bounds[1]), ..., [bounds[-1], n).  `start(k)` begins reducing bucket k -- its
    gradients must be complete on the current stream; `finish()` makes the current stream wait for every started
    bucket.  Nothing happens at world size 1 unless force=True (runs the collectives anyway: the RCCL path on one GPU).
    payload / algorithm: module docstring."""

    def __init__(self, grad: torch.Tensor, group=None, bounds: Sequence[int] = (), force: bool = False,
                 payload: str = "fp32", algorithm: str = "all_reduce"):
        if payload not in ("fp32", "bf16") or algorithm not in ("all_reduce", "rs_ag"):
            raise ValueError(f"GradReducer: payload={payload!r} (fp32 | bf16), algorithm={algorithm!r} (all_reduce | rs_ag)")
        self.grad, self.comm = grad, as_comm(group)
        self.group = self.comm.group
        self.world, self.rank = self.comm.world, self.comm.rank
        self.payload, self.algorithm = payload, algorithm
        n = grad.numel()
        cuts = [0] + [int(b) for b in bounds] + [n]
        if any(b <= a for a, b in zip(cuts, cuts[1:])):
            raise ValueError(f"bucket bounds {list(bounds)} must be ascending inside (0, {n})")
        self.buckets: List[Tuple[int, int]] = list(zip(cuts[:-1], cuts[1:]))
        self.force = bool(force) and self.comm.initialized
        self._pending: list = []
        self._wire: Optional[torch.Tensor] = None        # bf16 payload: the rounded buckets (one buffer, bucket layout)
        self._shard: dict = {}                           # rs_ag: the reduced shard of each bucket

    @property
    def active(self) -> bool:
        return self.world > 1 or self.force

    @property
    def grad_scale(self) -> float:
        return 1.0 / self.world

    def wire_bytes(self) -> int:
        """Bytes one rank contributes per step (the payload the fabric carries)."""
        return self.grad.numel() * (2 if self.payload == "bf16" else 4)

    def _exchange(self, t: torch.Tensor, key):
        """Sum `t` in place over the ranks; returns something with .wait()."""
        if self.algorithm == "all_reduce":
            return self.comm.all_reduce_async(t)
        # reduce-scatter + all-gather over a buffer padded to a multiple of the world size
        w, n = self.world, t.numel()
        per = (n + w - 1) // w
        ent = self._shard.get(key)
        if ent is None or ent[0].dtype != t.dtype:
            ent = self._shard[key] = (torch.zeros(per * w, dtype=t.dtype, device=t.device), torch.empty(per, dtype=t.dtype, device=t.device))
        full, shard = ent
        full[:n].copy_(t)
        w1 = self.comm.reduce_scatter_async(shard, full)
        if w1 is not None:
            w1.wait()                                   # stream-ordered: the gather reads the reduced shard
        w2 = self.comm.all_gather_into_async(full, shard)
        return _Joined([w2], after=lambda: t.copy_(full[:n]))

    def _reduce(self, lo: int, hi: int):
        g = self.grad[lo:hi]
        if self.payload == "fp32":
            return self._exchange(g, (lo, hi))
        if self._wire is None:
            self._wire = torch.empty(self.grad.numel(), dtype=torch.bfloat16, device=self.grad.device)
        wire = self._wire[lo:hi]
        wire.copy_(g)                                   # round to nearest even, stream-ordered
        w = self._exchange(wire, (lo, hi))
        return _Joined([w], after=lambda: g.copy_(wire))

    def start(self, k: Optional[int] = None) -> None:
        """Begin reducing bucket k (None: every bucket, in order)."""
        if not self.active:
            return
        for lo, hi in (self.buckets if k is None else [self.buckets[k]]):
            w = self._reduce(lo, hi)
            if w is not None:
                self._pending.append(w)

    def finish(self) -> None:
        """Make the current stream wait for the buckets started so far."""
        for w in self._pending:
            w.wait()
        self._pending = []


def _complement(ranges: Sequence[Tuple[int, int]], lo: int, hi: int) -> List[Tuple[int, int]]:
    """[lo, hi) minus the (sorted, merged) `ranges`."""
    out, at = [], lo
    for a, b in sorted(ranges):
        if a < lo or b > hi or a < at:
            raise ValueError(f"range ({a}, {b}) overlaps its predecessor or leaves [{lo}, {hi})")
        if a > at:
            out.append((at, a))
        at = max(at, b)
    if at < hi:
        out.append((at, hi))
    return out


class FactorExchange(GradReducer):
    """The head's gradient exchange in FACTOR form (module docstring, algorithm "factors").

    `linear_ranges`: the [begin, end) float ranges of the gradient buffer that the factor product writes (the Linear layers'
    weights and biases); everything else inside the head's buckets 0 and 1 -- [0, bounds[1]) or the whole buffer when there
    are no further buckets -- is summed by plain all-reduces; buckets 2... (trainable encoders) keep the parent's all-reduce.
    The owner of the step calls `start_factors(pack, form)` once its backward (run WITHOUT the Linear products) has packed the
    rank's factor panels into `pack`: the packs are all-gathered, and at `finish()` `form(packs, stride, ranks)` is called on
    the current stream to write the summed gradients (the HIP entry in the product; tests pass their own)."""

    factors = True

    def __init__(self, grad: torch.Tensor, group=None, bounds: Sequence[int] = (), force: bool = False,
                 linear_ranges: Sequence[Tuple[int, int]] = ()):
        super().__init__(grad, group, bounds, force, payload="fp32", algorithm="all_reduce")
        head_end = self.buckets[1][1] if len(self.buckets) > 1 else grad.numel()
        self.head_end = head_end
        self.linear_ranges = sorted((int(a), int(b)) for a, b in linear_ranges)
        self.small_ranges = _complement(self.linear_ranges, 0, head_end)
        self._recv: Optional[torch.Tensor] = None
        self._pack_floats = 0

    def start(self, k: Optional[int] = None) -> None:
        """Buckets 0 and 1 (the head) are exchanged by start_factors(); later buckets as in the parent."""
        if not self.active:
            return
        for i in (range(len(self.buckets)) if k is None else [k]):
            if i >= 2:
                super().start(i)

    def start_factors(self, pack: torch.Tensor, form) -> None:
        if not self.active:
            raise RuntimeError("FactorExchange.start_factors without an active exchange (world 1 and not forced): run the plain backward")
        n = pack.numel()
        if self._recv is None or self._recv.numel() != n * self.world or self._recv.device != pack.device:
            self._recv = torch.empty(n * self.world, dtype=pack.dtype, device=pack.device)
        self._pack_floats = n
        recv, world = self._recv, self.world
        for lo, hi in self.small_ranges:
            w = self.comm.all_reduce_async(self.grad[lo:hi])
            if w is not None:
                self._pending.append(w)
        w = self.comm.all_gather_into_async(recv, pack)
        self._pending.append(_Joined([w], after=lambda: form(recv, n, world)))

    def wire_bytes(self) -> int:
        """Bytes one rank contributes per step: its factor pack, the small all-reduced ranges and the later buckets."""
        small = sum(hi - lo for lo, hi in self.small_ranges)
        later = sum(hi - lo for lo, hi in self.buckets[2:])
        return 4 * (self._pack_floats + small + later)


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
    comm = as_comm(group)
    if comm.world == 1:
        return t
    t = t.contiguous()
    n = torch.tensor([t.shape[0]], dtype=torch.int64, device=t.device)
    counts = [int(c.item()) for c in comm.all_gather(n)]
    mx = max(counts)
    if t.shape[0] < mx:
        t = torch.cat([t, t.new_zeros((mx - t.shape[0],) + tuple(t.shape[1:]))], 0)
    out = comm.all_gather(t)
    return torch.cat([o[:c] for o, c in zip(out, counts)], 0)


def gather_epoch_outputs(y: torch.Tensor, p1: torch.Tensor, forensic: torch.Tensor, loss_sum: torch.Tensor, n_batches: int,
                         group=None):
    """Every rank's (labels, P(class 1), forensic (3, rows)) rows and the mean of the per-batch losses over ALL ranks'
    batches -- what the epoch metrics of forensic_trainer.py:316-328 are computed from."""
    comm = as_comm(group)
    if comm.world == 1:
        return y, p1, forensic, loss_sum / max(1, n_batches)
    y = gather_rows(y, comm)
    p1 = gather_rows(p1, comm)
    forensic = gather_rows(forensic.t().contiguous(), comm).t()
    acc = torch.stack([loss_sum.reshape(()).to(torch.float64), torch.tensor(float(n_batches), dtype=torch.float64, device=loss_sum.device)])
    comm.all_reduce(acc)
    return y, p1, forensic, (acc[0] / acc[1].clamp_min(1.0)).to(torch.float32).to(y.device)


def save_checkpoint(obj: dict, path: str, group=None) -> None:
    """Rank 0 writes `path` atomically (temporary file + rename); every rank returns only once it is complete."""
    comm = as_comm(group)
    if comm.rank == 0:
        tmp = f"{path}.tmp.{os.getpid()}"
        torch.save(obj, tmp)
        os.replace(tmp, path)
    if comm.world > 1:
        comm.barrier()


def broadcast_from_rank0(t: torch.Tensor, group=None) -> None:
    """Every rank continues with rank 0's values (after rank 0 loaded a checkpoint)."""
    comm = as_comm(group)
    if comm.world > 1:
        comm.broadcast(t, 0)
