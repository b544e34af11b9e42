"""Test-side collectives: two ranks that SHARE one GPU talk over gloo, which cannot reduce device memory, so device tensors
go through (pinned) host memory.  The product (ultrafnd_git_amd/dp.py) has no such path -- its Collectives apply
torch.distributed to tensors where they live (RCCL for device tensors); tests hand this subclass to the trainer as `group=`."""
import torch
import torch.distributed as dist

from ultrafnd_git_amd.dp import Collectives


class _Done:
    def wait(self):
        pass


class HostStagedCollectives(Collectives):
    def __init__(self, group=None):
        super().__init__(group)
        self.staged_calls = 0

    def _host(self, t: torch.Tensor) -> torch.Tensor:
        self.staged_calls += 1
        return t.detach().to("cpu")                       # (synchronises with the current stream)

    def all_reduce_async(self, t):
        if t.device.type != "cuda":
            return super().all_reduce_async(t)
        h = self._host(t)
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
        t.copy_(h)
        return _Done()

    def reduce_scatter_async(self, out, t):
        if t.device.type != "cuda":
            return super().reduce_scatter_async(out, t)
        h, ho = self._host(t), torch.empty(out.shape, dtype=out.dtype)
        dist.reduce_scatter_tensor(ho, h, op=dist.ReduceOp.SUM, group=self.group)
        out.copy_(ho)
        return _Done()

    def all_gather_into_async(self, out, t):
        if t.device.type != "cuda":
            return super().all_gather_into_async(out, t)
        h, ho = self._host(t), torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(ho, h, group=self.group)
        out.copy_(ho)
        return _Done()

    def all_reduce(self, t, op=None):
        if t.device.type != "cuda":
            return super().all_reduce(t, op)
        h = self._host(t)
        dist.all_reduce(h, op=op if op is not None else dist.ReduceOp.SUM, group=self.group)
        t.copy_(h)
        return t

    def all_gather(self, t):
        if t.device.type != "cuda":
            return super().all_gather(t)
        return [o.to(t.device) for o in super().all_gather(self._host(t))]

    def broadcast(self, t, src=0):
        if t.device.type != "cuda":
            return super().broadcast(t, src)
        h = self._host(t)
        super().broadcast(h, src)
        t.copy_(h)
