"""Flat fp32 parameter arena.

The reference hands ~117 separate tensors to clip_grad_norm_ and AdamW (56 % of its step is
per-tensor bookkeeping).  Here every trainable tensor of the fusion head is
a view into ONE flat HBM buffer, with gradients and both Adam moments in three more buffers of
the same layout: the norm, the optimizer and the data-parallel all-reduce each touch one
contiguous range.  Layout = [with-grad groups ... | no-grad groups ...]; every group starts on
a 256-byte boundary; tensors inside a group are packed back to back because the kernels read
some groups as one stacked matrix (include/ultrafnd_hip.h).
"""
from __future__ import annotations

from typing import Dict, Iterable, List, Sequence, Tuple

import torch
import torch.nn as nn

ALIGN = 64  # floats (256 B)

# (key, shape) lists; a group is packed contiguously
Group = List[Tuple[str, Tuple[int, ...]]]


def _numel(shape: Sequence[int]) -> int:
    n = 1
    for d in shape:
        n *= int(d)
    return n


class FlatArena:
    def __init__(self, grad_groups: List[Group], nograd_groups: List[Group], device: torch.device):
        self.device = torch.device(device)
        self.offsets: Dict[str, Tuple[int, Tuple[int, ...]]] = {}
        o = 0
        for groups in (grad_groups, nograd_groups):
            for g in groups:
                o = (o + ALIGN - 1) // ALIGN * ALIGN
                for key, shape in g:
                    if key in self.offsets:
                        raise ValueError(f"duplicate arena key {key}")
                    self.offsets[key] = (o, tuple(shape))
                    o += _numel(shape)
            o = (o + ALIGN - 1) // ALIGN * ALIGN
            if groups is grad_groups:
                self.n_grad = o          # floats in the with-grad region (multiple of 64)
        self.n_total = o
        self.grad_keys = [k for g in grad_groups for k, _ in g]
        self.data = torch.zeros(self.n_total, dtype=torch.float32, device=self.device)
        self.grad: torch.Tensor | None = None
        self.exp_avg: torch.Tensor | None = None
        self.exp_avg_sq: torch.Tensor | None = None

    # views
    def _v(self, buf: torch.Tensor, key: str) -> torch.Tensor:
        o, shape = self.offsets[key]
        return buf[o:o + _numel(shape)].view(shape)

    def view(self, key: str) -> torch.Tensor:
        return self._v(self.data, key)

    def ensure_grad(self) -> torch.Tensor:
        if self.grad is None:
            self.grad = torch.zeros(self.n_grad, dtype=torch.float32, device=self.device)
        return self.grad

    def grad_view(self, key: str) -> torch.Tensor:
        o, shape = self.offsets[key]
        if o >= self.n_grad:
            raise KeyError(f"{key} lives in the no-grad region")
        return self.ensure_grad()[o:o + _numel(shape)].view(shape)

    def has_grad(self, key: str) -> bool:
        return self.offsets[key][0] < self.n_grad

    def ensure_moments(self):
        if self.exp_avg is None:
            self.exp_avg = torch.zeros(self.n_grad, dtype=torch.float32, device=self.device)
            self.exp_avg_sq = torch.zeros(self.n_grad, dtype=torch.float32, device=self.device)
        return self.exp_avg, self.exp_avg_sq

    def span(self, keys: Iterable[str]) -> Tuple[int, int]:
        """[begin, end) float range covering `keys` (rounded out to the 64-float grid)."""
        lo = min(self.offsets[k][0] for k in keys)
        hi = max(self.offsets[k][0] + _numel(self.offsets[k][1]) for k in keys)
        return lo // ALIGN * ALIGN, (hi + ALIGN - 1) // ALIGN * ALIGN


class ArenaModule(nn.Module):
    """nn.Module whose parameters are views into a FlatArena.

    Subclasses implement `_arena_groups() -> (grad_groups, nograd_groups)` with keys equal to
    their own `named_parameters()` names.  `rehome([modules], prefixes)` builds one joint arena.
    """

    _arena: FlatArena | None = None
    _arena_prefix: str = ""

    def _arena_groups(self) -> Tuple[List[Group], List[Group]]:
        raise NotImplementedError

    def _on_rehome(self) -> None:  # subclasses drop cached pointer tables here
        pass

    def _apply(self, fn, recurse=True):
        # .to()/.cuda()/.float() replace every parameter's storage: re-flatten afterwards
        out = super()._apply(fn, recurse)
        rehome([self], [""])
        return out

    def akey(self, name: str) -> str:
        return self._arena_prefix + name

    def aview(self, name: str) -> torch.Tensor:
        return self._arena.view(self.akey(name))

    def gview(self, name: str) -> torch.Tensor:
        return self._arena.grad_view(self.akey(name))


def rehome(modules: Sequence[ArenaModule], prefixes: Sequence[str], extra_grad_groups: Sequence[Group] = ()) -> FlatArena:
    """Move the parameters of `modules` into one new FlatArena (values preserved) and re-point
    every nn.Parameter at its view.  Grad region order = modules in the given order, then `extra_grad_groups` (already
    prefixed keys of tensors that are not nn.Parameters: the trainable encoders' masters, bound by their owners)."""
    device = next(modules[0].parameters()).device
    grad_groups: List[Group] = []
    nograd_groups: List[Group] = []
    for m, pre in zip(modules, prefixes):
        gg, ng = m._arena_groups()
        grad_groups += [[(pre + k, s) for k, s in g] for g in gg]
        nograd_groups += [[(pre + k, s) for k, s in g] for g in ng]
    grad_groups += [list(g) for g in extra_grad_groups]
    arena = FlatArena(grad_groups, nograd_groups, device)
    for m, pre in zip(modules, prefixes):
        named = dict(m.named_parameters())
        laid = {k[len(pre):] for k in arena.offsets if k.startswith(pre)} if pre else set(arena.offsets)
        missing = set(named) - laid
        if missing:
            raise RuntimeError(f"parameters without an arena slot: {sorted(missing)}")
        with torch.no_grad():
            for name, p in named.items():
                v = arena.view(pre + name)
                if tuple(v.shape) != tuple(p.shape):
                    raise RuntimeError(f"arena shape mismatch for {name}: {tuple(v.shape)} vs {tuple(p.shape)}")
                v.copy_(p.detach().to(device=device, dtype=torch.float32))
                p.data = v
                p.grad = None
        m._arena = arena
        m._arena_prefix = pre
        m._on_rehome()
    return arena
