"""The integrated trainer variant's in-graph GNN, MI355X-native.

Mirrors
  src/models/gnn/gnn_model.py:7-41                    GNNModel(in_dim, hid, out_dim, dropout): lin1 -> A_norm -> ReLU -> dropout ->
                                                      A_norm -> lin2; same constructor, state_dict keys and same-seed initial weights
  src/training/forensic_trainer_integrated.py:77-98   build_adj_from_ocr_sets: weighted Jaccard adjacency of a mini-batch
The module's parameters live in a flat arena (ArenaModule), so the trainer can put it in the SAME arena as the fusion head and
the classifier: one gradient range for the norm, the clip, AdamW and the data-parallel exchange.  `forward` keeps what
`backward(d_z)` needs; there is no CPU path.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch
import torch.nn as nn

from . import _lib as L
from .arena import ArenaModule, Group, rehome
from .gcn import sets_to_csr
from .state import StepStateBuffer


def build_adj_from_ocr_sets(ocr_sets: Sequence[set], overlap_thresh: float = 0.12, device="cuda") -> torch.Tensor:
    """(N, N) fp32 adjacency on `device`: Jaccard(set_i, set_j) where >= overlap_thresh, i != j, both sets non-empty; zero
    diagonal (forensic_trainer_integrated.py:77-98; one launch instead of the O(N^2) Python loop)."""
    dev = torch.device(device)
    if dev.type != "cuda":
        raise L.UltrafndHipError("build_adj_from_ocr_sets runs on a HIP device only (no CPU fallback)")
    offs, toks = sets_to_csr(ocr_sets)
    n = len(ocr_sets)
    o = torch.from_numpy(offs).to(dev)
    t = torch.from_numpy(toks if toks.size else np.zeros(1, dtype=np.int32)).to(dev)
    adj = torch.empty(n, n, dtype=torch.float32, device=dev)
    L.check(L.lib().ufnd_ocr_adjacency_weighted(o.data_ptr(), t.data_ptr(), n, float(overlap_thresh), adj.data_ptr(), n,
                                                L.stream_ptr(dev)), "ufnd_ocr_adjacency_weighted")
    return adj


def batch_node_features(text: torch.Tensor, audio: torch.Tensor, visual: torch.Tensor, temporal: torch.Tensor,
                        out: torch.Tensor = None) -> torch.Tensor:
    """(B, 416) node features of a mini-batch: [T[:, :192], A[:, :32], V[:, :128], U[:, :64]], rows L2-normalised
    (forensic_trainer.py:193-195 -- the 416-wide feature the integrated variant's `gnn_in_dim = 416` names)."""
    dev = L.require_hip(text, audio, visual, temporal)
    t, a, v, u = L.f32c(text), L.f32c(audio), L.f32c(visual), L.f32c(temporal)
    B = t.shape[0]
    if out is None:
        out = torch.empty(B, 416, dtype=torch.float32, device=dev)
    L.check(L.lib().ufnd_node_features(t.data_ptr(), t.stride(0), a.data_ptr(), a.stride(0), v.data_ptr(), v.stride(0), u.data_ptr(),
                                       u.stride(0), 192, 32, 128, 64, B, out.data_ptr(), L.stream_ptr(dev)), "ufnd_node_features")
    return out


class GNNModel(ArenaModule):
    def __init__(self, in_dim: int, hid: int = 256, out_dim: int = 128, dropout: float = 0.2):
        super().__init__()
        if in_dim % 4 or hid % 32 or out_dim % 32:
            raise L.UltrafndHipError(f"GNNModel: in_dim={in_dim} must be a multiple of 4, hid={hid} / out_dim={out_dim} of 32")
        self.in_dim, self.hid, self.out_dim, self.dropout = int(in_dim), int(hid), int(out_dim), float(dropout)
        self.lin1 = nn.Linear(in_dim, hid)          # construction order == the reference's: same-seed initial weights
        self.lin2 = nn.Linear(hid, out_dim)
        self._ws: Dict[int, torch.Tensor] = {}
        self._rng = None
        self._last = None
        rehome([self], [""])

    def _arena_groups(self) -> Tuple[List[Group], List[Group]]:
        return [[("lin1.weight", (self.hid, self.in_dim))], [("lin1.bias", (self.hid,))], [("lin2.weight", (self.out_dim, self.hid))],
                [("lin2.bias", (self.out_dim,))]], []

    def _on_rehome(self) -> None:
        self._ws.clear()
        self._rng = None
        self._last = None

    def _params(self) -> L.GcnParams:
        p = L.GcnParams()
        p.w1, p.b1 = self.aview("lin1.weight").data_ptr(), self.aview("lin1.bias").data_ptr()
        p.w2, p.b2 = self.aview("lin2.weight").data_ptr(), self.aview("lin2.bias").data_ptr()
        return p

    def _workspace(self, n: int) -> torch.Tensor:
        if n not in self._ws:
            self._ws[n] = torch.empty(L.lib().ufnd_gnn_workspace_floats(n, self.in_dim, self.hid, self.out_dim), dtype=torch.float32,
                                      device=self._arena.device)
        return self._ws[n]

    @torch.no_grad()
    def forward(self, x: torch.Tensor, adj: torch.Tensor, state: StepStateBuffer = None, out: torch.Tensor = None) -> torch.Tensor:
        """Z (N, out_dim).  Train mode applies the dropout with the mask keyed by `state` (the trainer's step state, so that
        backward regenerates it); a module-owned counter state is used when none is given."""
        dev = L.require_hip(x, adj, self._arena.data)
        x, adj = L.f32c(x), L.f32c(adj)
        n = x.shape[0]
        if x.shape[1] != self.in_dim or tuple(adj.shape) != (n, n):
            raise RuntimeError(f"GNNModel: x {tuple(x.shape)} / adj {tuple(adj.shape)} do not match in_dim {self.in_dim}")
        p = self.dropout if self.training else 0.0
        if p > 0 and state is None:
            if self._rng is None:
                self._rng = StepStateBuffer(dev, seed=(torch.initial_seed() ^ 0x676E6E) & 0x7FFF_FFFF_FFFF_FFFF)
            else:
                self._rng.advance()
            state = self._rng
        z = out if out is not None else torch.empty(n, self.out_dim, dtype=torch.float32, device=dev)
        ws = self._workspace(n)
        L.check(L.lib().ufnd_gnn_forward(x.data_ptr(), adj.data_ptr(), adj.stride(0), C.byref(self._params()), z.data_ptr(), ws.data_ptr(), n,
                                         self.in_dim, self.hid, self.out_dim, p, state.ptr if p > 0 else None, L.stream_ptr(dev)),
                "ufnd_gnn_forward")
        self._last = (x, n, p, state)
        return z

    @torch.no_grad()
    def backward(self, d_z: torch.Tensor) -> None:
        """Parameter gradients (into the arena's gradient buffer) for a gradient d_z at the last forward's output."""
        if self._last is None:
            raise RuntimeError("GNNModel.backward: call forward first")
        x, n, p, state = self._last
        dev = self._arena.device
        d_z = L.f32c(d_z)
        self._arena.ensure_grad()
        L.check(L.lib().ufnd_gnn_backward(x.data_ptr(), C.byref(self._params()), self.gview("lin1.weight").data_ptr(),
                                          self.gview("lin1.bias").data_ptr(), self.gview("lin2.weight").data_ptr(),
                                          self.gview("lin2.bias").data_ptr(), d_z.data_ptr(), self._workspace(n).data_ptr(), n, self.in_dim,
                                          self.hid, self.out_dim, p, state.ptr if p > 0 else None, L.stream_ptr(dev)), "ufnd_gnn_backward")
