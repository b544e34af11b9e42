"""Graph side of the trainer's construction, MI355X-native.

Mirrors src/training/forensic_trainer.py of the reference:
  build_adj_from_ocr(ocr_sets, thresh)      :121-132   -> one HIP launch instead of an O(N^2) Python loop
  SimpleGCN(in_dim, hid, out_dim, dropout)  :25-53     same constructor / forward(x, adj) / state_dict keys
  node_features(cache)                      :193-195   the (N, 416) node-feature matrix
  pretrain_gnn(gnn, X, Adj, gnn_dim, epochs):214-224   the degree-regression Adam steps
  build_gnn_embeddings(cache, cfg)          :184-211   everything `_build_gnn` does, returns (gnn, Z)
There is no CPU path: tensors must live on a HIP device.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn as nn

from . import _lib as L
from .state import StepStateBuffer


def sets_to_csr(ocr_sets: Sequence[set]) -> Tuple[np.ndarray, np.ndarray]:
    """Phrase sets (any hashable phrases) -> (offsets (N+1,) int32, tokens int32), every set's ids sorted and
    duplicate-free.  The phrase -> id map is local to the call (Jaccard only tests equality)."""
    vocab: Dict[object, int] = {}
    offs, toks = [0], []
    for s in ocr_sets:
        ids = sorted({vocab.setdefault(ph, len(vocab)) for ph in s})
        toks.extend(ids)
        offs.append(len(toks))
    return np.asarray(offs, dtype=np.int32), np.asarray(toks, dtype=np.int32)


def build_adj_from_ocr(ocr_sets: Sequence[set], thresh: float = 0.12, device="cuda") -> torch.Tensor:
    """(N, N) fp32 0/1 adjacency with unit diagonal on `device` (forensic_trainer.py:121-132)."""
    dev = torch.device(device)
    if dev.type != "cuda":
        raise L.UltrafndHipError("build_adj_from_ocr runs on a HIP device only (no CPU fallback)")
    offs, toks = sets_to_csr(ocr_sets)
    n = len(ocr_sets)
    o = torch.from_numpy(offs).to(dev)
    t = torch.from_numpy(toks if toks.size else np.zeros(1, dtype=np.int32)).to(dev)
    adj = torch.empty(n, n, dtype=torch.float32, device=dev)
    L.check(L.lib().ufnd_ocr_adjacency(o.data_ptr(), t.data_ptr(), n, float(thresh), adj.data_ptr(), n, L.stream_ptr(dev)),
            "ufnd_ocr_adjacency")
    return adj


def node_features(cache: Dict) -> np.ndarray:
    """[T[:, :192], A[:, :32], V[:, :128], U[:, :64]] rows, L2-normalised (forensic_trainer.py:193-195)."""
    X = np.concatenate([cache["text"][:, :192], cache["audio"][:, :32], cache["visual"][:, :128], cache["temporal"][:, :64]],
                       axis=1).astype(np.float32)
    X /= (np.linalg.norm(X, axis=1, keepdims=True) + 1e-9)
    return X


class SimpleGCN(nn.Module):
    """Two-layer GCN over post nodes (forensic_trainer.py:25-53): same constructor, `forward(x, adj)`, state_dict
    keys (lin1.weight, lin1.bias, lin2.weight, lin2.bias) and, under the same torch seed, the same initial weights
    (the two nn.Linear inits are drawn in the reference's order).  The four tensors are views of one flat buffer."""

    def __init__(self, in_dim: int, hid: int = 128, out_dim: int = 128, dropout: float = 0.3):
        super().__init__()
        self.in_dim, self.hid, self.out_dim, self.dropout = in_dim, hid, out_dim, float(dropout)
        l1, l2 = nn.Linear(in_dim, hid), nn.Linear(hid, out_dim)          # RNG consumption identical to the reference
        flat = torch.cat([l1.weight.detach().flatten(), l1.bias.detach(), l2.weight.detach().flatten(), l2.bias.detach()])
        self.flat = nn.Parameter(flat, requires_grad=False)
        self._seed = (torch.initial_seed() ^ 0x6763_6E5F_6472_6F70) & 0x7FFF_FFFF_FFFF_FFFF   # no draw from the global RNG stream
        self._calls = 0

    def _views(self):
        n1, n2 = self.hid * self.in_dim, self.out_dim * self.hid
        f = self.flat.data
        return (f[:n1].view(self.hid, self.in_dim), f[n1:n1 + self.hid], f[n1 + self.hid:n1 + self.hid + n2].view(self.out_dim, self.hid),
                f[n1 + self.hid + n2:])

    def state_dict(self, *a, **kw):
        w1, b1, w2, b2 = self._views()
        return {"lin1.weight": w1.clone(), "lin1.bias": b1.clone(), "lin2.weight": w2.clone(), "lin2.bias": b2.clone()}

    def load_state_dict(self, sd, strict: bool = True):
        for dst, k in zip(self._views(), ("lin1.weight", "lin1.bias", "lin2.weight", "lin2.bias")):
            dst.copy_(sd[k].to(dst.device, torch.float32))

    def _params(self) -> L.GcnParams:
        p = L.GcnParams()
        p.w1, p.b1, p.w2, p.b2 = (t.data_ptr() for t in self._views())
        return p

    def _state(self, dev) -> StepStateBuffer:
        """dropout key of this call: (module seed, call counter) -- a fresh mask per forward, as nn.Dropout draws"""
        self._calls += 1
        st = StepStateBuffer(dev, seed=self._seed)
        st.set_u64("step", self._calls)
        return st

    @torch.no_grad()
    def forward(self, x: torch.Tensor, adj: torch.Tensor) -> torch.Tensor:
        dev = L.require_hip(x, adj, self.flat)
        x, adj = L.f32c(x), L.f32c(adj)
        n = x.shape[0]
        if x.shape[1] != self.in_dim or tuple(adj.shape) != (n, n):
            raise RuntimeError(f"SimpleGCN: x {tuple(x.shape)} / adj {tuple(adj.shape)} do not match in_dim {self.in_dim}")
        ws = torch.empty(L.lib().ufnd_gcn_workspace_floats(n, self.in_dim, self.hid, self.out_dim, 0), dtype=torch.float32, device=dev)
        z = torch.empty(n, self.out_dim, dtype=torch.float32, device=dev)
        p = self.dropout if self.training else 0.0
        st = self._state(dev) if p > 0 else None
        L.check(L.lib().ufnd_gcn_forward(x.data_ptr(), adj.data_ptr(), n, C.byref(self._params()), z.data_ptr(), ws.data_ptr(), n,
                                         self.in_dim, self.hid, self.out_dim, p, st.ptr if st is not None else None,
                                         L.stream_ptr(dev)), "ufnd_gcn_forward")
        return z


@torch.no_grad()
def pretrain_gnn(gnn: SimpleGCN, X: torch.Tensor, Adj: torch.Tensor, gnn_dim: int, epochs: int = 2, lr: float = 1e-3,
                 weight_decay: float = 1e-4, head: Optional[nn.Linear] = None) -> list:
    """ForensicTrainer._pretrain_gnn (forensic_trainer.py:214-224): `epochs` full-graph Adam steps of
    mse(sigmoid(head(gnn(X, Adj))), rowsum(Adj) / max(1, N)); only the GCN's parameters are updated (the
    reference's optimizer holds nothing else).  Returns the loss of every step."""
    dev = L.require_hip(X, Adj, gnn.flat)
    X, Adj = L.f32c(X), L.f32c(Adj)
    n = X.shape[0]
    if head is None:
        head = nn.Linear(gnn_dim, 1)                     # drawn after the GCN's own init, as in the reference
    hw, hb = head.weight.detach().to(dev, torch.float32).contiguous(), head.bias.detach().to(dev, torch.float32).contiguous()
    m, v = torch.zeros_like(gnn.flat.data), torch.zeros_like(gnn.flat.data)
    ws = torch.empty(L.lib().ufnd_gcn_workspace_floats(n, gnn.in_dim, gnn.hid, gnn.out_dim, 1), dtype=torch.float32, device=dev)
    z = torch.empty(n, gnn.out_dim, dtype=torch.float32, device=dev)
    loss = torch.zeros(1, dtype=torch.float32, device=dev)
    losses = []
    gnn.train()
    for e in range(epochs):
        p = gnn.dropout
        st = gnn._state(dev) if p > 0 else None
        L.check(L.lib().ufnd_gcn_pretrain_step(X.data_ptr(), Adj.data_ptr(), n, C.byref(gnn._params()), m.data_ptr(), v.data_ptr(),
                                               hw.data_ptr(), hb.data_ptr(), z.data_ptr(), ws.data_ptr(), n, gnn.in_dim, gnn.hid,
                                               gnn.out_dim, p, lr, weight_decay, e + 1, st.ptr if st is not None else None,
                                               loss.data_ptr(), L.stream_ptr(dev)), "ufnd_gcn_pretrain_step")
        losses.append(float(loss.item()))
    return losses


def build_gnn_embeddings(cache: Dict, gnn_dim: int = 128, overlap_thresh: float = 0.12, device="cuda",
                         pretrain_epochs: int = 2) -> Tuple[SimpleGCN, torch.Tensor, torch.Tensor, torch.Tensor]:
    """ForensicTrainer._build_gnn (forensic_trainer.py:184-211): node features, OCR-Jaccard adjacency,
    SimpleGCN(416, 2*gnn_dim, gnn_dim, dropout 0.2), two pre-training steps, then the cached node embeddings
    Z = gnn(X, Adj) -- computed, as in the reference, with the module still in train mode (its dropout is
    active in that forward).  Returns (gnn, X, Adj, Z)."""
    dev = torch.device(device)
    X = torch.from_numpy(node_features(cache)).to(dev)
    Adj = build_adj_from_ocr(cache["ocr_sets"], overlap_thresh, dev)
    gnn = SimpleGCN(in_dim=X.shape[1], hid=2 * gnn_dim, out_dim=gnn_dim, dropout=0.2).to(dev)
    pretrain_gnn(gnn, X, Adj, gnn_dim, epochs=pretrain_epochs)
    Z = gnn(X, Adj)
    return gnn, X, Adj, Z
