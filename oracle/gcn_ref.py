"""TEST INFRASTRUCTURE (see oracle/__init__.py) -- graph side of the trainer's construction.

CPU restatement of src/training/forensic_trainer.py:
  jaccard / build_adj_from_ocr   :114-132   OCR phrase-set Jaccard >= thresh -> (N,N) 0/1 adjacency, diagonal 1
  node_features                  :193-195   [T[:, :192], A[:, :32], V[:, :128], U[:, :64]] -> L2 rows (N,416)
  SimpleGCN.forward              :25-53     A_hat = A + I; A_norm = D^-1/2 A_hat D^-1/2 (deg + 1e-9);
                                            z = lin2(A_norm @ drop(gelu(lin1(A_norm @ x))))
  pretrain                       :214-224   `epochs` full-graph Adam steps (lr 1e-3, L2 weight decay 1e-4) of
                                            mse(sigmoid(head(Z)), rowsum(Adj) / max(1, N)); only the GCN's
                                            parameters are in that optimizer (the head stays at its init)
Phrase sets are sets of strings in the reference; here they are sets of non-negative ints (the index of the
phrase in any fixed vocabulary) -- Jaccard only compares for equality.

Pinned by tests/golden/make_golden.py (part `gcn`): adjacency and eval-mode forward against the reference's own
functions / class; the pre-training steps against torch.optim.Adam + autograd through the reference's SimpleGCN
with its dropout p set to 0 (train-mode dropout draws from the global RNG: checked statistically on the device).
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F


# ---------------------------------------------------------------- adjacency (forensic_trainer.py:114-132)
def jaccard(a: set, b: set) -> float:
    if not a and not b:
        return 0.0
    inter = len(a.intersection(b))
    union = len(a.union(b)) + 1e-9
    return float(inter / union)


def build_adj_from_ocr(ocr_sets: Sequence[set], thresh: float = 0.12) -> np.ndarray:
    n = len(ocr_sets)
    A = np.zeros((n, n), dtype=np.float32)
    for i in range(n):
        A[i, i] = 1.0
        for j in range(i + 1, n):
            if jaccard(ocr_sets[i], ocr_sets[j]) >= thresh:
                A[i, j] = A[j, i] = 1.0
    return A


def sets_to_csr(ocr_sets: Sequence[set]) -> Tuple[np.ndarray, np.ndarray]:
    """sorted unique int32 ids per set, CSR: (offsets (N+1,), tokens (nnz,)) -- the device kernel's input layout."""
    offs, toks = [0], []
    for s in ocr_sets:
        ids = sorted(int(x) for x in s)
        toks.extend(ids)
        offs.append(len(toks))
    return np.asarray(offs, dtype=np.int32), np.asarray(toks, dtype=np.int32)


def synthetic_ocr_sets(n: int, seed: int, vocab: int = 400, groups: int = 12) -> List[set]:
    """Phrase sets with structure: every node draws most phrases from its group's pool (so that some pairs clear
    the 0.12 threshold) plus noise; a few nodes have empty sets (the reference's both-empty -> 0 branch)."""
    g = np.random.RandomState(seed)
    pools = [g.choice(vocab, size=24, replace=False) for _ in range(groups)]
    out = []
    for i in range(n):
        if g.rand() < 0.06:
            out.append(set())
            continue
        pool = pools[g.randint(groups)]
        k = g.randint(1, 14)
        s = set(int(x) for x in g.choice(pool, size=min(k, len(pool)), replace=False))
        s |= set(int(x) for x in g.randint(0, vocab, size=g.randint(0, 5)))
        out.append(s)
    return out


# ---------------------------------------------------------------- node features (forensic_trainer.py:193-195)
def node_features(T: np.ndarray, A: np.ndarray, V: np.ndarray, U: np.ndarray) -> np.ndarray:
    X = np.concatenate([T[:, :192], A[:, :32], V[:, :128], U[:, :64]], axis=1).astype(np.float32)
    X /= (np.linalg.norm(X, axis=1, keepdims=True) + 1e-9)
    return X


# ---------------------------------------------------------------- SimpleGCN (forensic_trainer.py:25-53)
def shapes(in_dim: int = 416, hid: int = 256, out_dim: int = 128):
    return OrderedDict([("lin1.weight", (hid, in_dim)), ("lin1.bias", (hid,)), ("lin2.weight", (out_dim, hid)), ("lin2.bias", (out_dim,))])


def seeded_weights(seed: int, in_dim: int = 416, hid: int = 256, out_dim: int = 128) -> "OrderedDict[str, torch.Tensor]":
    g = torch.Generator().manual_seed(seed)
    out = OrderedDict()
    for k, shp in shapes(in_dim, hid, out_dim).items():
        out[k] = torch.randn(shp, generator=g) * ((1.0 / math.sqrt(shp[1])) if len(shp) == 2 else 0.05)
    return out


def norm_adj(adj: torch.Tensor) -> torch.Tensor:
    n = adj.shape[0]
    a_hat = adj + torch.eye(n, dtype=adj.dtype)
    deg = a_hat.sum(dim=-1) + 1e-9
    d = torch.pow(deg, -0.5)
    return d[:, None] * a_hat * d[None, :]


def gcn_forward(w: Dict[str, torch.Tensor], x: torch.Tensor, adj: torch.Tensor, drop_mul: Optional[torch.Tensor] = None) -> torch.Tensor:
    """drop_mul: the dropout multiplier applied to gelu(lin1(.)) ((N,hid): 0 or 1/(1-p)); None = eval mode."""
    an = norm_adj(adj.float())
    h = F.gelu(F.linear(an @ x.float(), w["lin1.weight"], w["lin1.bias"]))
    if drop_mul is not None:
        h = h * drop_mul
    return F.linear(an @ h, w["lin2.weight"], w["lin2.bias"])


def pretrain(w: Dict[str, torch.Tensor], x: torch.Tensor, adj: torch.Tensor, head_w: torch.Tensor, head_b: torch.Tensor,
             epochs: int = 2, lr: float = 1e-3, weight_decay: float = 1e-4,
             drop_muls: Optional[Sequence[torch.Tensor]] = None) -> Tuple["OrderedDict[str, torch.Tensor]", List[float]]:
    """forensic_trainer.py:214-224 with the dropout multipliers given explicitly (None = p 0).
    Returns (updated weights, loss of every epoch)."""
    p = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in w.items())
    opt = torch.optim.Adam(list(p.values()), lr=lr, weight_decay=weight_decay)
    target = adj.float().sum(dim=-1, keepdim=True) / max(1.0, adj.shape[0])
    losses = []
    for e in range(epochs):
        z = gcn_forward(p, x, adj, None if drop_muls is None else drop_muls[e])
        pred = torch.sigmoid(F.linear(z, head_w, head_b))
        loss = F.mse_loss(pred, target)
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(float(loss))
    return OrderedDict((k, v.detach()) for k, v in p.items()), losses
