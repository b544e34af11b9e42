"""TEST INFRASTRUCTURE (see oracle/__init__.py) -- the integrated trainer variant's in-graph GNN.

torch-fp32 restatement of
  src/training/forensic_trainer_integrated.py:77-98   build_adj_from_ocr_sets (weighted Jaccard adjacency of a mini-batch)
  src/models/gnn/gnn_model.py:7-41                    GNNModel: Z = lin2(A_norm @ drop(relu(A_norm @ lin1(X))))
Pinned against the real functions / class by tests/golden/make_golden.py gnn_model -> tests/golden/gnn_model.npz.
The reference's `_pack_batch` (:203-206) cannot execute (`torch.stack([T, A, V, U])` with widths 768/128/512/256); the node
feature used by the build is the 416-wide compact concat that line's comment and `gnn_in_dim = 416` describe -- the main
trainer's forensic_trainer.py:193-195 (oracle.gcn_ref.node_features).
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, List, Sequence

import numpy as np
import torch
import torch.nn.functional as F


def build_adj_from_ocr_sets(ocr_sets: Sequence[set], overlap_thresh: float = 0.12) -> np.ndarray:
    n = len(ocr_sets)
    a = np.zeros((n, n), dtype=np.float32)
    for i in range(n):
        si = ocr_sets[i]
        if not si:
            continue
        for j in range(i + 1, n):
            sj = ocr_sets[j]
            if not sj:
                continue
            inter, union = len(si & sj), len(si | sj)
            s = (inter / union) if union > 0 else 0.0
            if s >= overlap_thresh:
                a[i, j] = a[j, i] = s
    return a


def shapes(in_dim: int = 416, hid: int = 256, out_dim: int = 128):
    return OrderedDict([("lin1.weight", (hid, in_dim)), ("lin1.bias", (hid,)), ("lin2.weight", (out_dim, hid)), ("lin2.bias", (out_dim,))])


def seeded_weights(seed: int, in_dim: int = 416, hid: int = 256, out_dim: int = 128) -> "OrderedDict[str, torch.Tensor]":
    g = torch.Generator().manual_seed(seed)
    out = OrderedDict()
    for k, shp in shapes(in_dim, hid, out_dim).items():
        out[k] = torch.randn(shp, generator=g) * ((1.0 / math.sqrt(shp[1])) if len(shp) == 2 else 0.05)
    return out


def norm_adj(a: torch.Tensor) -> torch.Tensor:
    a_hat = a + torch.eye(a.shape[0], dtype=a.dtype)
    deg = a_hat.sum(dim=-1).clamp_min(1e-9)
    d = torch.diag(torch.pow(deg, -0.5))
    return d @ a_hat @ d


def forward(w: Dict[str, torch.Tensor], x: torch.Tensor, a: torch.Tensor) -> torch.Tensor:
    """eval-mode GNNModel.forward (dropout off); differentiable in w."""
    an = norm_adj(a)
    h = F.relu(an @ F.linear(x, w["lin1.weight"], w["lin1.bias"]))
    return F.linear(an @ h, w["lin2.weight"], w["lin2.bias"])


def synthetic_ocr_sets(n: int, seed: int, vocab: int = 40, empty_every: int = 7) -> List[set]:
    rng = np.random.RandomState(seed)
    out = []
    for i in range(n):
        if empty_every and i % empty_every == empty_every - 1:
            out.append(set())
            continue
        k = int(rng.randint(1, 9))
        out.append(set(int(t) for t in rng.randint(0, vocab, size=k)))
    return out
