"""TEST INFRASTRUCTURE (see oracle/__init__.py) -- TemporalSyncNet.align oracle.

torch-fp32 restatement of src/core_blocks/temporal_blocks.py:102-140 (+ `_cosine` :10-13):
  v is zero-padded / truncated to the text width D; feat = [t, v, t-v, t*v, cos(t,v)] (4D+1);
  out = Linear(2*out, out)(GELU(Linear(4D+1, 2*out)(feat)))   (dropout inactive: inference_mode).
The reference never trains these weights (random init, fixed projection).  Pinned against the real
class by tests/golden/make_golden.py -> tests/golden/temporal.npz.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict

import torch
import torch.nn.functional as F


def shapes(in_dim: int = 768, out_dim: int = 256):
    return OrderedDict([("proj.0.weight", (2 * out_dim, 4 * in_dim + 1)), ("proj.0.bias", (2 * out_dim,)),
                        ("proj.3.weight", (out_dim, 2 * out_dim)), ("proj.3.bias", (out_dim,))])


def seeded_weights(seed: int, in_dim: int = 768, out_dim: int = 256) -> "OrderedDict[str, torch.Tensor]":
    g = torch.Generator().manual_seed(seed)
    out = OrderedDict()
    for k, shp in shapes(in_dim, out_dim).items():
        out[k] = torch.randn(shp, generator=g) * ((1.0 / math.sqrt(shp[1])) if len(shp) == 2 else 0.05)
    return out


def align(w: Dict[str, torch.Tensor], t: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
    """(B,D), (B,Dv) -> (B,out_dim)."""
    t, v = t.float(), v.float()
    D = t.shape[-1]
    if v.shape[-1] < D:
        v = torch.cat([v, v.new_zeros(v.shape[0], D - v.shape[-1])], dim=-1)
    else:
        v = v[..., :D]
    an = t / (t.norm(dim=-1, keepdim=True) + 1e-9)
    bn = v / (v.norm(dim=-1, keepdim=True) + 1e-9)
    cos = (an * bn).sum(dim=-1, keepdim=True)
    feat = torch.cat([t, v, t - v, t * v, cos], dim=-1)
    h = F.gelu(F.linear(feat, w["proj.0.weight"], w["proj.0.bias"]))
    return F.linear(h, w["proj.3.weight"], w["proj.3.bias"])
