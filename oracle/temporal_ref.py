"""TEST INFRASTRUCTURE (see oracle/__init__.py) -- TemporalSyncNet.align oracle.

torch-fp32 restatement of src/core_blocks/temporal_blocks.py:102-140 (+ `_cosine` :10-13):
  v is zero-padded / truncated to the text width D; feat = [t, v, t-v, t*v, cos(t,v)] (4D+1);
  out = Linear(2*out, out)(Dropout(GELU(Linear(4D+1, 2*out)(feat)))).  This restates the EVAL-mode module (dropout
  off), which is what the golden pins (make_golden.py calls .eval() first).  torch.inference_mode on the reference's
  align() does not switch dropout off: in the module's default train mode its Dropout(0.1) is live (the product
  honours self.training; tests/test_gpu_tier_a.py checks the train-mode mask statistically).
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


# ------------------------------------------------------------------------------------------------
# Sequence path: `_TinyTCN` (:16-43) + TemporalSyncNet.forward (:141-157), restated frame-major.
# ------------------------------------------------------------------------------------------------
def seq_shapes(in_dim: int, out_dim: int, hid: int, layers: int, k: int):
    s = shapes(in_dim, out_dim)
    ch = in_dim
    for i in range(layers):
        s[f"tcn.convs.{i}.weight"], s[f"tcn.convs.{i}.bias"] = (hid, ch, k), (hid,)
        ch = hid
    for i in range(layers):
        for n in ("weight", "bias", "running_mean", "running_var"):
            s[f"tcn.norms.{i}.{n}"] = (hid,)
        s[f"tcn.norms.{i}.num_batches_tracked"] = ()
    s["head.weight"], s["head.bias"] = (out_dim, 2 * hid), (out_dim,)
    return s


def seq_seeded_weights(seed: int, in_dim: int, out_dim: int, hid: int, layers: int, k: int) -> "OrderedDict[str, torch.Tensor]":
    """Every tensor non-trivial (BatchNorm scale around 1, running variance positive) so no term drops out."""
    g = torch.Generator().manual_seed(seed)
    out = OrderedDict()
    for name, shp in seq_shapes(in_dim, out_dim, hid, layers, k).items():
        if name.endswith("num_batches_tracked"):
            out[name] = torch.tensor(0, dtype=torch.long)
        elif name.endswith("running_var"):
            out[name] = 0.5 + torch.rand(shp, generator=g)
        elif ".norms." in name and name.endswith("weight"):
            out[name] = 1.0 + 0.2 * torch.randn(shp, generator=g)
        elif len(shp) >= 2:
            fan = 1
            for d in shp[1:]:
                fan *= d
            out[name] = torch.randn(shp, generator=g) / math.sqrt(fan)
        else:
            out[name] = 0.1 * torch.randn(shp, generator=g)
    return out


def sequence_forward(w: Dict[str, torch.Tensor], text_seq: torch.Tensor, vis_seq: torch.Tensor, layers: int, k: int,
                     train: bool, momentum: float = 0.1, eps: float = 1e-5):
    """(B,T,Dt), (B,T,Dv) -> (out (B,out_dim), {running stats after the call}).  Dropout p = 0.
    conv 'same' (:29): output frame t reads input frames t - left + j*d, left = d*(k-1)//2, zeros outside the clip."""
    h = torch.cat([text_seq, vis_seq], dim=-1).float()                     # (B, T, C): frames are rows here
    B, T, _ = h.shape
    stats = {}
    for i in range(layers):
        W, b = w[f"tcn.convs.{i}.weight"], w[f"tcn.convs.{i}.bias"]
        d = 2 ** i
        left = d * (k - 1) // 2
        y = b.expand(B, T, -1).clone()
        for j in range(k):
            s = j * d - left                                               # y[:, t] += h[:, t + s] @ W[:, :, j]^T
            lo, hi = max(0, -s), min(T, T - s)
            if hi > lo:
                y[:, lo:hi] += h[:, lo + s:hi + s] @ W[:, :, j].t()
        flat = y.reshape(B * T, -1)
        if train:
            mean = flat.mean(dim=0)
            var = ((flat - mean) ** 2).mean(dim=0)
            n = B * T
            stats[f"tcn.norms.{i}.running_mean"] = (1 - momentum) * w[f"tcn.norms.{i}.running_mean"] + momentum * mean
            stats[f"tcn.norms.{i}.running_var"] = (1 - momentum) * w[f"tcn.norms.{i}.running_var"] + momentum * var * n / (n - 1)
        else:
            mean, var = w[f"tcn.norms.{i}.running_mean"], w[f"tcn.norms.{i}.running_var"]
        z = F.gelu((y - mean) / torch.sqrt(var + eps) * w[f"tcn.norms.{i}.weight"] + w[f"tcn.norms.{i}.bias"])
        h = h + z if z.shape == h.shape else z
    pooled = torch.cat([h.mean(dim=1), h.max(dim=1).values], dim=-1)
    return F.linear(pooled, w["head.weight"], w["head.bias"]), stats
