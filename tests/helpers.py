"""Shared test helpers: fixture loading and digest comparison (tests only)."""
import json
from pathlib import Path

import numpy as np
import torch

GOLDEN = Path(__file__).resolve().parent / "golden"


def load_npz(name):
    return np.load(GOLDEN / name, allow_pickle=False)


def digest(t: torch.Tensor):
    f = t.detach().double().flatten().cpu()
    n = f.numel()
    stride = max(1, n // 16)
    return {"norm": f.norm().item(), "sum": f.sum().item(),
            "head": f[:16].float().numpy(), "strided": f[::stride][:16].float().numpy()}


def assert_digest_close(z, prefix, t, rtol, atol, what=""):
    """Compare tensor t with the digest stored under `prefix` in npz z."""
    d = digest(t)
    ref_norm = float(z[f"{prefix}/norm"])
    scale = max(ref_norm / max(1.0, np.sqrt(t.numel())), 1e-12)   # rms of the reference tensor
    assert abs(d["norm"] - ref_norm) <= rtol * ref_norm + atol, (what, prefix, "norm", d["norm"], ref_norm)
    for k in ("head", "strided"):
        ref = z[f"{prefix}/{k}"]
        err = np.abs(d[k] - ref).max()
        assert err <= rtol * max(np.abs(ref).max(), scale) + atol, (what, prefix, k, err)


def nograd_keys(z):
    return json.loads(str(z["nograd_keys"]))


def feature_errors(got, ref):
    """Three views of the error of unit-norm feature rows (max-abs alone lets a 3x regression pass):
    max-abs, the worst row's relative L2 error ||got - ref|| / ||ref||, and the worst row's 1 - cosine."""
    g = torch.as_tensor(np.asarray(got) if not torch.is_tensor(got) else got).double().cpu()
    r = torch.as_tensor(np.asarray(ref) if not torch.is_tensor(ref) else ref).double().cpu()
    g, r = g.reshape(-1, g.shape[-1]), r.reshape(-1, r.shape[-1])
    d = g - r
    cos = (g * r).sum(-1) / (g.norm(dim=-1) * r.norm(dim=-1)).clamp_min(1e-30)
    return {"max_abs": d.abs().max().item(), "rel_l2": (d.norm(dim=-1) / r.norm(dim=-1).clamp_min(1e-30)).max().item(),
            "one_minus_cos": (1.0 - cos).max().item()}


def assert_features_close(got, ref, max_abs, rel_l2, one_minus_cos, what=""):
    e = feature_errors(got, ref)
    print(f"{what}: max-abs {e['max_abs']:.3e} (<= {max_abs:.1e}), rel-L2 {e['rel_l2']:.3e} (<= {rel_l2:.1e}), "
          f"1-cos {e['one_minus_cos']:.3e} (<= {one_minus_cos:.1e})")
    assert e["max_abs"] <= max_abs and e["rel_l2"] <= rel_l2 and e["one_minus_cos"] <= one_minus_cos, (what, e)
    return e


def hidden_errors(got, ref):
    """Hidden states (entries of order 1): relative RMS error and max-abs."""
    g, r = got.double().cpu(), ref.double().cpu()
    d = g - r
    return {"rel_rms": (d.pow(2).mean().sqrt() / r.pow(2).mean().sqrt()).item(), "max_abs": d.abs().max().item()}
