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
