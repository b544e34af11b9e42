"""Build libultrafnd_hip.so in-tree with hipcc for gfx950 (MI355X).

    python -m ultrafnd_git_amd.build [--verbose]

hipcc cross-compiles without a GPU.  The .so is git-ignored but travels to the GPU box with
the working tree.  Nothing here imports torch.
"""
from __future__ import annotations

import hashlib
import os
import subprocess
import sys
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
LIB = PKG / "libultrafnd_hip.so"
STAMP = PKG / ".libultrafnd_hip.stamp"
ARCH = "gfx950"


def sources():
    return sorted(CSRC.glob("*.hip"))


def _digest() -> str:
    h = hashlib.sha256()
    for f in sorted(list(CSRC.glob("*")) + [PKG.parent / "include" / "ultrafnd_hip.h"]):
        if f.is_file():
            h.update(f.name.encode())
            h.update(f.read_bytes())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = False) -> Path:
    dig = _digest()
    if not force and LIB.exists() and STAMP.exists() and STAMP.read_text().strip() == dig:
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-Wno-pass-failed", "-o", str(LIB)] + [str(s) for s in sources()]
    if os.environ.get("UFND_BUILD_DEFS"):          # experiments: extra -D switches (e.g. UFND_GEMM_WT=1)
        cmd[1:1] = ["-D" + d for d in os.environ["UFND_BUILD_DEFS"].split(",")]
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
        print(" ".join(cmd))
    r = subprocess.run(cmd, cwd=str(CSRC), capture_output=not verbose, text=True)
    if r.returncode != 0:
        sys.stderr.write((r.stdout or "") + (r.stderr or ""))
        raise RuntimeError("hipcc failed building libultrafnd_hip.so")
    STAMP.write_text(dig)
    return LIB


if __name__ == "__main__":
    p = build(force="--force" in sys.argv, verbose="--verbose" in sys.argv)
    print(p)
