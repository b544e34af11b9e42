"""Build the HIP libraries in-tree with hipcc for gfx950 (MI355X).

    python -m ultrafnd_git_amd.build [--verbose] [--force] [--diag] [--defs=A,B=1]     (--defs: extra -D switches, experiments)

  libultrafnd_hip.so        the product: every source of csrc/*.hip (include/ultrafnd_hip.h)
  libultrafnd_hip_diag.so   diagnostics only (csrc/diag/*.hip, built with -DUFND_DIAG): timing ablations, in-kernel
                            stamps, every experimental GEMM tile, the placement probe.  tools/ load it; the package never does.

hipcc cross-compiles without a GPU.  The .so files are git-ignored but travel to the GPU box with the working
tree.  Sources are compiled to objects in parallel, then linked.  Nothing here imports torch.
"""
from __future__ import annotations

import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
LIB = PKG / "libultrafnd_hip.so"
DIAG_LIB = PKG / "libultrafnd_hip_diag.so"
OBJ = PKG / "build"
ARCH = "gfx950"


def sources():
    return sorted(CSRC.glob("*.hip"))


def diag_sources():
    return sorted((CSRC / "diag").glob("*.hip")) + [CSRC / "api.hip"]


EXTRA_DEFS: list = []      # experiments: extra -D switches (python -m ultrafnd_git_amd.build --defs=A,B=1); part of the build digest


def _defs() -> list:
    return ["-D" + d for d in EXTRA_DEFS if d]


def _digest(extra: str) -> str:
    h = hashlib.sha256()
    h.update(extra.encode())
    h.update(" ".join(_defs()).encode())       # an experimental -D build is never mistaken for the current one
    h.update(repr(sorted(FILE_FLAGS.items())).encode())
    files = sorted(list(CSRC.glob("*")) + list((CSRC / "diag").glob("*")) + [PKG.parent / "include" / "ultrafnd_hip.h"])
    for f in files:
        if f.is_file():
            h.update(f.name.encode())
            h.update(f.read_bytes())
    return h.hexdigest()


# per-file switches.  The persistent GEMM is built without packed fp32 instructions (csrc/gemm_bf16_pp.hip says why); the feature
# switch reaches the host pass of hipcc too, which answers "not a recognized feature for this target (ignoring feature)".
FILE_FLAGS = {"gemm_bf16_pp": ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"],
              "gemm_pp_diag": ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]}


def _compile(src: Path, tag: str, flags: list, verbose: bool) -> Path:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    obj = OBJ / f"{tag}_{src.stem}.o"
    cmd = [hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-Wno-pass-failed", "-c", str(src), "-o", str(obj)] + flags + _defs()
    cmd += FILE_FLAGS.get(src.stem, [])
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
        print(" ".join(cmd))
    r = subprocess.run(cmd, cwd=str(src.parent), capture_output=not verbose, text=True)
    if r.returncode != 0:
        sys.stderr.write((r.stdout or "") + (r.stderr or ""))
        raise RuntimeError(f"hipcc failed on {src.name}")
    return obj


def _link(objs: list, lib: Path) -> None:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    r = subprocess.run([hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", str(lib)] + [str(o) for o in objs],
                       capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write((r.stdout or "") + (r.stderr or ""))
        raise RuntimeError(f"link of {lib.name} failed")


def _build_one(lib: Path, srcs: list, tag: str, flags: list, force: bool, verbose: bool) -> Path:
    stamp = PKG / f".{lib.stem}.stamp"
    dig = _digest(tag)
    if not force and lib.exists() and stamp.exists() and stamp.read_text().strip() == dig:
        return lib
    OBJ.mkdir(exist_ok=True)
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(lambda s: _compile(s, tag, flags, verbose), srcs))
    _link(objs, lib)
    stamp.write_text(dig)
    return lib


def build(force: bool = False, verbose: bool = False) -> Path:
    return _build_one(LIB, sources(), "prod", [], force, verbose)


def build_diag(force: bool = False, verbose: bool = False) -> Path:
    # api.hip is compiled again with renamed exports: the diagnostics library is self-contained
    flags = ["-DUFND_DIAG=1", "-Dufnd_last_error=ufnd_diag_last_error", "-Dufnd_abi_version=ufnd_diag_abi_version"]
    return _build_one(DIAG_LIB, diag_sources(), "diag", flags, force, verbose)


if __name__ == "__main__":
    for a in sys.argv[1:]:
        if a.startswith("--defs="):
            EXTRA_DEFS[:] = a[len("--defs="):].split(",")
    p = build(force="--force" in sys.argv, verbose="--verbose" in sys.argv)
    print(p)
    if "--diag" in sys.argv:
        print(build_diag(force="--force" in sys.argv, verbose="--verbose" in sys.argv))
