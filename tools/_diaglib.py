"""Loader of libultrafnd_hip_diag.so (csrc/diag/, built with -DUFND_DIAG): timing ablations, in-kernel stamps, every
experimental GEMM tile, the placement probe.  Tools only -- the product package never loads it."""
import ctypes as C
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: F401,E402  (its HIP runtime must be resident first)

from ultrafnd_git_amd import _lib as L  # noqa: E402
from ultrafnd_git_amd.build import DIAG_LIB, build_diag  # noqa: E402

_d = None


def diag() -> C.CDLL:
    global _d
    if _d is None:
        if not DIAG_LIB.exists():
            build_diag()
        d = C.CDLL(str(DIAG_LIB))
        P, I = C.c_void_p, C.c_int
        d.ufnd_diag_last_error.restype = C.c_char_p
        d.ufnd_diag_gemm_bf16_ex.argtypes = [P] * 6 + [I] * 10 + [P]
        d.ufnd_diag_gemm_bf16_ex.restype = I
        d.ufnd_diag_gemm_bf16_stamps.argtypes = [P, P, P, I, I, I, I, P, C.POINTER(L.GemmLn), P, P, P, P]
        d.ufnd_diag_gemm_bf16_stamps.restype = I
        d.ufnd_diag_qkv_attention_stamps.argtypes = [P, P, P, P, P, I, I, C.POINTER(L.GemmLn), P, P]
        d.ufnd_diag_qkv_attention_stamps.restype = I
        d.ufnd_diag_gemm_pp_stamps.argtypes = [P, P, P, I, I, I, P, C.POINTER(L.GemmLn), P, I, I, P]
        d.ufnd_diag_gemm_pp_stamps.restype = I
        d.ufnd_diag_where.argtypes = [P, I, C.c_uint64, P]
        d.ufnd_diag_where.restype = I
        _d = d
    return _d


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise RuntimeError(f"{what} failed (code {rc}): {diag().ufnd_diag_last_error().decode()}")
