#!/usr/bin/env python3
"""Reference point, not product: the vendor library (torch.matmul -> hipBLASLt / rocBLAS) on the encoder GEMM shapes,
timed like tools/gemm_sweep.py (median of per-launch HIP-event times minus the empty pair, warm operands), beside
ufnd_gemm_bf16 with the automatic tile.  No epilogue on either side (bias-free, bf16 out)."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch

from tools.gemm_sweep import SHAPES, empty_pair_ms
from ultrafnd_git_amd import _lib as L

DEV = "cuda"


def timeit(fn, rounds=6):
    empty = empty_pair_ms()
    ts = []
    for _ in range(rounds):
        torch.cuda._sleep(4_000_000)
        evs = []
        for _ in range(8):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record()
            evs.append((e0, e1))
        torch.cuda.synchronize()
        ts += [a.elapsed_time(b) - empty for a, b in evs]
    ts.sort()
    return ts[len(ts) // 2] * 1e-3


for name, M, N, K in SHAPES:
    g = torch.Generator().manual_seed(M + N)
    A = torch.randn(M, K, generator=g).to(DEV).bfloat16()
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(DEV).bfloat16()
    out = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    Wt = W.t()
    t_lib = timeit(lambda: torch.matmul(A, Wt, out=out))
    t_own = timeit(lambda: L.check(L.lib().ufnd_gemm_bf16(A.data_ptr(), W.data_ptr(), None, None, out.data_ptr(), None, M, N, K, K, K, 0, N, 0, 0,
                                                          L.stream_ptr(A.device)), "g"))
    fl = 2.0 * M * N * K
    print(f"{name:10s} {M}x{N}x{K}: vendor library {t_lib * 1e6:6.1f} us {fl / t_lib / 1e12:6.1f} TF | ufnd_gemm_bf16 {t_own * 1e6:6.1f} us {fl / t_own / 1e12:6.1f} TF")
