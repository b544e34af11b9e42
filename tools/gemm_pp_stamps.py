#!/usr/bin/env python3
"""In-kernel stamps of the persistent bf16 GEMM (diagnostics library): per workgroup {entry, ring prologue landed, last K loop
done, end} and its tile count.  Reports: K-loop time per tile, the share of a workgroup's life spent with its K loops running,
the un-overlapped last epilogue, and the clock held.  usage: gemm_pp_stamps.py [--shapes=ffn1,...] [--noepi]"""
import ctypes
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
sys.path.insert(0, str(Path(__file__).resolve().parent))
import torch

from ultrafnd_git_amd import _lib as L
from _diaglib import check as dcheck, diag
from gemm_pp_bench import Case, SHAPES


def run(name, dbg):
    case = Case(name, 2)
    M, N, K = case.M, case.N, case.K
    G = 256
    stamps = torch.zeros(9 * G, dtype=torch.int64, device="cuda")
    ln = L.GemmLn()
    ln.a_eps = ln.r_eps = 1e-5
    ln.width = 768
    lnp = None
    if case.mode == "fold":
        ln.a_stats, ln.colsum, ln.a_parts = case.stats.data_ptr(), case.colsum.data_ptr(), 24
        lnp = ctypes.byref(ln)
    elif case.mode in ("rln", "res"):
        ln.residual_bf16, ln.ldrb, ln.out_stats = case.res[0].data_ptr(), N, case.ostats[0].data_ptr()
        if case.mode == "rln":
            ln.r_stats, ln.r_gamma, ln.r_beta, ln.r_parts = case.stats.data_ptr(), case.gamma.data_ptr(), case.beta.data_ptr(), 24
        lnp = ctypes.byref(ln)
    res = None
    for it in range(6):      # warm launches; the last one is read
        dcheck(diag().ufnd_diag_gemm_pp_stamps(case.A[it % 2].data_ptr(), case.W[it % 2].data_ptr(), case.out[0].data_ptr(), M, N, K, stamps.data_ptr(),
                                               lnp, case.bias.data_ptr(), case.act, dbg, L.stream_ptr(stamps.device)), "pp_stamps")
    torch.cuda.synchronize()
    s = stamps.cpu().numpy().astype("float64")
    st, nt = s[:8 * G].reshape(G, 8), s[8 * G:]
    live = nt > 0
    st, nt = st[live], nt[live]
    rt = st[:, 1::2] / 100.0      # s_memrealtime: 100 MHz -> us
    cyc = st[:, 0::2]
    import numpy as np
    t0 = rt[:, 0].min()
    pro, loop, epi, life = rt[:, 1] - rt[:, 0], rt[:, 2] - rt[:, 1], rt[:, 3] - rt[:, 2], rt[:, 3] - rt[:, 0]
    ghz = (cyc[:, 2] - cyc[:, 1]) / np.maximum(loop, 1e-9) / 1000.0
    res = {"workgroups": int(live.sum()), "tiles_per_wg": [int(nt.min()), int(nt.max())], "prologue_us": round(float(np.median(pro)), 2),
           "kloops_us": round(float(np.median(loop)), 2), "per_tile_us": round(float(np.median(loop / nt)), 2),
           "last_epilogue_us": round(float(np.median(epi)), 2), "kloop_share_of_life": round(float(np.median(loop / life)), 3),
           "clock_ghz": round(float(np.median(ghz)), 2), "grid_span_us": round(float(rt[:, 3].max() - t0), 2),
           "start_skew_us": round(float(rt[:, 0].max() - t0), 2)}
    return res


def main():
    args = {a.split("=")[0]: (a.split("=")[1] if "=" in a else "1") for a in sys.argv[1:] if a.startswith("--")}
    names = args.get("--shapes", "ffn1,qkv,out,vit_ffn1").split(",")
    for name in names:
        for dbg, tag in ((1, "full"), (5, "no-stores"), (9, "no-patch-writes"), (17, "no-row-phase"), (33, "no-register-phase"), (3, "no-epilogue-slices")):
            print(name, SHAPES[name][:3], tag, json.dumps(run(name, dbg)), flush=True)


if __name__ == "__main__":
    main()
