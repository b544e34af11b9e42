#!/usr/bin/env python3
"""In-kernel phase times of the bf16 GEMM (GPU box): ufnd_diag_gemm_bf16_stamps (diagnostics library) writes s_memtime /
s_memrealtime at kernel entry, first K-step landed, K loop done, stores drained.  Prints, per tile
configuration and shape, the median over blocks of each phase in us (100 MHz realtime counter), the
shader clock held inside the K loop, and the first-start -> last-end span of the whole grid.
With --ln every tile that has a LayerNorm-aware kernel is also stamped in its two roles: "fold" (LayerNorm of
the A operand folded in) and "rln" (residual through a LayerNorm + bf16 copy + row statistics out).
--mscale=G: G times the rows (the launches of an encoder pass over G batches).
usage: gemm_stamps.py [--cfgs=8,22] [--shapes=bert_qkv,bert_ffn1] [--ln] [--mscale=4]"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
sys.path.insert(0, str(Path(__file__).resolve().parent))
import torch

from ultrafnd_git_amd import _lib as L
from _diaglib import check as dcheck, diag
from gemm_sweep import SHAPES, TILES


def main():
    only, want = None, None
    for a in sys.argv:
        if a.startswith("--cfgs="):
            only = {int(x) for x in a.split("=")[1].split(",")}
        if a.startswith("--shapes="):
            want = a.split("=")[1].split(",")
    dev = "cuda"
    mscale = max([int(a.split("=")[1]) for a in sys.argv if a.startswith("--mscale=")] + [1])
    for name, M, N, K in SHAPES:
        if want and name not in want:
            continue
        M *= mscale
        g = torch.Generator().manual_seed(M + N)
        A = torch.randn(M, K, generator=g).to(dev).bfloat16()
        W = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev).bfloat16()
        ob = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        print(f"== {name} M={M} N={N} K={K}: median over blocks, us")
        import ctypes
        stats = torch.rand(M, 24, 2, device=dev) + 1.0
        vec = torch.rand(max(N, 3072), device=dev)
        res = torch.randn(M, N, device=dev)
        resb = res.bfloat16()
        gslots = torch.zeros(L.FOLD_GUARD_SLOTS, device=dev)
        of = torch.empty(M, N, device=dev)
        sto = torch.empty(M, 24, 2, device=dev)
        variants = []
        for c, (bm, bn, lay) in TILES.items():
            if N % bn or (only is not None and c not in only) or c not in (8, 15, 16, 17, 20, 22, 28):      # stamp builds: tiles in use
                continue
            variants.append((c, bm, bn, lay, None))
            if "--ln" in sys.argv and c in (8, 15, 16, 17, 20, 22):
                variants.append((c, bm, bn, lay + " fold", "fold"))
                variants.append((c, bm, bn, lay + " fold+gelu", "fold+gelu"))
                variants.append((c, bm, bn, lay + " fold+gelu+guard", "fold+gelu+guard"))      # round 3: the fold guard inside the GEMM (ufnd_gemm_ln.guard)
                if N // 32 <= 24 and (bn // 2) % 32 == 0:
                    variants.append((c, bm, bn, lay + " rln", "rln"))
                    variants.append((c, bm, bn, lay + " rlnb", "rlnb"))      # round 3: bf16 residual stream (bf16 in, bf16 + statistics out)
                    variants.append((c, bm, bn, lay + " resb", "resb"))      # ... without the LayerNorm on the residual (ViT)
                    if "--ablate-rln" in sys.argv:     # the residual epilogue piece by piece
                        for m in ("res0", "res", "res+stats", "rln-nostats"):
                            variants.append((c, bm, bn, lay + " " + m, m))
        for c, bm, bn, lay, mode in variants:
            nblk = -(-M // bm) * (N // bn)
            st = torch.zeros(nblk, 8, dtype=torch.int64, device=dev)
            rows = []
            ln = None
            if mode is not None:
                ln = L.GemmLn()
                ln.a_eps = ln.r_eps = 1e-5
                ln.width = 768
                if mode in ("fold", "fold+gelu", "fold+gelu+guard"):
                    ln.a_stats, ln.colsum, ln.a_parts = stats.data_ptr(), vec.data_ptr(), 24
                    ln.tile_cfg = 0 if mode == "fold" else 1      # (the diagnostics entry reads the activation from this field)
                    if mode == "fold+gelu+guard":
                        ln.guard = gslots.data_ptr()
                elif mode in ("rlnb", "resb"):
                    if mode == "rlnb":
                        ln.r_stats, ln.r_gamma, ln.r_beta, ln.r_parts = stats.data_ptr(), vec.data_ptr(), vec.data_ptr(), 24
                    ln.out_stats, ln.residual_bf16, ln.ldrb = sto.data_ptr(), resb.data_ptr(), N
                elif mode == "rln":
                    ln.r_stats, ln.r_gamma, ln.r_beta, ln.r_parts, ln.out_stats = stats.data_ptr(), vec.data_ptr(), vec.data_ptr(), 24, sto.data_ptr()
                elif mode == "rln-nostats":
                    ln.r_stats, ln.r_gamma, ln.r_beta, ln.r_parts = stats.data_ptr(), vec.data_ptr(), vec.data_ptr(), 24
                elif mode == "res+stats":      # plain fp32 residual, fp32 + bf16 out, statistics out
                    ln.out_stats = sto.data_ptr()
                elif mode == "res0":           # the plain kernel: fp32 residual, fp32 + bf16 out
                    ln = None
            for it in range(12):
                dcheck(diag().ufnd_diag_gemm_bf16_stamps(A.data_ptr(), W.data_ptr(), ob.data_ptr(), M, N, K, c, st.data_ptr(),
                                                      ctypes.byref(ln) if ln is not None else None, vec.data_ptr() if mode is not None else None,
                                                      res.data_ptr() if mode not in (None, "fold", "fold+gelu", "fold+gelu+guard", "rlnb", "resb") else None, of.data_ptr() if mode not in (None, "fold", "fold+gelu", "fold+gelu+guard", "rlnb", "resb") else None,
                                                      L.stream_ptr(A.device)), "stamps")
                torch.cuda.synchronize()
                if it >= 4:
                    rows.append(st.clone())
            s = torch.stack(rows).double()              # (iters, blocks, 8)
            rt = s[..., 1::2] * 0.01                    # realtime stamps in us
            cy = s[..., 0::2]
            pro = (rt[..., 1] - rt[..., 0]).median().item()
            loop = (rt[..., 2] - rt[..., 1]).median().item()
            epi = (rt[..., 3] - rt[..., 2]).median().item()
            ghz = ((cy[..., 2] - cy[..., 1]) / (rt[..., 2] - rt[..., 1]).clamp_min(1e-9) * 1e-3).median().item()
            span = (rt[..., 3].amax(1) - rt[..., 0].amin(1)).median().item()
            skew = (rt[..., 0].amax(1) - rt[..., 0].amin(1)).median().item()
            nk = K // 64
            print(f"   {c}:{bm}x{bn}/{lay:10s} blocks {nblk:4d}  prologue {pro:5.2f}  loop {loop:6.2f} ({loop / nk:5.3f}/K-step, {ghz:4.2f} GHz)"
                  f"  epilogue {epi:5.2f}  grid span {span:6.2f}  start skew {skew:5.2f}")


if __name__ == "__main__":
    main()
