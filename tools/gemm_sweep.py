#!/usr/bin/env python3
"""Sweep the tile configurations of the bf16 GEMM (diagnostics library: every tile, ablations) over the encoder GEMM shapes (GPU box).
Interleaved rounds in ONE process (variants x rounds), median of per-launch HIP-event times minus
the empty event-pair time, random operands.  Every variant is first checked against an fp32 torch
matmul of the same bf16 data.   usage: gemm_sweep.py [rounds] [--ablate] [--real] [--cold] [--mscale=4] [--cfgs=8,22] [--shapes=bert_qkv,...]"""
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
sys.path.insert(0, str(Path(__file__).resolve().parent))
import torch

from ultrafnd_git_amd import _lib as L
from _diaglib import check as dcheck, diag

DEV = "cuda"
SHAPES = [("bert_qkv", 4096, 2304, 768), ("bert_out", 4096, 768, 768), ("bert_ffn1", 4096, 3072, 768),
          ("bert_ffn2", 4096, 768, 3072), ("vit_qkv", 1600, 2304, 768), ("vit_out", 1600, 768, 768),
          ("vit_ffn1", 1600, 3072, 768), ("vit_ffn2", 1600, 768, 3072), ("vit_patch", 1568, 768, 3072)]
# id -> (BM, BN, "WMxWN/ring slots A,W"): mirrors UFND_GEMM_TILES in csrc/gemm_bf16.hip
TILES = {0: (128, 128, "2x2/3,3"), 1: (128, 64, "2x2/3,3"), 2: (256, 128, "4x2/3,3"), 3: (128, 128, "2x2/2,2"), 4: (128, 64, "2x2/4,4"),
         5: (256, 64, "4x2/3,3"), 6: (128, 288, "2x2/3,3"), 7: (128, 96, "2x2/3,3"), 8: (256, 192, "4x2/2,2"), 9: (128, 384, "2x2/2,2"),
         10: (128, 192, "2x2/3,3"), 11: (128, 96, "2x2/4,4"), 12: (64, 96, "1x2/4,4"), 13: (64, 192, "1x2/3,3"), 14: (128, 256, "2x2/2,2"),
         15: (256, 256, "4x2/2,2"), 16: (128, 128, "4x2/3,3"), 17: (128, 192, "4x2/3,3"), 18: (256, 64, "4x2/4,4"), 19: (64, 64, "1x2/4,4"),
         20: (128, 64, "4x2/4,4"), 21: (256, 192, "2x4/2,2"), 22: (256, 192, "4x2/3,2"), 23: (256, 256, "4x2/3,2"),
         24: (128, 192, "4x2/4,4"), 25: (128, 256, "4x2/3,3"),
         # 32x32x16 MFMA forms
         26: (256, 192, "4x2/2,2/m32"), 27: (128, 128, "4x2/3,3/m32"), 28: (256, 144, "8x1/2,2"), 29: (128, 128, "4x2/2,2")}


REAL = {}   # name -> (act, use_residual_and_f32_out): the epilogue each shape has in the encoders


BASE = None   # ctypes handle of an older build of the library (--base=path): timed in the same rounds


def _base_lib(path):
    import ctypes as C
    lib = C.CDLL(path)
    P, I = C.c_void_p, C.c_int
    lib.ufnd_gemm_bf16_ex.argtypes = [P] * 6 + [I] * 10 + [P]
    lib.ufnd_gemm_bf16_ex.restype = I
    return lib


def run_ln(A, W, bias, ob, M, N, K, name):
    """the LayerNorm-aware entry (automatic tile) with the epilogue the folded encoders use for this shape"""
    import ctypes
    act, res = REAL.get(name, (0, False))
    ln = L.GemmLn()
    ln.a_eps = ln.r_eps = 1e-5
    ln.width = 768
    if res:      # residual through LayerNorm + bf16 copy + row statistics out
        ln.r_stats, ln.r_gamma, ln.r_beta, ln.r_parts, ln.out_stats = ST.data_ptr(), GB.data_ptr(), GB.data_ptr(), 24, STO.data_ptr()
        L.check(L.lib().ufnd_gemm_bf16_ln(A.data_ptr(), W.data_ptr(), bias.data_ptr(), RES.data_ptr(), ob.data_ptr(), OF.data_ptr(), M, N, K,
                                          K, K, N, N, N, act, ctypes.byref(ln), L.stream_ptr(A.device)), "gemm_ln")
    else:        # LayerNorm of the A operand folded in
        ln.a_stats, ln.colsum, ln.a_parts = ST.data_ptr(), GB.data_ptr(), 24
        L.check(L.lib().ufnd_gemm_bf16_ln(A.data_ptr(), W.data_ptr(), bias.data_ptr(), None, ob.data_ptr(), None, M, N, K,
                                          K, K, 0, N, 0, act, ctypes.byref(ln), L.stream_ptr(A.device)), "gemm_ln")


def run(cfg, A, W, bias, ob, M, N, K, name=None):
    act, res = REAL.get(name, (0, False))
    if cfg == 5000:
        return run_ln(A, W, bias, ob, M, N, K, name)
    if cfg >= 10000:     # the baseline library's tile `cfg - 10000` (its +1000 = software-pipelined schedule)
        r, o, f = (RES.data_ptr(), None, OF.data_ptr()) if res else (None, ob.data_ptr(), None)
        rc = BASE.ufnd_gemm_bf16_ex(A.data_ptr(), W.data_ptr(), bias.data_ptr(), r, o, f, M, N, K, K, K, N if res else 0, 0 if res else N,
                                    N if res else 0, act, cfg - 10000, L.stream_ptr(A.device))
        assert rc == 0, rc
        return
    if res:
        dcheck(diag().ufnd_diag_gemm_bf16_ex(A.data_ptr(), W.data_ptr(), bias.data_ptr(), RES.data_ptr(), None, OF.data_ptr(), M, N, K,
                                             K, K, N, 0, N, act, cfg, L.stream_ptr(A.device)), "gemm_ex")
    else:
        dcheck(diag().ufnd_diag_gemm_bf16_ex(A.data_ptr(), W.data_ptr(), bias.data_ptr(), None, ob.data_ptr(), None, M, N, K, K, K, 0,
                                             N, 0, act, cfg, L.stream_ptr(A.device)), "gemm_ex")


def empty_pair_ms():
    ev = []
    for _ in range(64):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); e1.record()
        ev.append((e0, e1))
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in ev)
    return t[len(t) // 2]


def main():
    global RES, OF, BASE, ST, GB, STO
    ST = torch.rand(4096, 24, 2, device=DEV) + 1.0
    GB = torch.rand(3072, device=DEV)
    STO = torch.empty(4096, 24, 2, device=DEV)
    base_cfgs = []
    for a in sys.argv:
        if a.startswith("--base="):
            BASE = _base_lib(a.split("=", 1)[1])
        if a.startswith("--base-cfgs="):
            base_cfgs = [int(x) for x in a.split("=")[1].split(",")]
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 5
    if "--real" in sys.argv:
        REAL.update({"bert_out": (0, True), "bert_ffn1": (1, False), "bert_ffn2": (0, True), "vit_out": (0, True),
                     "vit_ffn1": (2, False), "vit_ffn2": (0, True)})
    RES = torch.randn(4096, 3072, device=DEV)
    OF = torch.empty(4096, 3072, device=DEV)
    ablate = "--ablate" in sys.argv
    out = {}
    ONLY = None
    for a in sys.argv:
        if a.startswith("--cfgs="):
            ONLY = {int(x) for x in a.split("=")[1].split(",")}
    shapes = SHAPES
    for a in sys.argv:
        if a.startswith("--shapes="):
            want = a.split("=")[1].split(",")
            shapes = [s for s in SHAPES if s[0] in want]
    mscale = max([int(a.split("=")[1]) for a in sys.argv if a.startswith("--mscale=")] + [1])   # G x the rows (encoder lookahead G)
    if mscale > 1:
        ST, STO = ST.repeat(mscale, 1, 1), STO.repeat(mscale, 1, 1)
        RES, OF = torch.randn(4096 * mscale, 3072, device=DEV), torch.empty(4096 * mscale, 3072, device=DEV)
    for name, M, N, K in shapes:
        M *= mscale
        g = torch.Generator().manual_seed(M + N)
        A = torch.randn(M, K, generator=g).to(DEV).bfloat16()
        W = (torch.randn(N, K, generator=g) / K ** 0.5).to(DEV).bfloat16()
        bias = torch.randn(N, generator=g).to(DEV)
        ob = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
        ref = A.float() @ W.float().t() + bias
        cfgs = [c for c, (bm, bn, _) in TILES.items() if N % bn == 0 and (ONLY is None or c in ONLY)]
        for c in cfgs:
            ob.zero_()
            run(c, A, W, bias, ob, M, N, K)
            torch.cuda.synchronize()
            err = (ob.float() - ref).abs().max().item()
            assert err <= 0.02 * max(1.0, ref.abs().max().item()), (name, c, err)
        variants = list(cfgs)
        if BASE is not None:
            variants += [10000 + c for c in base_cfgs if N % TILES[c % 1000][1] == 0]
        if "--ln" in sys.argv:
            variants += [5000]
        if ablate:     # (timing-only builds exist for the tiles in use)
            variants += [100 + c for c in cfgs if c in (8, 16, 17, 20, 28)] + [200 + c for c in cfgs if c in (8, 16, 17, 20, 28)]
        times = {c: [] for c in variants}
        empty = empty_pair_ms()
        cold = "--cold" in sys.argv
        junk = torch.empty(96 << 20, dtype=torch.uint8, device=DEV) if cold else None
        # --rotate=R: R copies of the operands, a different one per launch (R x (A + W) beyond the 256 MiB Infinity Cache
        # means every launch streams its operands from HBM, like a layer's weights inside the encoder pass)
        rot = max([int(a.split("=")[1]) for a in sys.argv if a.startswith("--rotate=")] + [1])
        As = [A] + [A.clone() for _ in range(rot - 1)] if "--rotate-w-only" not in sys.argv else [A] * rot
        Ws = [W] + [W.clone() for _ in range(rot - 1)] if "--rotate-a-only" not in sys.argv else [W] * rot
        ctr = 0
        for _ in range(rounds):
            for c in variants:
                evs = []
                torch.cuda._sleep(4_000_000)          # let the host run ahead of the GPU
                for _ in range(8):
                    if cold:
                        junk.add_(1)                   # 192 MB of traffic: L2 and part of the MALL turn over
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    ctr += 1
                    e0.record(); run(c, As[ctr % rot], Ws[ctr % rot], bias, ob, M, N, K, name); e1.record()
                    evs.append((e0, e1))
                torch.cuda.synchronize()
                times[c] += [a.elapsed_time(b) - empty for a, b in evs]
        row = {}
        for c in variants:
            t = sorted(times[c])
            med = t[len(t) // 2] * 1e-3
            if c == 5000:
                bm, bn, tag = 0, 1, "LN-aware entry (auto tile)"
            elif c >= 10000:
                bm, bn, lay = TILES[c % 1000]
                tag = f"BASE {c % 1000}:{bm}x{bn}" + ("-sched" if (c - 10000) >= 1000 else "")
            else:
                bm, bn, lay = TILES[c % 100]
                tag = f"{c % 100}:{bm}x{bn}/{lay}" + {0: "", 1: "-noMFMA", 2: "-noDMA"}[(c % 1000) // 100]
            tiles = -(-M // bm) * (N // bn) if bm else 0
            row[tag] = {"us": round(med * 1e6, 2), "tflops": round(2.0 * M * N * K / med / 1e12, 1), "tiles": tiles}
        out[name] = row
        print(f"== {name} M={M} N={N} K={K} (empty event pair {empty * 1e3:.1f} us)")
        for k, v in sorted(row.items(), key=lambda kv: kv[1]["us"]):
            print(f"   {k:28s} {v['us']:7.1f} us {v['tflops']:7.1f} TF  tiles {v['tiles']}")
    Path("gpurun_out").mkdir(exist_ok=True)
    Path("gpurun_out/gemm_sweep.json").write_text(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
