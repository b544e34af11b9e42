#!/usr/bin/env python3
"""The persistent, software-pipelined bf16 GEMM (csrc/gemm_bf16_pp.hpp, tile id 64) against the one-tile-per-workgroup kernels
on the encoder shapes with K = 768 (GPU box; product library only).

1. bit-exactness: every epilogue mode (plain, folded LayerNorm + none / GELU / quick-GELU, residual through a LayerNorm,
   plain bf16 residual) must give the SAME bits as the old kernel on random operands (outputs and row statistics);
2. timing: interleaved rounds in ONE process (guide rule 24), HIP events around `reps` back-to-back launches per variant,
   operands rotated through `--rotate` copies (cold L2, as inside an encoder pass), median and min per variant.

usage: gemm_pp_bench.py [--rounds=7] [--reps=20] [--rotate=6] [--shapes=ffn1,qkv,out,...] [--no-check]"""
import ctypes
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch

from ultrafnd_git_amd import _lib as L

DEV = "cuda"
PP = 64
# name: (M, N, K, mode, act, old tile)      mode: plain / fold / rln / res
SHAPES = {
    "ffn1": (16384, 3072, 768, "fold", 1, 15), "qkv": (16384, 2304, 768, "fold", 0, 22), "out": (16384, 768, 768, "rln", 0, 22),
    "out_res": (16384, 768, 768, "res", 0, 22), "qkv0": (16384, 2304, 768, "plain", 0, 22),
    "vit_ffn1": (6400, 3072, 768, "fold", 2, 22), "vit_qkv": (6400, 2304, 768, "fold", 0, 15), "vit_out": (6400, 768, 768, "res", 0, 17),
    "l512_ffn1": (65536, 3072, 768, "fold", 1, 15), "b32_ffn1": (4096, 3072, 768, "fold", 1, 22),
    "m32k_qkv": (32768, 2304, 768, "fold", 0, 22), "m32k_out": (32768, 768, 768, "rln", 0, 22), "m32k_plain": (32768, 2304, 768, "plain", 0, 22),
    "m24k_qkv": (24576, 2304, 768, "fold", 0, 22), "l512_res": (65536, 768, 768, "res", 0, 22),
    "l512_qkv": (65536, 2304, 768, "fold", 0, 22), "l512_out": (65536, 768, 768, "rln", 0, 22), "f8_ffn1": (12800, 3072, 768, "fold", 2, 22),
}


class Case:
    def __init__(self, name, rotate):
        self.name = name
        M, N, K, mode, act, old = SHAPES[name]
        self.M, self.N, self.K, self.mode, self.act, self.old = M, N, K, mode, act, old
        g = torch.Generator(device=DEV).manual_seed(hash(name) % 1000)
        rnd = lambda *s: torch.randn(*s, device=DEV, generator=g)
        self.A = [(rnd(M, K) * 1.0).bfloat16() for _ in range(rotate)]
        self.W = [(rnd(N, K) * 0.05).bfloat16() for _ in range(rotate)]
        self.bias = rnd(N) * 0.1
        self.colsum = self.W[0].float().sum(1).contiguous()
        self.gamma, self.beta = 1 + 0.1 * rnd(N), 0.1 * rnd(N)
        self.res = [(rnd(M, N)).bfloat16() for _ in range(rotate)]
        # partial statistics of plausible rows (24 parts of 32 columns over 768)
        x = rnd(M, 768) * 1.3 + 0.2
        xs = x.view(M, 24, 32)
        self.stats = torch.stack([xs.sum(2), (xs * xs).sum(2)], 2).contiguous()
        self.out = [torch.empty(M, N, dtype=torch.bfloat16, device=DEV) for _ in range(2)]
        self.ostats = [torch.zeros(M, N // 32, 2, device=DEV) for _ in range(2)]
        self.guard = torch.zeros(L.FOLD_GUARD_SLOTS, device=DEV)

    def launch(self, tile, slot=0, rot=0):
        M, N, K = self.M, self.N, self.K
        A, W = self.A[rot % len(self.A)], self.W[rot % len(self.W)]
        out = self.out[slot]
        st = L.stream_ptr(A.device)
        if self.mode == "plain":
            L.check(L.lib().ufnd_gemm_bf16_ex(A.data_ptr(), W.data_ptr(), self.bias.data_ptr(), None, out.data_ptr(), None, M, N, K, K, K, 0, N, 0,
                                              self.act, tile, st), "gemm_ex")
            return
        ln = L.GemmLn()
        ln.a_eps = ln.r_eps = 1e-5
        ln.width = 768
        ln.tile_cfg = tile
        if self.mode == "fold":
            ln.a_stats, ln.colsum, ln.a_parts = self.stats.data_ptr(), self.colsum.data_ptr(), 24
            ln.guard = self.guard.data_ptr()
        else:
            r = self.res[rot % len(self.res)]
            ln.residual_bf16, ln.ldrb = r.data_ptr(), N
            ln.out_stats = self.ostats[slot].data_ptr()
            if self.mode == "rln":
                ln.r_stats, ln.r_gamma, ln.r_beta, ln.r_parts = self.stats.data_ptr(), self.gamma.data_ptr(), self.beta.data_ptr(), 24
        L.check(L.lib().ufnd_gemm_bf16_ln(A.data_ptr(), W.data_ptr(), self.bias.data_ptr(), None, out.data_ptr(), None, M, N, K, K, K, 0, N, 0,
                                          self.act, ctypes.byref(ln), st), "gemm_ln")

    def flops(self):
        return 2.0 * self.M * self.N * self.K


def check(case):
    case.out[0].zero_(); case.out[1].fill_(1.0)
    case.ostats[0].zero_(); case.ostats[1].fill_(1.0)
    case.launch(case.old, 0)
    case.launch(PP, 1)
    torch.cuda.synchronize()
    same = torch.equal(case.out[0].view(torch.int16), case.out[1].view(torch.int16))
    nbad = int((case.out[0].view(torch.int16) != case.out[1].view(torch.int16)).sum())
    res = {"out_bits_equal": same, "out_mismatches": nbad}
    if case.mode in ("rln", "res"):
        res["stats_bits_equal"] = torch.equal(case.ostats[0].view(torch.int32), case.ostats[1].view(torch.int32))
    # and against fp32 torch on the same bf16 operands (plain/res only: sanity of the reference kernel itself)
    if case.mode in ("plain", "res") and case.act == 0:
        ref = case.A[0].float() @ case.W[0].float().t() + case.bias
        if case.mode == "res":
            ref = ref + case.res[0].float()
        err = float((case.out[1].float() - ref).abs().max() / ref.abs().max())
        res["rel_err_vs_fp32"] = err
    return res


def time_variants(case, variants, rounds, reps):
    times = {v: [] for v in variants}
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    rot = 0
    for v in variants:      # warm-up
        for _ in range(3):
            case.launch(v, 0, rot); rot += 1
    torch.cuda.synchronize()
    for _ in range(rounds):
        for v in variants:
            ev[0].record()
            for _ in range(reps):
                case.launch(v, 0, rot); rot += 1
            ev[1].record()
            torch.cuda.synchronize()
            times[v].append(ev[0].elapsed_time(ev[1]) * 1e3 / reps)
    return times


def main():
    args = {a.split("=")[0]: (a.split("=")[1] if "=" in a else "1") for a in sys.argv[1:] if a.startswith("--")}
    rounds, reps, rotate = int(args.get("--rounds", 7)), int(args.get("--reps", 20)), int(args.get("--rotate", 6))
    names = args.get("--shapes", "ffn1,qkv,out,out_res,qkv0,vit_ffn1,vit_qkv,vit_out").split(",")
    report = {}
    for name in names:
        case = Case(name, rotate if SHAPES[name][0] * SHAPES[name][1] < 3e8 else 2)
        entry = {"shape": list(SHAPES[name][:3]), "mode": case.mode, "act": case.act}
        if "--no-check" not in args:
            entry["check"] = check(case)
            for _ in range(3):      # again, with every buffer warm / different interleavings
                c2 = check(case)
                if not c2["out_bits_equal"]:
                    entry["check"] = c2
        variants = [case.old, PP]
        if name in ("ffn1", "qkv", "vit_ffn1", "vit_qkv", "l512_ffn1") and case.old != 22:
            variants.append(22)
        t = time_variants(case, variants, rounds, reps)
        for v in variants:
            xs = sorted(t[v])
            med, mn = xs[len(xs) // 2], xs[0]
            entry[f"tile{v}"] = {"median_us": round(med, 2), "min_us": round(mn, 2), "tflops_median": round(case.flops() / med / 1e6, 1)}
        report[name] = entry
        print(name, json.dumps(entry), flush=True)
        del case
        torch.cuda.empty_cache()
    out = Path("gpurun_out")
    out.mkdir(exist_ok=True)
    (out / "gemm_pp_bench.json").write_text(json.dumps(report, indent=1))


if __name__ == "__main__":
    main()
