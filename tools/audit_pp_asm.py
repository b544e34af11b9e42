#!/usr/bin/env python3
"""Static audit of the persistent GEMM's ISA (csrc/gemm_bf16_pp.hip), run at build / test time on the CPU.

The kernel issues its epilogue loads by inline asm (hipcc must not see them: it would drain the LDS-DMA queue in front of
their first use) and retires them with hand-counted `s_waitcnt vmcnt(N)`.  The compiler does not know the destinations are in
flight: any instruction it places between a load and the wait that retires it and that READS OR WRITES a destination register
(a live-range copy, a spill) uses garbage -- on some launches only.  That happened once (two asm wait statements in the arms of
a branch: the register copies in front of the branch read the statistics before they had landed); this audit makes it a build
failure.  Every retiring wait carries a `; PPRETIRE <registers>` comment naming what it retires.

Also required: no scratch (a spill's reload waits vmcnt(0) and a spill's store changes the hand-counted queue), no packed fp32
instruction, and no compiler-inserted vmcnt wait inside the K loops.

usage: audit_pp_asm.py [file.s]     (without an argument: compiles csrc/gemm_bf16_pp.hip with -save-temps into a temp dir)"""
import re
import subprocess
import sys
import tempfile
from pathlib import Path

REPO = Path(__file__).resolve().parents[1]


def regs(tok):
    out = []
    for m in re.finditer(r"v\[(\d+):(\d+)\]|\bv(\d+)\b", tok):
        if m.group(1):
            out += list(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.append(int(m.group(3)))
    return out


def compile_to_asm(tmp: Path) -> Path:
    sys.path.insert(0, str(REPO))
    from ultrafnd_git_amd.build import ARCH, FILE_FLAGS
    src = REPO / "ultrafnd_git_amd" / "csrc" / "gemm_bf16_pp.hip"
    cmd = ["/opt/rocm/bin/hipcc", f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-Wno-pass-failed", "-save-temps=obj", "-c", str(src),
           "-o", str(tmp / "pp.o")] + FILE_FLAGS["gemm_bf16_pp"]
    subprocess.run(cmd, cwd=str(src.parent), check=True, capture_output=True)
    return next(tmp.glob("*gfx950.s"))


def audit(asm_path: Path) -> list:
    S = asm_path.read_text().split("\n")
    problems = []
    starts = [i for i, l in enumerate(S) if re.match(r"^_ZN\d+_GLOBAL__N_1\d+gemm_pp_kernel.*:", l)]
    assert starts, "no gemm_pp_kernel in the listing"
    for st in starts:
        name = S[st].split(":")[0]
        en = st
        while not S[en].startswith(".Lfunc_end"):
            en += 1
        L = S[st:en]
        pend = {}
        inasm = False
        compiler_waits = []
        for i, l in enumerate(L):
            if "ASMSTART" in l:
                inasm = True
                continue
            if "ASMEND" in l:
                inasm = False
                continue
            if "PPRETIRE" in l:
                for r in regs(l.split("PPRETIRE")[1]):
                    pend.pop(r, None)
                continue
            c = l.split(";")[0].strip()
            if not c or c.startswith("."):
                continue
            parts = c.split(None, 1)
            op, rest = parts[0], (parts[1] if len(parts) > 1 else "")
            if "scratch_" in op:
                problems.append((name, i, "scratch access: " + c))
            if op.startswith("v_pk_") and "f32" in op:
                problems.append((name, i, "packed fp32 instruction: " + c))
            if inasm and (op.startswith("global_load") or op.startswith("buffer_load")) and " lds" not in rest:
                for r in regs(rest.split(",")[0]):
                    pend[r] = i
                continue
            if not inasm and op == "s_waitcnt" and "vmcnt" in rest:
                # allowed: exactly one, the wait of the __syncthreads() in front of the un-overlapped last epilogue (directly followed
                # by its s_barrier; the K loops' barriers are raw s_barrier builtins behind asm waits)
                k = i + 1
                while k < len(L) and (not L[k].split(";")[0].strip() or L[k].split(";")[0].strip().startswith(".")):
                    k += 1
                compiler_waits.append((i, c, L[k].strip() if k < len(L) else ""))
            if op.startswith("s_"):
                continue
            hit = set(regs(rest)) & set(pend)
            if hit:
                problems.append((name, i, f"touches in-flight asm-load destination(s) v{sorted(hit)}: " + c))
        if len(compiler_waits) != 1 or not compiler_waits[0][2].startswith("s_barrier"):
            problems.append((name, -1, f"compiler-inserted vmcnt waits: {compiler_waits} (exactly one expected, the last epilogue's __syncthreads)"))
    return problems


def main():
    if len(sys.argv) > 1:
        probs = audit(Path(sys.argv[1]))
    else:
        with tempfile.TemporaryDirectory() as t:
            probs = audit(compile_to_asm(Path(t)))
    for p in probs[:40]:
        print(p)
    print(f"{len(probs)} problem(s)")
    return 1 if probs else 0


if __name__ == "__main__":
    sys.exit(main())
