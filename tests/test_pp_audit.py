"""CPU: static audit of the persistent GEMM's ISA (tools/audit_pp_asm.py).  hipcc cross-compiles csrc/gemm_bf16_pp.hip for gfx950
and the listing must show: no scratch, no packed fp32, no compiler-inserted vmcnt wait inside the K loops, and no instruction
touching the destination of an inline-asm load before the wait that retires it (a hazard the compiler cannot see; it showed up
once as garbage statistics on a workgroup's last tile, on some launches only)."""
import sys
import tempfile
from pathlib import Path

REPO = Path(__file__).resolve().parents[1]


def test_persistent_gemm_isa_passes_the_static_audit():
    sys.path.insert(0, str(REPO / "tools"))
    import audit_pp_asm as A
    with tempfile.TemporaryDirectory() as t:
        problems = A.audit(A.compile_to_asm(Path(t)))
    assert not problems, problems[:5]


def test_the_audit_sees_a_planted_hazard(tmp_path):
    """The audit itself: a listing in which a v_mov reads an asm load's destination before its retiring wait is flagged."""
    sys.path.insert(0, str(REPO / "tools"))
    import audit_pp_asm as A
    good = """_ZN12_GLOBAL__N_114gemm_pp_kernelILi9ELi9ELi0EEEvNS_8GemmArgsE:
\t;;#ASMSTART
\tglobal_load_dwordx4 v[10:13], v[2:3], off
\t;;#ASMEND
\tv_add_f32_e32 v20, v21, v22
\t;;#ASMSTART
\ts_waitcnt vmcnt(0) ; PPRETIRE v[10:13]
\t;;#ASMEND
\tv_add_f32_e32 v20, v10, v22
\ts_waitcnt vmcnt(0) lgkmcnt(0)
\ts_barrier
.Lfunc_end0:
"""
    p = tmp_path / "good.s"
    p.write_text(good)
    assert A.audit(p) == []
    bad = good.replace("v_add_f32_e32 v20, v21, v22", "v_mov_b32_e32 v30, v11")
    p.write_text(bad)
    probs = A.audit(p)
    assert len(probs) == 1 and "in-flight" in probs[0][2]
    p.write_text(good.replace("v_add_f32_e32 v20, v21, v22", "scratch_store_dword off, v40, off"))
    assert any("scratch" in q[2] for q in A.audit(p))
