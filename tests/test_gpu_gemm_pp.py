"""GPU: the persistent, software-pipelined bf16 GEMM (csrc/gemm_bf16_pp.hpp, tile id 64) against the one-tile-per-workgroup
kernels.  The two are held BIT-IDENTICAL: a row's results must not depend on which kernel -- or which batch -- computed it
(the encoders mix both within one pass), and a race in the persistent kernel's hand-counted vmcnt schedule would show up as a
mismatch on some launch, so every comparison is repeated on fresh output buffers."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
PP = 64


def _lib():
    from ultrafnd_git_amd import _lib as L
    return L


def _split_stats(x, parts):
    M, H = x.shape
    xs = x.view(M, parts, H // parts)
    return torch.stack([xs.sum(2), (xs * xs).sum(2)], 2).contiguous()


class _Case:
    def __init__(self, M, N, mode, act, seed):
        L = _lib()
        g = torch.Generator().manual_seed(seed)
        K = 768
        self.M, self.N, self.K, self.mode, self.act = M, N, K, mode, act
        self.A = torch.randn(M, K, generator=g).to(DEV).bfloat16()
        self.W = (torch.randn(N, K, generator=g) / K ** 0.5).to(DEV).bfloat16()
        self.bias = torch.randn(N, generator=g).to(DEV)
        self.colsum = self.W.float().sum(1).contiguous()
        self.gamma, self.beta = (1 + 0.2 * torch.randn(N, generator=g)).to(DEV), (0.1 * torch.randn(N, generator=g)).to(DEV)
        self.res = torch.randn(M, N, generator=g).to(DEV).bfloat16()
        self.stats = _split_stats((torch.randn(M, 768, generator=g) * 1.3 + 0.2).to(DEV), 24)
        self.guard = torch.zeros(L.FOLD_GUARD_SLOTS, device=DEV)

    def run(self, tile):
        L = _lib()
        M, N, K = self.M, self.N, self.K
        out = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
        ost = torch.full((M, N // 32, 2), float("nan"), device=DEV)
        st = L.stream_ptr(out.device)
        if self.mode == "plain":
            rc = L.lib().ufnd_gemm_bf16_ex(self.A.data_ptr(), self.W.data_ptr(), self.bias.data_ptr(), None, out.data_ptr(), None, M, N, K, K, K, 0, N, 0,
                                           self.act, tile, st)
        else:
            ln = L.GemmLn()
            ln.a_eps = ln.r_eps = 1e-5
            ln.width = 768
            ln.tile_cfg = tile
            if self.mode == "fold":
                ln.a_stats, ln.colsum, ln.a_parts, ln.guard = self.stats.data_ptr(), self.colsum.data_ptr(), 24, self.guard.data_ptr()
            else:
                ln.residual_bf16, ln.ldrb, ln.out_stats = self.res.data_ptr(), N, ost.data_ptr()
                if self.mode == "rln":
                    ln.r_stats, ln.r_gamma, ln.r_beta, ln.r_parts = self.stats.data_ptr(), self.gamma.data_ptr(), self.beta.data_ptr(), 24
            rc = L.lib().ufnd_gemm_bf16_ln(self.A.data_ptr(), self.W.data_ptr(), self.bias.data_ptr(), None, out.data_ptr(), None, M, N, K, K, K, 0, N, 0,
                                           self.act, C.byref(ln), st)
        return rc, out, ost


# (M, N, mode, activation, the one-tile kernel to compare with)
CASES = [
    (4096, 3072, "fold", 1, 15),      # FFN1, erf-GELU: the shape family the automatic choice sends here
    (6400, 3072, "fold", 2, 22),      # ViT FFN1, quick-GELU: 25 row panels (a ragged last group of the walk order)
    (2048, 2304, "fold", 0, 22),      # folded Q/K/V
    (2048, 768, "rln", 0, 22),        # residual through a LayerNorm, bf16 stream, row statistics out
    (6400, 768, "res", 0, 17),        # plain bf16 residual (ViT), row statistics out
    (512, 2304, "plain", 0, 22),      # plain call; 48 tiles: fewer workgroups than CUs
    (256, 3072, "plain", 1, 15),      # one row panel, activation without a fold
]


@pytest.mark.parametrize("M,N,mode,act,old", CASES)
def test_persistent_gemm_is_bit_identical_to_the_one_tile_kernels(M, N, mode, act, old):
    case = _Case(M, N, mode, act, seed=M + N + act)
    rc, ref, ref_st = case.run(old)
    assert rc == 0, _lib().lib().ufnd_last_error()
    torch.cuda.synchronize()
    assert torch.isfinite(ref.float()).all()
    for rep in range(6):      # fresh NaN-filled outputs every time: a rare race shows as a mismatch or an unwritten element
        rc, out, ost = case.run(PP)
        assert rc == 0, _lib().lib().ufnd_last_error()
        torch.cuda.synchronize()
        assert torch.equal(out.view(torch.int16), ref.view(torch.int16)), (mode, rep, int((out.view(torch.int16) != ref.view(torch.int16)).sum()))
        if mode in ("rln", "res"):
            assert torch.equal(ost.view(torch.int32), ref_st.view(torch.int32)), (mode, rep)
    if mode == "fold":      # the fold guard of the persistent kernel reports the same ratio as the rows imply
        st = case.stats.double()
        mean = st[..., 0].sum(1) / 768
        var = (st[..., 1].sum(1) / 768 - mean * mean).clamp_min(0)
        want = float((mean.abs() / (var + 1e-5).sqrt()).max())
        assert abs(float(case.guard.max()) - want) <= 1e-3 * max(want, 1.0)


def test_automatic_choice_and_refusals():
    """The automatic tile choice sends folded-activation calls with at least two tiles per workgroup to the persistent form
    (same bits either way); a forced call it cannot serve is refused with a message, never mis-run."""
    L = _lib()
    case = _Case(8192, 3072, "fold", 1, seed=5)      # 32 x 24 = 768 tiles >= 512: automatic = persistent
    rc, a, _ = case.run(-1)
    assert rc == 0
    rc, b, _ = case.run(15)
    assert rc == 0
    torch.cuda.synchronize()
    assert torch.equal(a.view(torch.int16), b.view(torch.int16))
    ragged = _Case(1000, 3072, "fold", 1, seed=6)    # M % 256 != 0: automatic falls back, forced is refused
    rc, _, _ = ragged.run(-1)
    assert rc == 0
    rc, _, _ = ragged.run(PP)
    assert rc != 0 and b"persistent" in L.lib().ufnd_last_error()
    torch.cuda.synchronize()
