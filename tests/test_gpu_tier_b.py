"""GPU parity (through the C ABI): bf16 encoder kernels vs torch references and vs the Tier-B oracle /
third-party golden outputs (tests/golden/tier_b.npz).

Tolerances.  Kernels are compared with an fp32 torch evaluation of the SAME bf16-rounded operands, so
only accumulation order and the output rounding differ: fp32 outputs 2e-3 * scale, bf16 outputs one
bf16 ulp (2^-8 relative) on top.  Encoder features (unit-norm vectors, rms entry 0.036 / 0.044) are
compared with the all-fp32 oracle / third-party outputs: 4e-3 max-abs (bf16 operand rounding through
2..12 layers); the end-to-end LOGIT error that north_star bounds by 1e-3 is reported by
test_end_to_end_logit_error.
"""
import ctypes as C
import json

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests.helpers import assert_features_close, load_npz

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _lib():
    from ultrafnd_git_amd import _lib as L
    return L


def _gemm(A, W, bias=None, residual=None, act=0, want_bf16=True, want_f32=True):
    L = _lib()
    M, K = A.shape
    N = W.shape[0]
    ob = torch.empty(M, N, dtype=torch.bfloat16, device=DEV) if want_bf16 else None
    of = torch.empty(M, N, dtype=torch.float32, device=DEV) if want_f32 else None
    L.check(L.lib().ufnd_gemm_bf16(A.data_ptr(), W.data_ptr(), L.ptr(bias), L.ptr(residual), L.ptr(ob), L.ptr(of), M, N, K,
                                   A.stride(0), W.stride(0), N if residual is not None else 0, N, N, act,
                                   L.stream_ptr(A.device)), "gemm")
    torch.cuda.synchronize()
    return ob, of


@pytest.mark.parametrize("M,N,K,act,use_bias,use_res", [
    (4096, 2304, 768, 0, True, False),     # BERT QKV (wide tile)
    (4096, 3072, 768, 1, True, False),     # FFN1 + GELU
    (4096, 768, 3072, 0, True, True),      # FFN2 + residual (narrow tile)
    (1600, 3072, 768, 2, True, False),     # ViT fc1 + quick_gelu, ragged M
    (130, 512, 768, 0, False, False),      # projection, tiny ragged M
    (1, 64, 64, 0, True, True),            # minimum sizes
    (257, 128, 192, 0, True, False),
])
def test_gemm_bf16(M, N, K, act, use_bias, use_res):
    g = torch.Generator().manual_seed(M + N + K)
    A = (torch.randn(M, K, generator=g)).to(DEV).bfloat16()
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(DEV).bfloat16()
    bias = torch.randn(N, generator=g).to(DEV) if use_bias else None
    res = torch.randn(M, N, generator=g).to(DEV) if use_res else None
    ob, of = _gemm(A, W, bias, res, act)
    ref = A.float() @ W.float().t()
    if use_bias:
        ref = ref + bias
    if act == 1:
        ref = F.gelu(ref)
    elif act == 2:
        ref = ref * torch.sigmoid(1.702 * ref)
    if use_res:
        ref = ref + res
    scale = ref.abs().max().item()
    e32 = (of - ref).abs().max().item()
    e16 = (ob.float() - ref).abs().max().item()
    print(f"gemm {M}x{N}x{K} act={act}: f32 err {e32:.2e} bf16 err {e16:.2e} scale {scale:.2f}")
    assert e32 <= 2e-3 * max(scale, 1.0), e32
    assert e16 <= (2e-3 + 2 ** -8) * max(scale, 1.0), e16


def _exported_tiles():
    """(id, bm, bn, ln_aware) of every tile built into libultrafnd_hip.so (host-side query, no GPU call)."""
    L = _lib()
    out = []
    for t in range(L.lib().ufnd_gemm_bf16_tile_count()):
        bm, bn, ln = C.c_int(), C.c_int(), C.c_int()
        if L.lib().ufnd_gemm_bf16_tile_info(t, C.byref(bm), C.byref(bn), C.byref(ln)):
            out.append((t, bm.value, bn.value, ln.value))
    return out


def test_gemm_bf16_every_exported_tile():
    """Every tile a caller can name through ufnd_gemm_bf16_ex / ufnd_gemm_ln.tile_cfg: plain epilogue (bias, GELU,
    residual, both outputs) on a ragged M, and -- for the LayerNorm-aware tiles -- the folded form and the
    residual-through-LayerNorm form with row statistics.  Ids outside the library are rejected."""
    L = _lib()
    tiles = _exported_tiles()
    assert len(tiles) >= 4 and {22, 16, 17, 20} <= {t[0] for t in tiles}
    g = torch.Generator().manual_seed(99)
    for t, bm, bn, ln_aware in tiles:
        M, N, K = 2 * bm + 37, 2304, 768              # 2304 = lcm-friendly: divisible by 64, 128, 144, 192, 256 x 9
        if N % bn:
            N = bn * 6
        A = torch.randn(M, K, generator=g).to(DEV).bfloat16()
        W = (torch.randn(N, K, generator=g) / K ** 0.5).to(DEV).bfloat16()
        bias, res = torch.randn(N, generator=g).to(DEV), torch.randn(M, N, generator=g).to(DEV)
        ob = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
        of = torch.empty(M, N, device=DEV)
        L.check(L.lib().ufnd_gemm_bf16_ex(A.data_ptr(), W.data_ptr(), bias.data_ptr(), res.data_ptr(), ob.data_ptr(), of.data_ptr(), M, N, K,
                                          K, K, N, N, N, 1, t, L.stream_ptr(A.device)), f"tile {t}")
        torch.cuda.synchronize()
        ref = F.gelu(A.float() @ W.float().t() + bias) + res
        scale = max(ref.abs().max().item(), 1.0)
        assert (of - ref).abs().max().item() <= 2e-3 * scale and (ob.float() - ref).abs().max().item() <= (2e-3 + 2 ** -8) * scale, t
        if not ln_aware:
            continue
        # folded LayerNorm of the A operand
        x = (torch.randn(M, K, generator=g) * 1.5 + 0.2).to(DEV)
        gm, bt = (1 + 0.2 * torch.randn(K, generator=g)).to(DEV), (0.1 * torch.randn(K, generator=g)).to(DEV)
        Wf = (torch.randn(N, K, generator=g) / K ** 0.5).to(DEV)
        Wp = (Wf * gm[None, :]).bfloat16()
        lnp = L.GemmLn()
        st = _split_stats(x, 24)
        lnp.a_stats, lnp.colsum, lnp.a_parts, lnp.a_eps, lnp.r_eps, lnp.width, lnp.tile_cfg = st.data_ptr(), Wp.float().sum(1).contiguous().data_ptr(), 24, 1e-5, 1e-5, K, t
        cs = Wp.float().sum(1).contiguous()
        lnp.colsum = cs.data_ptr()
        b2 = (bias + Wf @ bt).contiguous()
        L.check(L.lib().ufnd_gemm_bf16_ln(x.bfloat16().data_ptr(), Wp.data_ptr(), b2.data_ptr(), None, ob.data_ptr(), of.data_ptr(), M, N, K,
                                          K, K, 0, N, N, 0, C.byref(lnp), L.stream_ptr(A.device)), f"ln tile {t}")
        torch.cuda.synchronize()
        ref = F.layer_norm(x, (K,), gm, bt, 1e-5) @ Wf.t() + bias
        scale = max(ref.abs().max().item(), 1.0)
        assert (of - ref).abs().max().item() <= 6e-3 * scale, t
        # residual through a LayerNorm + row statistics out (N = 768 only: that is where the encoders use it)
        N2 = 768
        if N2 % bn == 0:
            W2 = (torch.randn(N2, K, generator=g) / K ** 0.5).to(DEV).bfloat16()
            r = (torch.randn(M, N2, generator=g) * 1.5 - 0.2).to(DEV)
            gm2, bt2, b3 = (1 + 0.2 * torch.randn(N2, generator=g)).to(DEV), (0.1 * torch.randn(N2, generator=g)).to(DEV), torch.randn(N2, generator=g).to(DEV)
            rs = _split_stats(r, 12)
            of2, ob2 = torch.empty(M, N2, device=DEV), torch.empty(M, N2, dtype=torch.bfloat16, device=DEV)
            sto = torch.full((M, N2 // 32, 2), float("nan"), device=DEV)
            ln2 = L.GemmLn()
            ln2.r_stats, ln2.r_gamma, ln2.r_beta, ln2.r_parts, ln2.out_stats = rs.data_ptr(), gm2.data_ptr(), bt2.data_ptr(), 12, sto.data_ptr()
            ln2.a_eps = ln2.r_eps = 1e-12
            ln2.width, ln2.tile_cfg = N2, t
            rc = L.lib().ufnd_gemm_bf16_ln(A.data_ptr(), W2.data_ptr(), b3.data_ptr(), r.data_ptr(), ob2.data_ptr(), of2.data_ptr(), M, N2, K,
                                           K, K, N2, N2, N2, 0, C.byref(ln2), L.stream_ptr(A.device))
            if rc != 0:          # a tile whose wave columns are not whole 32-column groups has no statistics epilogue
                assert b"out_stats" in L.lib().ufnd_last_error(), (t, L.lib().ufnd_last_error())
                continue
            torch.cuda.synchronize()
            ref2 = A.float() @ W2.float().t() + b3 + F.layer_norm(r, (N2,), gm2, bt2, 1e-12)
            sc2 = max(ref2.abs().max().item(), 1.0)
            assert (of2 - ref2).abs().max().item() <= 2e-3 * sc2, t
            assert not torch.isnan(sto).any() and (sto - _split_stats(of2, N2 // 32)).abs().max().item() <= 1e-4 * sc2 * sc2 * 32, t
    # ids that are not part of the library: rejected, nothing launched
    built = {t[0] for t in tiles}
    A = torch.zeros(64, 64, dtype=torch.bfloat16, device=DEV)
    o = torch.zeros(64, 64, device=DEV)
    for bad in [t for t in range(L.lib().ufnd_gemm_bf16_tile_count()) if t not in built][:3] + [L.lib().ufnd_gemm_bf16_tile_count(), 100, 122, 222]:
        rc = L.lib().ufnd_gemm_bf16_ex(A.data_ptr(), A.data_ptr(), None, None, None, o.data_ptr(), 64, 64, 64, 64, 64, 0, 0, 64, 0, bad, None)
        assert rc == 1 and b"not part of this library" in L.lib().ufnd_last_error(), bad


def test_gemm_rejects_bad_shapes():
    L = _lib()
    A = torch.zeros(8, 96, dtype=torch.bfloat16, device=DEV)
    W = torch.zeros(64, 96, dtype=torch.bfloat16, device=DEV)
    o = torch.zeros(8, 64, device=DEV)
    rc = L.lib().ufnd_gemm_bf16(A.data_ptr(), W.data_ptr(), None, None, None, o.data_ptr(), 8, 64, 96, 96, 96, 0, 0, 64, 0, None)
    assert rc == 1 and b"K%64" in L.lib().ufnd_last_error()


@pytest.mark.parametrize("M,H,eps", [(515, 768, 1e-12), (50, 768, 1e-5), (7, 512, 1e-5)])
def test_layernorm(M, H, eps):
    L = _lib()
    g = torch.Generator().manual_seed(M)
    x = (torch.randn(M, H, generator=g) * 3 + 0.5).to(DEV)
    gm, bt = torch.randn(H, generator=g).to(DEV), torch.randn(H, generator=g).to(DEV)
    ob = torch.empty(M, H, dtype=torch.bfloat16, device=DEV)
    of = torch.empty(M, H, device=DEV)
    L.check(L.lib().ufnd_layernorm(x.data_ptr(), H, gm.data_ptr(), bt.data_ptr(), ob.data_ptr(), of.data_ptr(), M, H, eps,
                                   L.stream_ptr(x.device)), "ln")
    ref = F.layer_norm(x, (H,), gm, bt, eps)
    assert (of - ref).abs().max().item() <= 2e-5
    assert (ob.float() - ref).abs().max().item() <= 2 ** -8 * ref.abs().max().item() + 2e-5


def _gemm_ln(A, W, bias, M, N, K, act=0, residual=None, a_stats=None, colsum=None, r_stats=None, r_gamma=None, r_beta=None,
             want_stats=False, eps=1e-5, width=768):
    L = _lib()
    ob = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    of = torch.empty(M, N, dtype=torch.float32, device=DEV)
    parts = L.lib().ufnd_gemm_bf16_stat_parts(M, N, K)
    st = torch.full((M, max(parts, 1), 2), float("nan"), device=DEV) if want_stats else None
    ln = L.GemmLn()
    ln.a_stats, ln.colsum, ln.r_stats, ln.r_gamma, ln.r_beta = L.ptr(a_stats), L.ptr(colsum), L.ptr(r_stats), L.ptr(r_gamma), L.ptr(r_beta)
    ln.out_stats = L.ptr(st)
    ln.a_parts = a_stats.shape[1] if a_stats is not None else 0
    ln.r_parts = r_stats.shape[1] if r_stats is not None else 0
    ln.a_eps = ln.r_eps = eps
    ln.width = width
    L.check(L.lib().ufnd_gemm_bf16_ln(A.data_ptr(), W.data_ptr(), L.ptr(bias), L.ptr(residual), ob.data_ptr(), of.data_ptr(), M, N, K,
                                      K, K, N if residual is not None else 0, N, N, act, C.byref(ln), L.stream_ptr(A.device)), "gemm_ln")
    torch.cuda.synchronize()
    return ob, of, st


def _split_stats(x, parts):
    """(M, H) fp32 -> (M, parts, 2) partial {sum, sumsq} over `parts` equal column slices."""
    M, H = x.shape
    xs = x.view(M, parts, H // parts)
    return torch.stack([xs.sum(-1), (xs * xs).sum(-1)], -1).contiguous()


@pytest.mark.parametrize("M,N,K,act,parts", [(4096, 2304, 768, 0, 12), (4096, 3072, 768, 1, 12), (1600, 3072, 768, 2, 24),
                                             (1600, 2304, 768, 0, 2), (130, 2304, 768, 0, 12)])
def test_gemm_ln_folds_the_layernorm_of_its_input(M, N, K, act, parts):
    """LayerNorm(x) W^T + b through ufnd_gemm_bf16_ln (un-normalised bf16 rows + row statistics + W*gamma)
    against the fp32 torch evaluation; same bounds as the plain GEMM (one more bf16 rounding of x)."""
    g = torch.Generator().manual_seed(M + N)
    x = (torch.randn(M, K, generator=g) * 2.0 + 0.3).to(DEV)
    Wf = (torch.randn(N, K, generator=g) / K ** 0.5).to(DEV)
    b, gm, bt = (torch.randn(N, generator=g).to(DEV), (1 + 0.2 * torch.randn(K, generator=g)).to(DEV), (0.1 * torch.randn(K, generator=g)).to(DEV))
    Wp = (Wf * gm[None, :]).bfloat16()
    colsum = Wp.float().sum(1).contiguous()
    bias = (b + Wf @ bt).contiguous()
    ob, of, _ = _gemm_ln(x.bfloat16(), Wp, bias, M, N, K, act=act, a_stats=_split_stats(x, parts), colsum=colsum, eps=1e-5, width=K)
    ref = F.layer_norm(x, (K,), gm, bt, 1e-5) @ Wf.t() + b
    ref = F.gelu(ref) if act == 1 else (ref * torch.sigmoid(1.702 * ref) if act == 2 else ref)
    scale = max(ref.abs().max().item(), 1.0)
    e32, e16 = (of - ref).abs().max().item(), (ob.float() - ref).abs().max().item()
    print(f"gemm_ln fold {M}x{N}x{K} act={act}: f32 err {e32:.2e} bf16 err {e16:.2e} scale {scale:.2f}")
    assert e32 <= 6e-3 * scale and e16 <= (6e-3 + 2 ** -8) * scale, (e32, e16)


@pytest.mark.parametrize("M,N,K", [(4096, 768, 768), (4096, 768, 3072), (1600, 768, 3072), (131, 768, 768)])
def test_gemm_ln_residual_through_layernorm_and_row_statistics(M, N, K):
    """out = A W^T + b + LayerNorm(r) * gamma + beta, plus the partial row statistics of out."""
    L = _lib()
    g = torch.Generator().manual_seed(M + K)
    A = torch.randn(M, K, generator=g).to(DEV).bfloat16()
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(DEV).bfloat16()
    b = torch.randn(N, generator=g).to(DEV)
    r = (torch.randn(M, N, generator=g) * 1.5 - 0.2).to(DEV)
    gm, bt = (1 + 0.2 * torch.randn(N, generator=g)).to(DEV), (0.1 * torch.randn(N, generator=g)).to(DEV)
    ob, of, st = _gemm_ln(A, W, b, M, N, K, residual=r, r_stats=_split_stats(r, 12), r_gamma=gm, r_beta=bt, want_stats=True,
                          eps=1e-12, width=N)
    ref = A.float() @ W.float().t() + b + F.layer_norm(r, (N,), gm, bt, 1e-12)
    scale = max(ref.abs().max().item(), 1.0)
    assert (of - ref).abs().max().item() <= 2e-3 * scale
    assert (ob.float() - ref).abs().max().item() <= (2e-3 + 2 ** -8) * scale
    parts = L.lib().ufnd_gemm_bf16_stat_parts(M, N, K)
    assert parts == N // 32 and st.shape[1] == parts and not torch.isnan(st).any()
    assert (st - _split_stats(of, parts)).abs().max().item() <= 1e-4 * scale * scale * 32
    tot = st.sum(1)
    assert (tot[:, 0] - of.sum(1)).abs().max().item() <= 2e-3 * N ** 0.5 * scale
    assert ((tot[:, 1] - (of * of).sum(1)).abs() / (of * of).sum(1)).max().item() <= 1e-5
    # plain residual (no LayerNorm on it) through the same entry point
    ob2, of2, _ = _gemm_ln(A, W, b, M, N, K, residual=r, eps=1e-12, width=N)
    assert (of2 - (A.float() @ W.float().t() + b + r)).abs().max().item() <= 2e-3 * scale


@pytest.mark.parametrize("which", ["bert", "vit"])
def test_folded_layernorm_encoders_match_the_unfolded_ones(which):
    """fold_ln=True (no LayerNorm kernels between Linears) against fold_ln=False (one LayerNorm kernel per
    LayerNorm), same weights with non-trivial gamma / beta, 3 layers: hidden states agree to bf16 noise."""
    from ultrafnd_git_amd.encoders import BertTextEncoder, ClipVisualEncoder
    g = torch.Generator().manual_seed(5)
    if which == "bert":
        a, b = BertTextEncoder(layers=3, vocab_size=1000, fold_ln=True), BertTextEncoder(layers=3, vocab_size=1000, fold_ln=False)
    else:
        a, b = ClipVisualEncoder(layers=3, fold_ln=True), ClipVisualEncoder(layers=3, fold_ln=False)
    sd = a.state_dict()
    for k in sd:
        if "LayerNorm" in k or "layer_norm" in k or "layrnorm" in k or "layernorm" in k:
            sd[k] = sd[k] + (0.3 if k.endswith("weight") else 0.2) * torch.randn(sd[k].shape, generator=g)
        elif k.endswith("bias"):
            sd[k] = 0.05 * torch.randn(sd[k].shape, generator=g)
    a.load_state_dict(sd); b.load_state_dict(sd)
    a, b = a.to(DEV), b.to(DEV)
    if which == "bert":
        ids = torch.randint(0, 1000, (5, 128), generator=g)
        mask = (torch.arange(128)[None, :] < torch.tensor([128, 77, 16, 100, 1])[:, None]).int()
        ha, hb = a.last_hidden_state(ids, mask).clone(), b.last_hidden_state(ids, mask).clone()
        fa, fb = a(ids, mask).clone(), b(ids, mask).clone()
        assert "st" in a._workbufs(5, 128) and "st" not in b._workbufs(5, 128)
    else:
        fr = torch.randn(3, 2, 3, 224, 224, generator=g)
        ha, hb = a._run(fr)[0].clone(), b._run(fr)[0].clone()
        fa, fb = a(fr).clone(), b(fr).clone()
        assert "st0" in a._workbufs(3, 2) and "st0" not in b._workbufs(3, 2)
    eh = (ha - hb).abs().max().item() / hb.abs().max().item()
    ef = (fa - fb).abs().max().item()
    print(f"{which}: folded vs unfolded hidden rel err {eh:.2e}, feature err {ef:.2e}")
    assert eh <= 2e-2 and ef <= 2e-3, (eh, ef)


@pytest.mark.parametrize("B,Lq,masked", [(3, 128, True), (2, 50, False), (2, 40, True), (2, 512, True), (1, 130, True)])
def test_attention(B, Lq, masked):
    L = _lib()
    heads, H = 12, 768
    g = torch.Generator().manual_seed(Lq)
    qkv = (torch.randn(B * Lq, 3 * H, generator=g) * 1.5).to(DEV).bfloat16()
    mask = None
    if masked:
        lens = torch.randint(1, Lq + 1, (B,), generator=g)
        mask = (torch.arange(Lq)[None] < lens[:, None]).to(torch.int32)
        if B > 1:
            mask[1] = 0                                   # fully masked row: HF semantics = uniform average
        mask = mask.to(DEV)
    ctx = torch.empty(B * Lq, H, dtype=torch.bfloat16, device=DEV)
    L.check(L.lib().ufnd_attention_bf16(qkv.data_ptr(), L.ptr(mask), ctx.data_ptr(), B, Lq, heads, L.stream_ptr(qkv.device)), "attn")
    torch.cuda.synchronize()
    q, k, v = [t.view(B, Lq, heads, 64).transpose(1, 2) for t in qkv.float().split(H, dim=1)]
    s = (q @ k.transpose(-1, -2)) * 0.125
    if mask is not None:
        s = s + (1.0 - mask[:, None, None, :].float()) * torch.finfo(torch.float32).min
    ref = (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(B * Lq, H)
    err = (ctx.float() - ref).abs().max().item()
    print(f"attention B={B} L={Lq} masked={masked}: max-abs-err {err:.3e} (|ref| max {ref.abs().max().item():.2f})")
    assert torch.isfinite(ctx.float()).all()
    assert err <= 3e-2, err          # P and the output are rounded to bf16 (2^-8 relative each); |V| <= ~7


def test_attention_masks_that_are_not_a_prefix():
    """Key blocks that are masked as a whole are passed over once a query has seen a live key (attn_softmax.hpp:
    masked_block_is_noop) -- a padded suffix is the common case, but the rule has to hold for any mask: a masked FIRST block
    (left padding: nothing live seen yet, the block must be computed), a masked block between two live ones, live keys only in
    the last block, a single live key, and no live key at all (HF's uniform average)."""
    L = _lib()
    B, Lq, heads, H = 6, 512, 12, 768
    g = torch.Generator().manual_seed(99)
    qkv = (torch.randn(B * Lq, 3 * H, generator=g) * 1.5).to(DEV).bfloat16()
    mask = torch.zeros(B, Lq, dtype=torch.int32)
    mask[0, 200:] = 1                      # left padding: blocks 0-2 masked as a whole, block 3 partly
    mask[1, :64] = 1; mask[1, 128:192] = 1; mask[1, 448:] = 1      # masked blocks between live ones
    mask[2, 500:] = 1                      # live keys in the last block only
    mask[3, 77] = 1                        # one live key
    mask[4, :] = 0                         # none
    mask[5, ::2] = 1                       # no block masked as a whole
    mask = mask.to(DEV)
    ctx = torch.empty(B * Lq, H, dtype=torch.bfloat16, device=DEV)
    L.check(L.lib().ufnd_attention_bf16(qkv.data_ptr(), mask.data_ptr(), ctx.data_ptr(), B, Lq, heads, L.stream_ptr(qkv.device)), "attn")
    torch.cuda.synchronize()
    q, k, v = [t.view(B, Lq, heads, 64).transpose(1, 2) for t in qkv.float().split(H, dim=1)]
    s = (q @ k.transpose(-1, -2)) * 0.125 + (1.0 - mask[:, None, None, :].float()) * torch.finfo(torch.float32).min
    ref = (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(B, Lq, H)
    got = ctx.float().view(B, Lq, H)
    assert torch.isfinite(got).all()
    for b in range(B):
        err = (got[b] - ref[b]).abs().max().item()
        assert err <= 3e-2, (b, err)


@pytest.mark.parametrize("tag", ["bert2_L128", "bert2_L512", "bert2_L40"])
def test_text_features_match_third_party(tag):
    from oracle import encoders_ref as E
    from ultrafnd_git_amd.encoders import BertTextEncoder
    z = load_npz("tier_b.npz")
    meta = json.loads(str(z[f"{tag}/meta"]))
    w = E.seeded_weights(E.bert_shapes(layers=meta["layers"], vocab=meta["vocab"]), meta["weight_seed"])
    enc = BertTextEncoder(layers=meta["layers"], vocab_size=meta["vocab"])
    enc.load_state_dict(w)
    enc = enc.to(DEV)
    ids, mask = torch.from_numpy(z[f"{tag}/ids"]), torch.from_numpy(z[f"{tag}/mask"])
    hid = enc.last_hidden_state(ids, mask)
    n0 = int(mask[0].sum())
    eh = np.abs(hid[0, :n0].cpu().numpy() - z[f"{tag}/hidden_row0"]).max()
    feat = enc(ids, mask).cpu().numpy()
    print(f"{tag}: hidden max-abs-err {eh:.3e}")
    # bounds = 2 x measured (4.8e-4 / 3.3e-3 / 5.4e-6 at worst over the three fixtures): max-abs, rel-L2, 1 - cos
    assert_features_close(feat, z[f"{tag}/features"], 1.0e-3, 7.0e-3, 1.2e-5, what=f"{tag} features vs third party")
    # hidden entries are of order 1 after the final LayerNorm; two layers of bf16 GEMMs: 2^-9 * sqrt(8) ~ 6e-3 rms, tail x5
    assert eh <= 3e-2, eh


@pytest.mark.parametrize("tag", ["vit2_F1", "vit2_F4"])
def test_visual_features_match_third_party(tag):
    from oracle import encoders_ref as E
    from ultrafnd_git_amd.encoders import ClipVisualEncoder
    z = load_npz("tier_b.npz")
    meta = json.loads(str(z[f"{tag}/meta"]))
    w = E.seeded_weights(E.vit_shapes(layers=meta["layers"]), meta["weight_seed"])
    enc = ClipVisualEncoder(layers=meta["layers"])
    enc.load_state_dict(w)
    enc = enc.to(DEV)
    frames = E.synthetic_frames(meta["frame_seed"], meta["B"], meta["F"])
    feat = enc(frames).cpu().numpy()
    ef = np.abs(feat - z[f"{tag}/features"]).max()
    ee = np.abs(enc.image_embeds(frames.reshape(-1, 3, 224, 224)).cpu().numpy() - z[f"{tag}/image_embeds"]).max()
    print(f"{tag}: image_embeds max-abs-err {ee:.3e}")
    # bounds = 2 x measured (5.9e-4 / 4.0e-3 / 8.0e-6 at worst over the two fixtures)
    assert_features_close(feat, z[f"{tag}/features"], 1.2e-3, 8.0e-3, 1.6e-5, what=f"{tag} features vs third party")


def test_encode_fields_batched():
    """encode_fields: title + OCR + <=10 comments per record, empty parts skipped, record without parts -> 0."""
    from oracle import encoders_ref as E
    from ultrafnd_git_amd.encoders import BertTextEncoder
    w = E.seeded_weights(E.bert_shapes(layers=2, vocab=1000), 21)
    enc = BertTextEncoder(layers=2, vocab_size=1000)
    enc.load_state_dict(w)
    enc = enc.to(DEV)
    N, Mx, Lq = 5, 12, 24
    g = torch.Generator().manual_seed(3)
    ids = torch.randint(0, 1000, (N, Mx, Lq), generator=g)
    lens = torch.randint(1, Lq + 1, (N, Mx), generator=g)
    mask = (torch.arange(Lq)[None, None] < lens[..., None]).long()
    valid = (torch.rand(N, Mx, generator=g) < 0.6).long()
    valid[1] = 0                                           # a record with no title / OCR / comments
    valid[2] = 1                                           # a record with all 12 parts
    got = enc.encode_fields(ids, mask, valid).cpu()
    ref = torch.zeros(N, 768)
    for n in range(N):
        sel = valid[n].bool()
        if sel.any():
            ref[n] = E.field_mean_l2(E.text_features(w, ids[n][sel], mask[n][sel]))
    err = (got - ref).abs().max().item()
    print(f"encode_fields max-abs-err {err:.3e}")
    assert got[1].abs().max().item() == 0.0
    assert err <= 4e-3, err


def test_full_depth_encoders_vs_oracle():
    """12-layer BERT-base / ViT-B/32 geometry (small vocab to keep the CPU oracle quick)."""
    from oracle import encoders_ref as E
    from ultrafnd_git_amd.encoders import BertTextEncoder, ClipVisualEncoder
    w = E.seeded_weights(E.bert_shapes(layers=12, vocab=1000), 41)
    enc = BertTextEncoder(layers=12, vocab_size=1000)
    enc.load_state_dict(w)
    enc = enc.to(DEV)
    ids, mask = E.synthetic_tokens(141, 4, 128, vocab=1000)
    ft = enc(ids, mask).cpu()
    et = (ft - E.text_features(w, ids, mask)).abs().max().item()
    assert_features_close(ft, E.text_features(w, ids, mask), 2.2e-3, 1.6e-2, 6.5e-5, what="12-layer BERT features")     # measured 1.1e-3 / 8.0e-3 / 3.2e-5
    w = E.seeded_weights(E.vit_shapes(layers=12), 42)
    venc = ClipVisualEncoder(layers=12)
    venc.load_state_dict(w)
    venc = venc.to(DEV)
    fr = E.synthetic_frames(142, 2, 1)
    fv = venc(fr).cpu()
    ev = (fv - E.visual_features(w, fr)).abs().max().item()
    assert_features_close(fv, E.visual_features(w, fr), 2.0e-3, 1.4e-2, 5.0e-5, what="12-layer ViT features")          # measured 7.4e-4 / 5.5e-3 / 1.5e-5 (fp32 stream)
    print(f"12-layer feature max-abs-err: text {et:.3e} visual {ev:.3e}")


def test_end_to_end_logit_error():
    """north_star's bound: logits of the whole GPU path (bf16 encoders -> fp32 fusion head) within 1e-3 of
    the all-fp32 CPU path on the same synthetic FakeSV batch and weights."""
    from oracle import encoders_ref as E
    from oracle import tier_a as O
    from ultrafnd_git_amd.classifier import DeepTruthClassifier
    from ultrafnd_git_amd.encoders import BertTextEncoder, ClipVisualEncoder
    from ultrafnd_git_amd.fusion import CrossModalTransformer
    B, Lq = 4, 128
    wt = E.seeded_weights(E.bert_shapes(layers=12, vocab=1000), 41)
    wv = E.seeded_weights(E.vit_shapes(layers=12), 42)
    ids, mask = E.synthetic_tokens(7, B, Lq, vocab=1000)
    frames = E.synthetic_frames(8, B, 1)
    batch = O.seeded_batch(9, B)
    fus_sd, clf_sd = O.seeded_params(1234)
    # CPU fp32 reference path
    ref_b = dict(batch)
    ref_b["text_features"] = E.text_features(wt, ids, mask)
    ref_b["visual_features"] = E.visual_features(wv, frames)
    ref = O.forward_batch(fus_sd, clf_sd, ref_b)
    # GPU path
    tenc, venc = BertTextEncoder(layers=12, vocab_size=1000), ClipVisualEncoder(layers=12)
    tenc.load_state_dict(wt); venc.load_state_dict(wv)
    tenc, venc = tenc.to(DEV), venc.to(DEV)
    fusion, clf = CrossModalTransformer(), DeepTruthClassifier()
    fusion.load_state_dict(fus_sd); clf.load_state_dict(clf_sd)
    fusion, clf = fusion.to(DEV).eval(), clf.to(DEV).eval()
    feats = {k: batch[k].to(DEV) for k in ("audio_features", "temporal_features", "gnn_feat")}
    feats["text_features"] = tenc(ids, mask)
    feats["visual_features"] = venc(frames)
    with torch.no_grad():
        fo = fusion(feats)
        co = clf(fo["fused"], batch["aux"].to(DEV))
    err = (co["logits"].cpu() - ref["logits"]).abs().max().item()
    perr = (co["probs"].cpu() - ref["probs"]).abs().max().item()
    print(f"end-to-end logits max-abs-err {err:.3e} (probs {perr:.3e}); |logits| max {ref['logits'].abs().max().item():.3f}")
    assert err <= 1e-3, err


@pytest.mark.parametrize("with_ln", [False, True])
def test_fused_qkv_attention_kernel_is_bit_identical_to_gemm_plus_attention(with_ln):
    """ufnd_qkv_attention_bf16 (one launch: projection of one sample x two heads, bf16 rounding into LDS, attention) against
    ufnd_gemm_bf16[_ln] + ufnd_attention_bf16 on the same operands: every ctx element identical; masks incl. a fully
    masked sample and a one-token sample; an odd sample count (198 workgroups)."""
    L = _lib()
    B, Lq, heads, H = 33, 128, 12, 768
    g = torch.Generator().manual_seed(17)
    x = (torch.randn(B * Lq, H, generator=g) * 1.3 + 0.1).to(DEV)
    Wf = (torch.randn(3 * H, H, generator=g) / H ** 0.5).to(DEV)
    bias = (0.1 * torch.randn(3 * H, generator=g)).to(DEV)
    lens = torch.randint(1, Lq + 1, (B,), generator=g)
    lens[0], lens[1], lens[2] = Lq, 1, 0
    mask = (torch.arange(Lq)[None] < lens[:, None]).to(torch.int32).to(DEV).contiguous()
    ln = None
    if with_ln:
        gm, bt = (1 + 0.2 * torch.randn(H, generator=g)).to(DEV), (0.1 * torch.randn(H, generator=g)).to(DEV)
        W = (Wf * gm[None, :]).bfloat16()
        cs = W.float().sum(1).contiguous()
        b2 = (bias + Wf @ bt).contiguous()
        st = _split_stats(x, 24)
        ln = L.GemmLn()
        ln.a_stats, ln.colsum, ln.a_parts, ln.a_eps, ln.r_eps, ln.width = st.data_ptr(), cs.data_ptr(), 24, 1e-12, 1e-12, H
    else:
        W, b2 = Wf.bfloat16(), bias
    xb = x.bfloat16()
    qkv = torch.empty(B * Lq, 3 * H, dtype=torch.bfloat16, device=DEV)
    ctx_ref = torch.empty(B * Lq, H, dtype=torch.bfloat16, device=DEV)
    ctx = torch.full((B * Lq, H), float("nan"), dtype=torch.bfloat16, device=DEV)
    s = L.stream_ptr(x.device)
    if with_ln:
        L.check(L.lib().ufnd_gemm_bf16_ln(xb.data_ptr(), W.data_ptr(), b2.data_ptr(), None, qkv.data_ptr(), None, B * Lq, 3 * H, H, H, H, 0, 3 * H, 0, 0,
                                          C.byref(ln), s), "gemm_ln")
    else:
        L.check(L.lib().ufnd_gemm_bf16(xb.data_ptr(), W.data_ptr(), b2.data_ptr(), None, qkv.data_ptr(), None, B * Lq, 3 * H, H, H, H, 0, 3 * H, 0, 0, s), "gemm")
    L.check(L.lib().ufnd_attention_bf16(qkv.data_ptr(), mask.data_ptr(), ctx_ref.data_ptr(), B, Lq, heads, s), "attention")
    L.check(L.lib().ufnd_qkv_attention_bf16(xb.data_ptr(), W.data_ptr(), b2.data_ptr(), mask.data_ptr(), ctx.data_ptr(), B, Lq, heads, H, H,
                                            C.byref(ln) if ln is not None else None, s), "qkv_attention")
    torch.cuda.synchronize()
    assert torch.isfinite(ctx.float()).all()
    assert torch.equal(ctx, ctx_ref), (ctx.float() - ctx_ref.float()).abs().max().item()
    # masks that are not a prefix (the key-block skip of attn_softmax.hpp must hold for any mask): left padding with the whole first
    # 64-key block masked, a masked second block behind a live first one, one live key in the second block only
    m2 = torch.ones(B, Lq, dtype=torch.int32)
    m2[0, :70] = 0
    m2[1, 64:] = 0
    m2[2, :] = 0; m2[2, 100] = 1
    m2[3, ::3] = 0
    m2 = m2.to(DEV).contiguous()
    ctx.fill_(float("nan"))
    L.check(L.lib().ufnd_attention_bf16(qkv.data_ptr(), m2.data_ptr(), ctx_ref.data_ptr(), B, Lq, heads, s), "attention")
    L.check(L.lib().ufnd_qkv_attention_bf16(xb.data_ptr(), W.data_ptr(), b2.data_ptr(), m2.data_ptr(), ctx.data_ptr(), B, Lq, heads, H, H,
                                            C.byref(ln) if ln is not None else None, s), "qkv_attention")
    torch.cuda.synchronize()
    assert torch.isfinite(ctx.float()).all() and torch.equal(ctx, ctx_ref)
    qq = qkv.float().view(B, Lq, 3, heads, 64)[:4]
    sc = torch.einsum("bqhd,bkhd->bhqk", qq[:, :, 0], qq[:, :, 1]) * 0.125 + (1.0 - m2[:4, None, None, :].float()) * torch.finfo(torch.float32).min
    ref4 = torch.einsum("bhqk,bkhd->bqhd", torch.softmax(sc, -1), qq[:, :, 2]).reshape(4, Lq, H)
    assert (ctx.float().view(B, Lq, H)[:4] - ref4).abs().max().item() <= 3e-2
    # no mask at all == an all-ones mask
    ctx2 = torch.empty_like(ctx)
    ones = torch.ones_like(mask)
    L.check(L.lib().ufnd_qkv_attention_bf16(xb.data_ptr(), W.data_ptr(), b2.data_ptr(), None, ctx2.data_ptr(), B, Lq, heads, H, H,
                                            C.byref(ln) if ln is not None else None, s), "qkv_attention")
    L.check(L.lib().ufnd_qkv_attention_bf16(xb.data_ptr(), W.data_ptr(), b2.data_ptr(), ones.data_ptr(), ctx.data_ptr(), B, Lq, heads, H, H,
                                            C.byref(ln) if ln is not None else None, s), "qkv_attention")
    torch.cuda.synchronize()
    assert torch.equal(ctx, ctx2)
    # shapes the fused kernel is not built for are refused (the encoder then uses the two-launch form)
    rc = L.lib().ufnd_qkv_attention_bf16(xb.data_ptr(), W.data_ptr(), b2.data_ptr(), None, ctx.data_ptr(), B, 64, heads, H, H, None, s)
    assert rc == 1 and b"128-token" in L.lib().ufnd_last_error()


@pytest.mark.parametrize("fold", [True, False])
def test_text_encoder_with_fused_attention_is_bit_identical(fold):
    from oracle import encoders_ref as E
    from ultrafnd_git_amd.encoders import BertTextEncoder
    wt = E.seeded_weights(E.bert_shapes(layers=3, vocab=1000), 73)
    enc = BertTextEncoder(layers=3, vocab_size=1000, fold_ln=fold)
    enc.load_state_dict(wt)
    enc = enc.to(DEV)
    ids, mask = E.synthetic_tokens(173, 7, 128, vocab=1000, min_len=3)
    assert enc.fuse_qkv_attention
    a_h, a_f = enc.last_hidden_state(ids, mask).clone(), enc(ids, mask).clone()
    enc.fuse_qkv_attention = False
    b_h, b_f = enc.last_hidden_state(ids, mask).clone(), enc(ids, mask).clone()
    assert torch.equal(a_h, b_h) and torch.equal(a_f, b_f)
    assert (a_f.cpu() - E.text_features(wt, ids, mask)).abs().max().item() <= 4e-3


@pytest.mark.parametrize("fold", [True, False])
def test_unpadded_text_encoder_is_bit_identical_for_prefix_masks(fold):
    """BertTextEncoder(..., unpad=True) computes only the tokens the padding mask keeps (packed rows, per-sequence
    attention).  Padded positions never reach the pooling, so with prefix masks every feature is bit-identical to the
    padded run; arbitrary masks regroup the keys and agree to rounding; an all-masked row gives the zero vector."""
    from oracle import encoders_ref as E
    from ultrafnd_git_amd.encoders import BertTextEncoder
    wt = E.seeded_weights(E.bert_shapes(layers=3, vocab=1000), 71)
    enc = BertTextEncoder(layers=3, vocab_size=1000, fold_ln=fold)
    enc.load_state_dict(wt)
    enc = enc.to(DEV)
    g = torch.Generator().manual_seed(9)
    B, Lq = 9, 256                                    # the reference pads every string to max_length = 256
    ids = torch.randint(0, 1000, (B, Lq), generator=g)
    lens = torch.tensor([256, 1, 17, 64, 65, 128, 200, 31, 5])
    mask = (torch.arange(Lq)[None, :] < lens[:, None]).int()
    dense = enc(ids, mask).clone()
    packed = enc(ids, mask, unpad=True).clone()
    assert torch.equal(dense, packed)
    ref = E.text_features(wt, ids, mask)
    assert (packed.cpu() - ref).abs().max().item() <= 4e-3
    # a hole in the middle of a mask and an empty row
    mask2 = mask.clone()
    mask2[3, 10:20] = 0
    mask2[7] = 0
    d2, p2 = enc(ids, mask2).clone(), enc(ids, mask2, unpad=True).clone()
    assert (d2 - p2).abs().max().item() <= 2e-3
    assert torch.equal(p2[7], torch.zeros_like(p2[7])) and torch.equal(d2[[0, 1, 2, 4, 5, 6, 8]], p2[[0, 1, 2, 4, 5, 6, 8]])


def test_encode_fields_unpadded_equals_padded():
    from ultrafnd_git_amd.encoders import BertTextEncoder
    enc = BertTextEncoder(layers=2, vocab_size=500).to(DEV)
    g = torch.Generator().manual_seed(4)
    N, Mx, Lq = 6, 12, 64
    ids = torch.randint(0, 500, (N, Mx, Lq), generator=g)
    lens = torch.randint(1, 30, (N, Mx), generator=g)
    mask = (torch.arange(Lq)[None, None, :] < lens[..., None]).int()
    valid = (torch.rand(N, Mx, generator=g) < 0.7).int()
    valid[2] = 0
    a = enc.encode_fields(ids, mask, valid, unpad=False).clone()
    b = enc.encode_fields(ids, mask, valid, unpad=True).clone()
    assert torch.equal(a, b) and torch.equal(a[2], torch.zeros_like(a[2]))
