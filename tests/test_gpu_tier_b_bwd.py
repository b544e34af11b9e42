"""GPU: the Tier-B backward entry points (include/ultrafnd_hip.h, ABI v3), op by op, through the C ABI, against torch autograd
in fp32 on the SAME bf16-rounded operands.  The reference never trains its encoders (src/core_blocks/text_blocks.py:52,63), so
there is no reference vector for any of this: "parity unpinned by the reference" (DESIGN.md section 2); the whole-encoder
gradient test against the oracle's autograd is tests/test_gpu_encoder_train.py.

Tolerances: a bf16 operand carries 2^-9 relative rounding; products are accumulated in fp32, so a gradient computed from
bf16-rounded inputs agrees with the fp32 autograd of those same rounded inputs to accumulation noise (<= 1e-5 of the scale
for fp32 outputs); bf16 outputs add one rounding (2^-8 of the element)."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _L():
    from ultrafnd_git_amd import _lib as L
    return L


def _s():
    return _L().stream_ptr(torch.device(DEV, torch.cuda.current_device()))


def _bf(t):
    return t.to(torch.bfloat16)


def _transpose(src, rows_pad=None, colsum=False):
    """ufnd_transpose_bf16 of a (rows, cols) device tensor (bf16 or fp32) -> (cols, rows_pad) bf16 [, column sums]."""
    L = _L()
    rows, cols = src.shape
    rows_pad = rows_pad or ((rows + 63) // 64 * 64)
    dst = torch.full((cols, rows_pad), float("nan"), dtype=torch.bfloat16, device=DEV)
    cs = ws = None
    if colsum:
        cs = torch.empty(cols, dtype=torch.float32, device=DEV)
        ws = torch.empty(L.lib().ufnd_transpose_colsum_workspace_floats(rows_pad, cols), dtype=torch.float32, device=DEV)
    L.check(L.lib().ufnd_transpose_bf16(src.data_ptr(), int(src.dtype == torch.float32), rows, cols, src.stride(0), dst.data_ptr(), rows_pad, rows_pad,
                                        L.ptr(cs), L.ptr(ws), 0, _s()), "ufnd_transpose_bf16")
    return (dst, cs) if colsum else dst


@pytest.mark.parametrize("rows,cols", [(4096, 768), (1568, 3072), (100, 64), (1, 8)])
def test_transpose_and_column_sums(rows, cols):
    g = torch.Generator().manual_seed(rows + cols)
    x = _bf(torch.randn(rows, cols, generator=g)).to(DEV)
    xt, cs = _transpose(x, colsum=True)
    assert torch.equal(xt[:, :rows], x.t()) and (xt[:, rows:] == 0).all()                 # exact; pad columns zero
    ref = x.float().sum(0)
    assert (cs - ref).abs().max().item() <= 1e-5 * max(1.0, ref.abs().max().item()) * rows ** 0.5
    w = torch.randn(rows, cols, generator=g).to(DEV)                                        # fp32 source: cast on the way
    assert torch.equal(_transpose(w)[:, :rows], _bf(w).t())


@pytest.mark.parametrize("M,N,K", [(4096, 768, 3072), (4096, 3072, 768), (1600, 768, 2304), (200, 768, 768)])
def test_gemm_dgrad_plain_residual_and_activation_backward(M, N, K):
    """dx = dy Wt^T; + fp32 residual; x GELU'(pre) / quick-GELU'(pre): the three epilogues of the backward GEMM."""
    L = _L()
    g = torch.Generator().manual_seed(M + N)
    dy = _bf(torch.randn(M, K, generator=g)).to(DEV)
    wt = _bf(torch.randn(N, K, generator=g) / K ** 0.5).to(DEV)
    res = torch.randn(M, N, generator=g).to(DEV)
    pre = _bf(torch.randn(M, N, generator=g) * 1.5).to(DEV)
    ref = dy.float() @ wt.float().t()
    scale = ref.std().item()

    def run(residual=None, aux=None, act=0):
        of = torch.empty(M, N, device=DEV)
        ob = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
        L.check(L.lib().ufnd_gemm_bf16_dgrad(dy.data_ptr(), wt.data_ptr(), L.ptr(residual), L.ptr(aux), ob.data_ptr(), of.data_ptr(), M, N, K,
                                             K, K, N, N, N, N, act, _s()), "ufnd_gemm_bf16_dgrad")
        return of, ob
    of, ob = run()
    assert (of - ref).abs().max().item() <= 2e-5 * scale * K ** 0.5 and torch.equal(ob, _bf(of))
    of, _ = run(residual=res)
    assert (of - (ref + res)).abs().max().item() <= 2e-5 * scale * K ** 0.5 + 1e-6
    x = pre.float().requires_grad_(True)
    F.gelu(x).backward(torch.ones_like(x))
    of, _ = run(aux=pre, act=3)
    assert (of - ref * x.grad).abs().max().item() <= 1e-4 * scale              # the derivative itself is accurate to ~1e-6
    q = pre.float().requires_grad_(True)
    (q * torch.sigmoid(1.702 * q)).backward(torch.ones_like(q))
    of, _ = run(aux=pre, act=4)
    assert (of - ref * q.grad).abs().max().item() <= 1e-4 * scale
    assert L.lib().ufnd_gemm_bf16_dgrad(dy.data_ptr(), wt.data_ptr(), res.data_ptr(), pre.data_ptr(), None, of.data_ptr(), M, N, K, K, K, N, N, N, N, 3,
                                        _s()) == 1                                  # aux and residual are exclusive


@pytest.mark.parametrize("tokens,n_out,k_in", [(4096, 768, 768), (4096, 2304, 768), (4096, 768, 3072), (1568, 768, 3072), (16384, 3072, 768), (50, 768, 768)])
def test_gemm_wgrad_split_k_with_transposed_operands(tokens, n_out, k_in):
    """dW = dy^T x through ufnd_transpose_bf16 + ufnd_gemm_bf16_wgrad (token slices, slab reduce), overwrite and accumulate;
    identical bits on a second run (no atomics)."""
    L = _L()
    g = torch.Generator().manual_seed(tokens + n_out)
    dy = _bf(torch.randn(tokens, n_out, generator=g)).to(DEV)
    x = _bf(torch.randn(tokens, k_in, generator=g)).to(DEV)
    tp = (tokens + 63) // 64 * 64
    dyt, xt = _transpose(dy, tp), _transpose(x, tp)
    ws = torch.empty(L.lib().ufnd_gemm_bf16_wgrad_workspace_floats(n_out, k_in, tp), dtype=torch.float32, device=DEV)
    dw = torch.full((n_out, k_in), float("nan"), device=DEV)

    def run(acc):
        L.check(L.lib().ufnd_gemm_bf16_wgrad(dyt.data_ptr(), xt.data_ptr(), dw.data_ptr(), n_out, k_in, tp, tp, tp, k_in, ws.data_ptr(), acc, _s()),
                "ufnd_gemm_bf16_wgrad")
    run(0)
    ref = dy.float().t() @ x.float()
    scale = ref.std().item()
    first = dw.clone()
    assert (first - ref).abs().max().item() <= 2e-5 * scale * tokens ** 0.5
    run(1)
    assert (dw - 2 * ref).abs().max().item() <= 4e-5 * scale * tokens ** 0.5
    dw.fill_(float("nan"))
    run(0)
    assert torch.equal(dw, first)


@pytest.mark.parametrize("M,H,ld", [(4096, 768, 768), (37, 768, 768), (64, 768, 50 * 768), (300, 512, 512)])
def test_layernorm_backward(M, H, ld):
    L = _L()
    g = torch.Generator().manual_seed(M)
    xs = (torch.randn(M, ld, generator=g) * 2 + 0.5).to(DEV)                  # rows of stride ld (the ViT's CLS rows: stride 50 H)
    gamma = (1 + 0.3 * torch.randn(H, generator=g)).to(DEV)
    beta = torch.randn(H, generator=g).to(DEV)
    dy = torch.randn(M, H, generator=g).to(DEV)
    add = torch.randn(M, H, generator=g).to(DEV)
    x = xs[:, :H].clone().requires_grad_(True)
    gm = gamma.clone().requires_grad_(True)
    bt = beta.clone().requires_grad_(True)
    F.layer_norm(x, (H,), gm, bt, 1e-5).backward(dy)
    dx = torch.empty(M, H, device=DEV)
    dxb = torch.empty(M, H, dtype=torch.bfloat16, device=DEV)
    dg, db = torch.empty(H, device=DEV), torch.empty(H, device=DEV)
    ws = torch.empty(L.lib().ufnd_layernorm_bwd_workspace_floats(M, H), device=DEV)
    L.check(L.lib().ufnd_layernorm_bwd(xs.data_ptr(), ld, gamma.data_ptr(), dy.data_ptr(), H, add.data_ptr(), H, dx.data_ptr(), dxb.data_ptr(), H,
                                       dg.data_ptr(), db.data_ptr(), ws.data_ptr(), 0, M, H, 1e-5, _s()), "ufnd_layernorm_bwd")
    assert (dx - (x.grad + add)).abs().max().item() <= 2e-5 and torch.equal(dxb, _bf(dx))
    assert (dg - gm.grad).abs().max().item() <= 2e-5 * M ** 0.5 and (db - bt.grad).abs().max().item() <= 2e-5 * M ** 0.5
    L.check(L.lib().ufnd_layernorm_bwd(xs.data_ptr(), ld, gamma.data_ptr(), dy.data_ptr(), H, None, 0, dx.data_ptr(), None, H,
                                       dg.data_ptr(), db.data_ptr(), ws.data_ptr(), 1, M, H, 1e-5, _s()), "ufnd_layernorm_bwd")
    assert (dx - x.grad).abs().max().item() <= 2e-5 and (dg - 2 * gm.grad).abs().max().item() <= 4e-5 * M ** 0.5     # accumulate


def _attention_ref(qkv, mask, heads):
    """fp32 autograd reference on the bf16-rounded q | k | v rows (HF masking: additive finfo.min on masked keys)."""
    B, Lq, H3 = qkv.shape
    H = H3 // 3
    x = qkv.float().requires_grad_(True)
    q, k, v = (x[..., i * H:(i + 1) * H].view(B, Lq, heads, 64).transpose(1, 2) for i in range(3))
    s = (q @ k.transpose(-1, -2)) * 0.125
    if mask is not None:
        s = s + (1.0 - mask[:, None, None, :].float()) * torch.finfo(torch.float32).min
    ctx = (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(B, Lq, H)
    return x, ctx


@pytest.mark.parametrize("B,Lq,masked", [(3, 128, True), (4, 50, False), (2, 512, True), (2, 77, True), (1, 200, False)])
def test_attention_backward_vs_autograd(B, Lq, masked):
    """ufnd_attention_bf16_lse + ufnd_attention_bf16_bwd against softmax-attention autograd: 128 tokens (BERT), 50 (ViT), 512
    (configs[3]), ragged lengths that end inside a 64-row block; padded queries carry zero upstream gradient (as in the encoder:
    nothing downstream reads them)."""
    L = _L()
    heads, H = 12, 768
    g = torch.Generator().manual_seed(B * 1000 + Lq)
    qkv = _bf(torch.randn(B, Lq, 3 * H, generator=g)).to(DEV)
    mask = None
    if masked:
        lens = torch.randint(max(1, Lq // 4), Lq + 1, (B,), generator=g)
        lens[0] = Lq
        mask = (torch.arange(Lq)[None] < lens[:, None]).to(torch.int32).to(DEV)
    dctx = torch.randn(B, Lq, H, generator=g).to(DEV)
    if mask is not None:
        dctx = dctx * mask[..., None]
    dctx = _bf(dctx)
    ctx = torch.empty(B * Lq, H, dtype=torch.bfloat16, device=DEV)
    lse = torch.empty(B * Lq, heads, device=DEV)
    L.check(L.lib().ufnd_attention_bf16_lse(qkv.data_ptr(), L.ptr(mask), ctx.data_ptr(), lse.data_ptr(), B, Lq, heads, _s()), "attention_lse")
    x, ref_ctx = _attention_ref(qkv, mask, heads)
    live = torch.ones(B, Lq, 1, device=DEV) if mask is None else mask[..., None].float()
    assert ((ctx.view(B, Lq, H).float() - ref_ctx) * live).abs().max().item() <= 3e-2
    ref_ctx.backward(dctx.float())
    dqkv = torch.full((B * Lq, 3 * H), float("nan"), dtype=torch.bfloat16, device=DEV)
    ws = torch.empty(L.lib().ufnd_attention_bwd_workspace_floats(B, Lq, heads), device=DEV)
    L.check(L.lib().ufnd_attention_bf16_bwd(qkv.data_ptr(), ctx.data_ptr(), dctx.data_ptr(), lse.data_ptr(), L.ptr(mask), dqkv.data_ptr(), ws.data_ptr(),
                                            B, Lq, heads, _s()), "attention_bwd")
    got = dqkv.view(B, Lq, 3 * H).float()
    assert torch.isfinite(got).all()
    for name, sl in (("dq", slice(0, H)), ("dk", slice(H, 2 * H)), ("dv", slice(2 * H, 3 * H))):
        r = x.grad[..., sl]
        err = (got[..., sl] - r).abs().max().item()
        rel = ((got[..., sl] - r).norm() / r.norm().clamp_min(1e-20)).item()
        print(f"L={Lq} {name}: max-abs {err:.3e} (|ref| max {r.abs().max().item():.2f}), rel-L2 {rel:.3e}")
        # P, dS and the outputs are rounded to bf16 (2^-9 each), delta comes from the bf16 ctx: 1e-2 relative L2, max-abs 3 % of the largest entry
        assert rel <= 1.5e-2 and err <= 3e-2 * r.abs().max().item() + 1e-3, (name, err, rel)
    if mask is not None:      # masked keys receive no gradient at all
        dead = (1 - mask)[..., None].float()
        assert (got[..., H:] * dead).abs().max().item() == 0.0


def test_pooling_backwards_vs_autograd():
    L = _L()
    g = torch.Generator().manual_seed(9)
    B, Lq, H = 5, 40, 768
    hid = torch.randn(B, Lq, H, generator=g).to(DEV)
    lens = torch.tensor([40, 1, 17, 0, 33])
    mask = (torch.arange(Lq)[None] < lens[:, None]).to(torch.int32).to(DEV)
    df = torch.randn(B, H, generator=g).to(DEV)
    h = hid.clone().requires_grad_(True)
    m = mask[..., None].float()
    rep = (h * m).sum(1) / m.sum(1).clamp_min(1e-6)
    (rep / (rep.norm(dim=-1, keepdim=True) + 1e-9)).backward(df)
    dh = torch.empty(B * Lq, H, device=DEV)
    L.check(L.lib().ufnd_masked_meanpool_l2_bwd(hid.data_ptr(), mask.data_ptr(), df.data_ptr(), dh.data_ptr(), B, Lq, H, _s()), "meanpool_bwd")
    assert (dh.view(B, Lq, H) - h.grad).abs().max().item() <= 1e-5 * max(1.0, h.grad.abs().max().item())
    for Fr in (1, 4):
        e = torch.randn(B * Fr, 512, generator=g).to(DEV)
        dfe = torch.randn(B, 512, generator=g).to(DEV)
        ee = e.clone().requires_grad_(True)
        u = ee / (ee.norm(dim=-1, keepdim=True) + 1e-9)
        if Fr == 1:
            out = u
        else:
            mm = u.view(B, Fr, -1).mean(1)
            out = mm / (mm.norm(dim=-1, keepdim=True) + 1e-9)
        out.backward(dfe)
        de = torch.empty_like(e)
        L.check(L.lib().ufnd_l2norm_frames_bwd(e.data_ptr(), dfe.data_ptr(), de.data_ptr(), B, Fr, 512, _s()), "l2norm_frames_bwd")
        assert (de - ee.grad).abs().max().item() <= 1e-5 * max(1.0, ee.grad.abs().max().item()), Fr


def test_embedding_backwards_vs_autograd():
    """BERT: word (repeated ids, an id nobody uses), position and token-type gradients; ViT: class / position gradients and the
    patch rows.  ufnd_bert_embed / ufnd_vit_assemble with gamma = NULL return the raw sums the training forward keeps."""
    L = _L()
    g = torch.Generator().manual_seed(4)
    B, Lq, H, V, MP = 3, 20, 768, 50, 32
    ids = torch.randint(0, 12, (B, Lq), generator=g).to(DEV)            # few distinct ids: many repeats
    ds = torch.randn(B * Lq, H, generator=g).to(DEV)
    word = torch.randn(V, H, generator=g).to(DEV).requires_grad_(True)
    pos = torch.randn(MP, H, generator=g).to(DEV).requires_grad_(True)
    typ = torch.randn(2, H, generator=g).to(DEV).requires_grad_(True)
    s = word[ids] + pos[:Lq][None] + typ[0][None, None]
    raw = torch.empty(B * Lq, H, device=DEV)
    L.check(L.lib().ufnd_bert_embed(ids.data_ptr(), word.data_ptr(), pos.data_ptr(), typ.data_ptr(), None, None, None, raw.data_ptr(), B, Lq, H, V, 1e-12,
                                    _s()), "bert_embed raw")
    assert torch.equal(raw.view(B, Lq, H), s.detach())
    s.backward(ds.view(B, Lq, H))
    dw, dp, dt = (torch.full_like(t, float("nan")) for t in (word, pos, typ))
    L.check(L.lib().ufnd_bert_embed_bwd(ids.data_ptr(), ds.data_ptr(), dw.data_ptr(), dp.data_ptr(), dt.data_ptr(), B, Lq, H, V, MP, 2, _s()), "bert_embed_bwd")
    for got, ref in ((dw, word.grad), (dp, pos.grad), (dt, typ.grad)):
        assert (got - ref).abs().max().item() <= 1e-5 * max(1.0, ref.abs().max().item())
    N, P = 6, 49
    dsv = torch.randn(N * (P + 1), H, generator=g).to(DEV)
    dcls, dpos = torch.empty(H, device=DEV), torch.empty(P + 1, H, device=DEV)
    dpe = torch.empty(N * P, H, dtype=torch.bfloat16, device=DEV)
    L.check(L.lib().ufnd_vit_assemble_bwd(dsv.data_ptr(), dcls.data_ptr(), dpos.data_ptr(), dpe.data_ptr(), N, P, H, _s()), "vit_assemble_bwd")
    v3 = dsv.view(N, P + 1, H)
    assert (dpos - v3.sum(0)).abs().max().item() <= 1e-5 and (dcls - v3[:, 0].sum(0)).abs().max().item() <= 1e-5
    assert torch.equal(dpe.view(N, P, H), _bf(v3[:, 1:]))
