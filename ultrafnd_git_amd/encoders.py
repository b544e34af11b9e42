"""Frozen, forward-only encoders that produce the `text` (B,768) and `visual` (B,512) inputs of
the fusion step, MI355X-native.

  BertTextEncoder   replaces BERTContextEncoder.encode's model call + pooling
                    (src/core_blocks/text_blocks.py:71-101): BertModel(...).last_hidden_state
                    -> masked mean-pool -> L2-norm, batched over (B, L) instead of one
                    string at a time.  Tokenisation stays outside (ids/mask in), there is
                    no tokenizer vocabulary offline.
  ClipVisualEncoder the ViT-B/32 @224 geometry the reference points at
                    (configs/model_configs/semantic.yaml:2): pooled -> bias-free projection ->
                    L2-norm; multi-frame = mean over frames, L2-norm again
                    (text_blocks.py:126-128 idiom).

Weights keep the third-party `state_dict` names so real checkpoints load unchanged; fp32
masters are converted once to bf16 MFMA operands (Q/K/V stacked into one (3H,H) matrix).
Every Linear is `ufnd_gemm_bf16` with fused bias / GELU / residual epilogues; the residual
stream and all LayerNorm / softmax statistics stay fp32.  No CPU path.

LayerNorm folding (`fold_ln=True`, the default): no LayerNorm kernel runs between two Linears.
The GEMM that produces a residual-stream row also writes its bf16 rounding and partial
{sum, sum of squares}; the Linear that consumes LayerNorm(row) multiplies the UN-normalised bf16
row by W * gamma and applies  rstd (acc - mean colsum) + (b + W beta)  in its epilogue
(`ufnd_gemm_bf16_ln`); BERT's post-LN residual LayerNorm(row) * gamma + beta is evaluated on the fly
in the epilogue of the GEMM that adds it.  `fold_ln=False` keeps one LayerNorm kernel per LayerNorm.

Residual stream dtype (`residual_dtype`, folded encoders only).  "fp32": the stream is kept in fp32 next to the bf16
rounding the consumers read -- a residual GEMM moves 4 B in and 4 + 2 B out per element.  "bf16" (default): the rounding IS the stream
-- the residual operand is the bf16 copy the previous GEMM wrote for its consumer, no fp32 copy is written (except by the last
layer, for the final LayerNorm), 2 B in and 2 B out per element: 125 -> 53 MB per out-projection launch at 16,384 rows.  The
stream then carries 8 significant bits per layer; row statistics are still taken from the fp32 sums before rounding.
Measured at BASELINE's sizes (tests/test_gpu_fullsize.py): BERT features 9.9e-4 max-abs against 1.1e-3 with the fp32 stream,
hidden states 9.3e-3 against 8.7e-3 relative RMS after 12 layers; ViT (pre-LN: the stream is never re-normalised) features
7.4e-4 against 5.8e-4, hidden 8.7e-3 against 5.4e-3; end-to-end logits 3.2e-5 against 2.7e-5 (bound 1e-3) -- for
out-projection launches 49 -> 31 us and FFN2 100 -> 83 us at 16,384 rows, the step +7 %.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, Optional, Tuple

import torch
import torch.nn as nn

from . import _lib as L

ACT_NONE, ACT_GELU, ACT_QUICK_GELU = 0, 1, 2


def _bf16(t: torch.Tensor) -> torch.Tensor:
    """fp32 device tensor -> bf16 copy through the library's own cast kernel."""
    t = t.contiguous()
    out = torch.empty(t.shape, dtype=torch.bfloat16, device=t.device)
    L.check(L.lib().ufnd_cast_bf16(t.data_ptr(), out.data_ptr(), t.numel(), L.stream_ptr(t.device)), "ufnd_cast_bf16")
    return out


class _EncoderBase(nn.Module):
    """Weight store with third-party key names + lazily packed bf16 operands + per-shape buffers."""

    # Largest |mean| / std of a row whose LayerNorm may be folded into the next GEMM.  Folding rounds the UN-normalised
    # row to bf16, so its error relative to the normalised output is about (1 + |mean|/std) * 2^-9: at 4 it is still
    # below the bf16 floor of the materialised path; beyond, the encoder falls back to one LayerNorm kernel per LayerNorm.
    FOLD_GUARD_MAX = 4.0

    def __init__(self):
        super().__init__()
        self.tiles = {}
        self._guard: Optional[torch.Tensor] = None      # device float: largest |mean| / std any folded LayerNorm has seen
        self._w: "OrderedDict[str, torch.Tensor]" = OrderedDict()
        self._packed: Optional[dict] = None
        self._bufs: Dict[Tuple, dict] = {}
        self.weights_version = 0      # bumped whenever the packed operands are invalidated (captured graphs name them)

    # ---- state_dict with the third-party names
    def state_dict(self, *args, **kwargs):
        return OrderedDict((k, v.detach().clone()) for k, v in self._w.items())

    def load_state_dict(self, sd, strict: bool = True):
        missing = [k for k in self._w if k not in sd]
        unexpected = [k for k in sd if k not in self._w]
        if strict and missing:
            raise RuntimeError(f"missing keys: {missing[:5]}{'...' if len(missing) > 5 else ''}")
        for k in self._w:
            if k in sd:
                if tuple(sd[k].shape) != tuple(self._w[k].shape):
                    raise RuntimeError(f"shape mismatch for {k}: {tuple(sd[k].shape)} vs {tuple(self._w[k].shape)}")
                self._w[k].copy_(sd[k].to(self._w[k].device, torch.float32))
        self._packed = None
        self.weights_version += 1
        return missing, unexpected

    def _apply(self, fn, recurse=True):
        super()._apply(fn, recurse)
        for k in self._w:
            self._w[k] = fn(self._w[k]).float()
        self._packed = None
        self._bufs.clear()
        self.weights_version += 1
        return self

    @property
    def device(self) -> torch.device:
        return next(iter(self._w.values())).device

    def _require_hip(self):
        if self.device.type != "cuda":
            raise L.UltrafndHipError(f"{type(self).__name__} runs on a HIP device only: call .to('cuda') (no CPU fallback)")

    # ---- thin wrappers over the C ABI
    # GEMM tile per Linear ("qkv", "out", "ffn1", "ffn2"): -1 = the library's own choice for the shape.  Set by the
    # trainer when the encoder runs on a compute-unit partition (a launch wants about one tile per CU it may use).
    tiles: Dict[str, int] = {}

    def _gemm(self, A, W, bias, out_bf16=None, out_f32=None, residual=None, act=ACT_NONE, M=None, which=None):
        M = A.shape[0] if M is None else M
        N, K = W.shape
        L.check(L.lib().ufnd_gemm_bf16_ex(A.data_ptr(), W.data_ptr(), L.ptr(bias), L.ptr(residual), L.ptr(out_bf16),
                                          L.ptr(out_f32), M, N, K, A.stride(0), W.stride(0),
                                          residual.stride(0) if residual is not None else 0,
                                          out_bf16.stride(0) if out_bf16 is not None else 0,
                                          out_f32.stride(0) if out_f32 is not None else 0, act, self.tiles.get(which, -1),
                                          L.stream_ptr(A.device)), "ufnd_gemm_bf16_ex")

    def _gemm_ln(self, A, W, bias, out_bf16=None, out_f32=None, residual=None, act=ACT_NONE, a_stats=None, colsum=None,
                 r_stats=None, r_gamma=None, r_beta=None, out_stats=None, eps=1e-5, which=None, residual_bf16=None):
        """ufnd_gemm_bf16_ln: a_stats / r_stats / out_stats are (M, parts, 2) fp32 tensors."""
        M = A.shape[0]
        N, K = W.shape
        ln = L.GemmLn()
        ln.a_stats, ln.colsum = L.ptr(a_stats), L.ptr(colsum)
        ln.r_stats, ln.r_gamma, ln.r_beta = L.ptr(r_stats), L.ptr(r_gamma), L.ptr(r_beta)
        ln.out_stats = L.ptr(out_stats)
        ln.a_parts = a_stats.shape[1] if a_stats is not None else 0
        ln.r_parts = r_stats.shape[1] if r_stats is not None else 0
        ln.a_eps = ln.r_eps = eps
        ln.width = self.hidden
        ln.tile_cfg = self.tiles.get(which, -1)
        if residual_bf16 is not None:
            ln.residual_bf16, ln.ldrb = residual_bf16.data_ptr(), residual_bf16.stride(0)
        if a_stats is not None:      # fold guard: the launch reports the largest |mean| / std among the rows it folds
            ln.guard = self._guard_buf(A.device).data_ptr()
        import ctypes
        L.check(L.lib().ufnd_gemm_bf16_ln(A.data_ptr(), W.data_ptr(), L.ptr(bias), L.ptr(residual), L.ptr(out_bf16),
                                          L.ptr(out_f32), M, N, K, A.stride(0), W.stride(0),
                                          residual.stride(0) if residual is not None else 0,
                                          out_bf16.stride(0) if out_bf16 is not None else 0,
                                          out_f32.stride(0) if out_f32 is not None else 0, act, ctypes.byref(ln),
                                          L.stream_ptr(A.device)), "ufnd_gemm_bf16_ln")

    def _guard_buf(self, dev) -> torch.Tensor:
        """The fold guard's slots (UFND_FOLD_GUARD_SLOTS floats): every folded GEMM launch -- inside captured graphs too -- leaves
        the largest |mean| / std among the rows it folds in them (one atomicMax per workgroup, ufnd_gemm_ln.guard), so every
        row of every batch is looked at by the kernel that folds it; the host reads the slots when it chooses to (check_fold)."""
        if self._guard is None or self._guard.device != dev:
            self._guard = torch.zeros(L.FOLD_GUARD_SLOTS, dtype=torch.float32, device=dev)
        return self._guard

    def fold_ratio(self) -> float:
        """Largest |mean| / std over the rows that went through a folded LayerNorm since the last reset (synchronises)."""
        return 0.0 if self._guard is None else float(self._guard.max().cpu())

    def check_fold(self, reset: bool = True, ratio: Optional[float] = None) -> bool:
        """The fold guard: True (and folding switched off for every later call, buffers rebuilt) when a folded row's
        |mean| / std exceeded FOLD_GUARD_MAX.  Every folded GEMM evaluates the guard on the device (inside captured graphs
        too); this reads it: `strict=True` forwards after each batch (which is then
        repeated unfolded), the trainer after every encoder pass through an asynchronous copy (`ratio=` hands that value in,
        so that nothing synchronises)."""
        r = self.fold_ratio() if ratio is None else float(ratio)
        if reset and self._guard is not None:
            self._guard.zero_()
        if self.fold_ln and r > self.FOLD_GUARD_MAX:
            import warnings
            warnings.warn(f"{type(self).__name__}: a residual-stream row with |mean|/std = {r:.1f} (> {self.FOLD_GUARD_MAX}) went through a "
                          "folded LayerNorm; switching to materialised LayerNorms (fold_ln=False)")
            self.fold_ln = False
            self._bufs.clear()
            return True
        return False

    @staticmethod
    def _fold(W, b, gamma, beta):
        """LayerNorm(x; gamma, beta) W^T + b  ==  rstd (x W'^T - mean colsum) + b'  with
        W' = bf16(W * gamma), colsum = row sums of the ROUNDED W', b' = b + W beta."""
        Wp = _bf16(W * gamma[None, :])
        # (W * beta).sum(1), not W @ beta: packing is one-off setup, but it was the only vendor-BLAS call (rocblas gemv) of the tree
        return Wp, Wp.float().sum(1).contiguous(), (b + (W * beta[None, :]).sum(1)).contiguous()

    def _ln(self, x, ldx, gamma, beta, out_bf16, out_f32, M, H, eps):
        L.check(L.lib().ufnd_layernorm(x.data_ptr(), ldx, gamma.data_ptr(), beta.data_ptr(), L.ptr(out_bf16),
                                       L.ptr(out_f32), M, H, eps, L.stream_ptr(x.device)), "ufnd_layernorm")

    def _qkv_attn(self, A, W, bias, mask_i32, ctx, B, Lq, heads, a_stats=None, colsum=None, eps=1e-5):
        """ufnd_qkv_attention_bf16: fused Q/K/V projection (+ folded LayerNorm of A) + attention, one launch per layer."""
        import ctypes
        ln = None
        if a_stats is not None:
            ln = L.GemmLn()
            ln.a_stats, ln.colsum, ln.a_parts, ln.a_eps, ln.r_eps, ln.width = a_stats.data_ptr(), colsum.data_ptr(), a_stats.shape[1], eps, eps, self.hidden
            ln.guard = self._guard_buf(A.device).data_ptr()
        L.check(L.lib().ufnd_qkv_attention_bf16(A.data_ptr(), W.data_ptr(), L.ptr(bias), L.ptr(mask_i32), ctx.data_ptr(), B, Lq, heads,
                                                A.stride(0), W.stride(0), ctypes.byref(ln) if ln is not None else None,
                                                L.stream_ptr(A.device)), "ufnd_qkv_attention_bf16")

    def _attn(self, qkv, mask_i32, ctx, B, Lq, heads):
        L.check(L.lib().ufnd_attention_bf16(qkv.data_ptr(), L.ptr(mask_i32), ctx.data_ptr(), B, Lq, heads,
                                            L.stream_ptr(qkv.device)), "ufnd_attention_bf16")


# =============================================================================================
class BertTextEncoder(_EncoderBase):
    def __init__(self, layers: int = 12, hidden: int = 768, heads: int = 12, intermediate: int = 3072,
                 vocab_size: int = 30522, max_position: int = 512, type_vocab: int = 2, eps: float = 1e-12,
                 fold_ln: bool = True, residual_dtype: str = "bf16"):
        super().__init__()
        self.fold_ln = fold_ln
        if residual_dtype not in ("fp32", "bf16"):
            raise ValueError(f"residual_dtype={residual_dtype!r}: 'fp32' or 'bf16'")
        self.residual_dtype = residual_dtype
        # 128-token samples: Q/K/V projection + attention of a layer as ONE launch (ufnd_qkv_attention_bf16); bit-identical
        # to the two-launch form, which every other length uses
        self.fuse_qkv_attention = True
        if hidden != heads * 64:
            raise ValueError("head_dim must be 64 (hidden == heads * 64)")
        self.layers, self.hidden, self.heads, self.inter, self.vocab, self.eps = layers, hidden, heads, intermediate, vocab_size, eps
        self.max_position = max_position
        # output.dense as a 2-way split-K GEMM when the token count gives a full wave of 128x192 tiles
        w = self._w
        g = torch.Generator().manual_seed(0)

        def init(shape, std=0.02):
            return torch.randn(shape, generator=g) * std
        w["embeddings.word_embeddings.weight"] = init((vocab_size, hidden))
        w["embeddings.position_embeddings.weight"] = init((max_position, hidden))
        w["embeddings.token_type_embeddings.weight"] = init((type_vocab, hidden))
        w["embeddings.LayerNorm.weight"] = torch.ones(hidden)
        w["embeddings.LayerNorm.bias"] = torch.zeros(hidden)
        for i in range(layers):
            P = f"encoder.layer.{i}."
            for n in ("query", "key", "value"):
                w[P + f"attention.self.{n}.weight"] = init((hidden, hidden))
                w[P + f"attention.self.{n}.bias"] = torch.zeros(hidden)
            w[P + "attention.output.dense.weight"] = init((hidden, hidden))
            w[P + "attention.output.dense.bias"] = torch.zeros(hidden)
            w[P + "attention.output.LayerNorm.weight"] = torch.ones(hidden)
            w[P + "attention.output.LayerNorm.bias"] = torch.zeros(hidden)
            w[P + "intermediate.dense.weight"] = init((intermediate, hidden))
            w[P + "intermediate.dense.bias"] = torch.zeros(intermediate)
            w[P + "output.dense.weight"] = init((hidden, intermediate))
            w[P + "output.dense.bias"] = torch.zeros(hidden)
            w[P + "output.LayerNorm.weight"] = torch.ones(hidden)
            w[P + "output.LayerNorm.bias"] = torch.zeros(hidden)

    def _pack(self):
        if self._packed is None:
            w, layers = self._w, []
            for i in range(self.layers):
                P = f"encoder.layer.{i}."
                layers.append({
                    "wqkv": _bf16(torch.cat([w[P + f"attention.self.{n}.weight"] for n in ("query", "key", "value")], 0)),
                    "bqkv": torch.cat([w[P + f"attention.self.{n}.bias"] for n in ("query", "key", "value")], 0).contiguous(),
                    "wo": _bf16(w[P + "attention.output.dense.weight"]), "bo": w[P + "attention.output.dense.bias"],
                    "g1": w[P + "attention.output.LayerNorm.weight"], "b1": w[P + "attention.output.LayerNorm.bias"],
                    "w1": _bf16(w[P + "intermediate.dense.weight"]), "bi": w[P + "intermediate.dense.bias"],
                    "w2": _bf16(w[P + "output.dense.weight"]), "b2": w[P + "output.dense.bias"],
                    "g2": w[P + "output.LayerNorm.weight"], "b2n": w[P + "output.LayerNorm.bias"]})
                ly = layers[-1]
                # folded forms: intermediate.dense consumes LayerNorm_1; the NEXT layer's Q/K/V consume LayerNorm_2
                ly["w1f"], ly["cs1"], ly["bif"] = self._fold(w[P + "intermediate.dense.weight"], ly["bi"], ly["g1"], ly["b1"])
                if i > 0:
                    prev = layers[-2]
                    wq = torch.cat([w[P + f"attention.self.{n}.weight"] for n in ("query", "key", "value")], 0)
                    ly["wqkvf"], ly["csqkv"], ly["bqkvf"] = self._fold(wq, ly["bqkv"], prev["g2"], prev["b2n"])
            self._packed = {"layers": layers}
        return self._packed

    def _workbufs(self, B: int, Lq: int) -> dict:
        key = (B, Lq)
        if key not in self._bufs:
            dev, M, H = self.device, B * Lq, self.hidden
            bf, f32 = dict(dtype=torch.bfloat16, device=dev), dict(dtype=torch.float32, device=dev)
            self._bufs[key] = {"xb": torch.empty(M, H, **bf), "xf": torch.empty(M, H, **f32), "y": torch.empty(M, H, **f32),
                               "x1b": torch.empty(M, H, **bf), "x1f": torch.empty(M, H, **f32),
                               "qkv": torch.empty(M, 3 * H, **bf), "ctx": torch.empty(M, H, **bf),
                               "h": torch.empty(M, self.inter, **bf), "feat": torch.empty(B, H, **f32)}
            p1 = L.lib().ufnd_gemm_bf16_stat_parts(M, H, H)
            p2 = L.lib().ufnd_gemm_bf16_stat_parts(M, H, self.inter)
            if self.fold_ln and p1 > 0 and p1 == p2 and p1 % 2 == 0:
                # one statistics buffer per folded LayerNorm of the pass (st[2i]: layer i's attention half, st[2i+1]: its
                # feed-forward half); every buffer is read by a folded GEMM, which reports its rows' |mean| / std (fold guard)
                self._bufs[key].update({"y2": torch.empty(M, H, **f32), "y1b": torch.empty(M, H, **bf), "y2b": torch.empty(M, H, **bf),
                                        "st": torch.zeros(2 * self.layers, M, p1, 2, **f32)})
                self._guard_buf(dev)
        return self._bufs[key]

    @torch.no_grad()
    def last_hidden_state(self, input_ids: torch.Tensor, attention_mask: torch.Tensor, n_layers: Optional[int] = None) -> torch.Tensor:
        """BertModel(input_ids, attention_mask).last_hidden_state -> (B, L, H) fp32 (a view of an
        internal buffer, valid until the next call with the same shape).  n_layers: stop after that many layers (the
        hidden state BertModel reports as hidden_states[n_layers]; per-layer localisation in the parity tests)."""
        self._require_hip()
        dev = self.device
        B, Lq = input_ids.shape
        if Lq > self.max_position:
            raise RuntimeError(f"sequence length {Lq} exceeds max_position_embeddings {self.max_position}")
        ids = input_ids.to(dev, torch.int64).contiguous()
        mask = attention_mask.to(dev, torch.int32).contiguous()
        self._mask_i32 = mask
        p, b, w = self._pack(), self._workbufs(B, Lq), self._w
        M, H = B * Lq, self.hidden
        L.check(L.lib().ufnd_bert_embed(ids.data_ptr(), w["embeddings.word_embeddings.weight"].data_ptr(),
                                        w["embeddings.position_embeddings.weight"].data_ptr(),
                                        w["embeddings.token_type_embeddings.weight"].data_ptr(),
                                        w["embeddings.LayerNorm.weight"].data_ptr(), w["embeddings.LayerNorm.bias"].data_ptr(),
                                        b["xb"].data_ptr(), None if ("st" in b and self.residual_dtype == "bf16") else b["xf"].data_ptr(),
                                        B, Lq, H, self.vocab, self.eps, L.stream_ptr(dev)), "ufnd_bert_embed")      # (bf16 stream, folded: nothing reads the fp32 copy)
        fused = (B, Lq, mask) if (self.fuse_qkv_attention and Lq == 128 and self.heads % 2 == 0) else None
        if n_layers is not None:
            p = {"layers": p["layers"][:max(1, int(n_layers))]}
        self._layers(p, b, M, lambda qkv, ctx: self._attn(qkv, mask, ctx, B, Lq, self.heads), fused)
        return b["xf"].view(B, Lq, H)

    def _layers(self, p, b, M, attn, fused=None) -> None:
        """The encoder layers over the first M rows of the work buffers; attn(qkv, ctx) runs the attention (padded
        batch or packed sequences).  Leaves last_hidden_state in b["xf"] (and its bf16 rounding in b["xb"])."""
        H = self.hidden
        v = {k: (x[:M] if torch.is_tensor(x) and x.dim() >= 2 and x.shape[0] >= M and k not in ("feat", "st") else x) for k, x in b.items()}
        if "st" in b:
            return self._layers_folded(p, v, M, attn, fused)
        for ly in p["layers"]:
            if fused is not None:
                self._qkv_attn(v["xb"], ly["wqkv"], ly["bqkv"], fused[2], v["ctx"], fused[0], fused[1], self.heads)
            else:
                self._gemm(v["xb"], ly["wqkv"], ly["bqkv"], out_bf16=v["qkv"], which="qkv")
                attn(v["qkv"], v["ctx"])
            self._gemm(v["ctx"], ly["wo"], ly["bo"], out_f32=v["y"], residual=v["xf"], which="out")
            self._ln(v["y"], H, ly["g1"], ly["b1"], v["x1b"], v["x1f"], M, H, self.eps)
            self._gemm(v["x1b"], ly["w1"], ly["bi"], out_bf16=v["h"], act=ACT_GELU, which="ffn1")
            self._gemm(v["h"], ly["w2"], ly["b2"], out_f32=v["y"], residual=v["x1f"], which="ffn2")
            self._ln(v["y"], H, ly["g2"], ly["b2n"], v["xb"], v["xf"], M, H, self.eps)

    def _layers_folded(self, p, b, M, attn, fused=None) -> None:
        """The layers without a LayerNorm kernel between Linears (module docstring).  y1 / y2 are the
        PRE-LayerNorm sums of the attention and the feed-forward halves (fp32 + bf16 + row statistics)."""
        H, eps = self.hidden, self.eps
        y1, y2 = b["y"], b["y2"]
        st = b["st"]
        prev = None
        for i, ly in enumerate(p["layers"]):
            st1, st2 = st[2 * i][:M], st[2 * i + 1][:M]
            stp = st[2 * i - 1][:M] if i > 0 else None           # the previous layer's feed-forward half
            if fused is not None:
                if prev is None:
                    self._qkv_attn(b["xb"], ly["wqkv"], ly["bqkv"], fused[2], b["ctx"], fused[0], fused[1], self.heads)
                else:
                    self._qkv_attn(b["y2b"], ly["wqkvf"], ly["bqkvf"], fused[2], b["ctx"], fused[0], fused[1], self.heads,
                                   a_stats=stp, colsum=ly["csqkv"], eps=eps)
            elif prev is None:      # layer 0 consumes the embeddings' own (materialised) LayerNorm
                self._gemm(b["xb"], ly["wqkv"], ly["bqkv"], out_bf16=b["qkv"], which="qkv")
            else:
                self._gemm_ln(b["y2b"], ly["wqkvf"], ly["bqkvf"], out_bf16=b["qkv"], a_stats=stp, colsum=ly["csqkv"], eps=eps, which="qkv")
            if fused is None:
                attn(b["qkv"], b["ctx"])
            rb = self.residual_dtype == "bf16"
            last = i == len(p["layers"]) - 1
            if prev is None:
                self._gemm_ln(b["ctx"], ly["wo"], ly["bo"], out_f32=None if rb else y1, out_bf16=b["y1b"], residual=None if rb else b["xf"],
                              residual_bf16=b["xb"] if rb else None, out_stats=st1, eps=eps, which="out")
            else:
                self._gemm_ln(b["ctx"], ly["wo"], ly["bo"], out_f32=None if rb else y1, out_bf16=b["y1b"], residual=None if rb else y2,
                              residual_bf16=b["y2b"] if rb else None, r_stats=stp, r_gamma=prev["g2"], r_beta=prev["b2n"], out_stats=st1,
                              eps=eps, which="out")
            self._gemm_ln(b["y1b"], ly["w1f"], ly["bif"], out_bf16=b["h"], act=ACT_GELU, a_stats=st1, colsum=ly["cs1"], eps=eps, which="ffn1")
            self._gemm_ln(b["h"], ly["w2"], ly["b2"], out_f32=y2 if (last or not rb) else None, out_bf16=b["y2b"], residual=None if rb else y1,
                          residual_bf16=b["y1b"] if rb else None, r_stats=st1, r_gamma=ly["g1"], r_beta=ly["b1"], out_stats=st2, eps=eps,
                          which="ffn2")      # (the last layer's fp32 sums feed the final, materialised LayerNorm)
            prev = ly
        self._ln(y2, H, prev["g2"], prev["b2n"], None, b["xf"], M, H, eps)           # last_hidden_state is materialised once (fp32 only: no GEMM reads it)

    @torch.no_grad()
    def forward(self, input_ids: torch.Tensor, attention_mask: torch.Tensor, unpad: bool = False, strict: bool = False) -> torch.Tensor:
        """Batched BERTContextEncoder.encode: (B,L) ids/mask -> (B,768) L2-normalised features.
        strict=True is the offline feature builder's mode: fp32 residual stream (whatever `residual_dtype` says), the fold guard checked
        after the pass (one host sync) and, if it tripped, the batch repeated with materialised LayerNorms; not for hipGraph capture.
        unpad=True runs the encoder over the kept tokens only (packed rows, per-sequence attention): the padded
        positions never reach the pooling (text_blocks.py:82-86), so no returned value changes -- bit-identical for
        prefix masks -- while the work drops with the padding fraction.  Shapes vary per batch: not for hipGraph capture."""
        if strict:
            # the offline feature builder (encode_fields; caches for the reference-style training mode): the more faithful fp32
            # residual stream whatever the in-step default is -- the bf16 stream's speed only matters inside the train step
            rd, self.residual_dtype = self.residual_dtype, "fp32"
            try:
                if self.fold_ln:
                    out = self.forward(input_ids, attention_mask, unpad=unpad)
                    if not self.check_fold():
                        return out
                return self.forward(input_ids, attention_mask, unpad=unpad)
            finally:
                self.residual_dtype = rd
        if unpad:
            return self._forward_packed(input_ids, attention_mask)
        hid = self.last_hidden_state(input_ids, attention_mask)
        B, Lq, H = hid.shape
        feat = self._workbufs(B, Lq)["feat"]
        L.check(L.lib().ufnd_masked_meanpool_l2(hid.data_ptr(), self._mask_i32.data_ptr(), feat.data_ptr(), B, Lq, H,
                                                L.stream_ptr(self.device)), "ufnd_masked_meanpool_l2")
        return feat

    def _forward_packed(self, input_ids: torch.Tensor, attention_mask: torch.Tensor) -> torch.Tensor:
        self._require_hip()
        dev = self.device
        B, Lq = input_ids.shape
        if Lq > self.max_position:
            raise RuntimeError(f"sequence length {Lq} exceeds max_position_embeddings {self.max_position}")
        ids = input_ids.to(dev, torch.int64).reshape(-1)
        mask = attention_mask.to(dev, torch.int32).contiguous()
        keep = torch.nonzero(mask.reshape(-1), as_tuple=False).flatten()          # (host sync: the packed row count)
        T = int(keep.numel())
        p, b, w = self._pack(), self._workbufs(B, Lq), self._w
        H, s = self.hidden, L.stream_ptr(dev)
        feat = b["feat"]
        if T == 0:
            return feat.zero_()
        ids_p = ids[keep].contiguous()
        pos = (keep % Lq).to(torch.int32).contiguous()
        cu = torch.zeros(B + 1, dtype=torch.int32, device=dev)
        cu[1:] = torch.cumsum(mask.sum(1, dtype=torch.int32), 0)
        L.check(L.lib().ufnd_bert_embed_packed(ids_p.data_ptr(), pos.data_ptr(), w["embeddings.word_embeddings.weight"].data_ptr(),
                                               w["embeddings.position_embeddings.weight"].data_ptr(),
                                               w["embeddings.token_type_embeddings.weight"].data_ptr(),
                                               w["embeddings.LayerNorm.weight"].data_ptr(), w["embeddings.LayerNorm.bias"].data_ptr(),
                                               b["xb"].data_ptr(), b["xf"].data_ptr(), T, self.max_position, H, self.vocab, self.eps, s),
                "ufnd_bert_embed_packed")

        def attn(qkv, ctx):
            L.check(L.lib().ufnd_attention_bf16_varlen(qkv.data_ptr(), cu.data_ptr(), ctx.data_ptr(), B, Lq, self.heads, s),
                    "ufnd_attention_bf16_varlen")
        self._layers(p, b, T, attn)
        L.check(L.lib().ufnd_meanpool_l2_packed(b["xf"].data_ptr(), cu.data_ptr(), pos.data_ptr(), feat.data_ptr(), B, H, s),
                "ufnd_meanpool_l2_packed")
        return feat

    encode_batch = forward

    @torch.no_grad()
    def encode_fields(self, input_ids: torch.Tensor, attention_mask: torch.Tensor, part_valid: torch.Tensor,
                      chunk: int = 1024, unpad: bool = True) -> torch.Tensor:
        """Batched BERTContextEncoder.encode_fields (text_blocks.py:108-128).
        input_ids / attention_mask (N, M, L): the tokenised parts of N records in the reference's
        order [title, ocr, up to 10 comments] (M <= 12); part_valid (N, M): 1 where the part exists
        (the reference skips empty strings).  Every existing part goes through the encoder in one
        batched pass (the reference runs <= 12 batch-1 forwards per record), then the per-record mean
        of the part vectors is L2-normalised.  Returns (N, 768); zero rows for records without parts."""
        self._require_hip()
        dev = self.device
        N, Mx, Lq = input_ids.shape
        valid = part_valid.to(dev, torch.int32).contiguous()
        ids = input_ids.to(dev, torch.int64).reshape(N * Mx, Lq)
        mask = attention_mask.to(dev, torch.int32).reshape(N * Mx, Lq)
        rows = torch.nonzero(valid.reshape(-1), as_tuple=False).flatten()
        parts = torch.zeros(N * Mx, self.hidden, dtype=torch.float32, device=dev)
        for s0 in range(0, rows.numel(), chunk):
            r = rows[s0:s0 + chunk]
            parts[r] = self.forward(ids[r], mask[r], unpad=unpad, strict=True)      # (texts are short against max_length: skip the padding)
        out = torch.empty(N, self.hidden, dtype=torch.float32, device=dev)
        L.check(L.lib().ufnd_field_mean_l2(parts.data_ptr(), valid.data_ptr(), out.data_ptr(), N, Mx, self.hidden,
                                           L.stream_ptr(dev)), "ufnd_field_mean_l2")
        return out


# =============================================================================================
class ClipVisualEncoder(_EncoderBase):
    def __init__(self, layers: int = 12, hidden: int = 768, heads: int = 12, intermediate: int = 3072, patch: int = 32,
                 image: int = 224, projection_dim: int = 512, eps: float = 1e-5, fold_ln: bool = True, residual_dtype: str = "bf16"):
        super().__init__()
        self.fold_ln = fold_ln
        if residual_dtype not in ("fp32", "bf16"):
            raise ValueError(f"residual_dtype={residual_dtype!r}: 'fp32' or 'bf16'")
        self.residual_dtype = residual_dtype
        if hidden != heads * 64:
            raise ValueError("head_dim must be 64 (hidden == heads * 64)")
        self.layers, self.hidden, self.heads, self.inter = layers, hidden, heads, intermediate
        self.patch, self.image, self.proj, self.eps = patch, image, projection_dim, eps
        self.n_patches = (image // patch) ** 2
        w, V = self._w, "vision_model."
        g = torch.Generator().manual_seed(0)

        def init(shape, std=0.02):
            return torch.randn(shape, generator=g) * std
        w[V + "embeddings.class_embedding"] = init((hidden,))
        w[V + "embeddings.patch_embedding.weight"] = init((hidden, 3, patch, patch))
        w[V + "embeddings.position_embedding.weight"] = init((self.n_patches + 1, hidden))
        w[V + "pre_layrnorm.weight"], w[V + "pre_layrnorm.bias"] = torch.ones(hidden), torch.zeros(hidden)
        for i in range(layers):
            P = V + f"encoder.layers.{i}."
            for n in ("k_proj", "v_proj", "q_proj", "out_proj"):
                w[P + f"self_attn.{n}.weight"] = init((hidden, hidden))
                w[P + f"self_attn.{n}.bias"] = torch.zeros(hidden)
            w[P + "layer_norm1.weight"], w[P + "layer_norm1.bias"] = torch.ones(hidden), torch.zeros(hidden)
            w[P + "mlp.fc1.weight"], w[P + "mlp.fc1.bias"] = init((intermediate, hidden)), torch.zeros(intermediate)
            w[P + "mlp.fc2.weight"], w[P + "mlp.fc2.bias"] = init((hidden, intermediate)), torch.zeros(hidden)
            w[P + "layer_norm2.weight"], w[P + "layer_norm2.bias"] = torch.ones(hidden), torch.zeros(hidden)
        w[V + "post_layernorm.weight"], w[V + "post_layernorm.bias"] = torch.ones(hidden), torch.zeros(hidden)
        w["visual_projection.weight"] = init((projection_dim, hidden))

    def _pack(self):
        if self._packed is None:
            w, V, layers = self._w, "vision_model.", []
            for i in range(self.layers):
                P = V + f"encoder.layers.{i}."
                layers.append({
                    "wqkv": _bf16(torch.cat([w[P + f"self_attn.{n}.weight"] for n in ("q_proj", "k_proj", "v_proj")], 0)),
                    "bqkv": torch.cat([w[P + f"self_attn.{n}.bias"] for n in ("q_proj", "k_proj", "v_proj")], 0).contiguous(),
                    "wo": _bf16(w[P + "self_attn.out_proj.weight"]), "bo": w[P + "self_attn.out_proj.bias"],
                    "g1": w[P + "layer_norm1.weight"], "b1": w[P + "layer_norm1.bias"],
                    "w1": _bf16(w[P + "mlp.fc1.weight"]), "bi": w[P + "mlp.fc1.bias"],
                    "w2": _bf16(w[P + "mlp.fc2.weight"]), "b2": w[P + "mlp.fc2.bias"],
                    "g2": w[P + "layer_norm2.weight"], "b2n": w[P + "layer_norm2.bias"]})
                ly = layers[-1]
                wq = torch.cat([w[P + f"self_attn.{n}.weight"] for n in ("q_proj", "k_proj", "v_proj")], 0)
                ly["wqkvf"], ly["csqkv"], ly["bqkvf"] = self._fold(wq, ly["bqkv"], ly["g1"], ly["b1"])
                ly["w1f"], ly["cs1"], ly["bif"] = self._fold(w[P + "mlp.fc1.weight"], ly["bi"], ly["g2"], ly["b2n"])
            self._packed = {"layers": layers,
                            "wpatch": _bf16(w[V + "embeddings.patch_embedding.weight"].reshape(self.hidden, -1)),
                            "wproj": _bf16(w["visual_projection.weight"])}
        return self._packed

    def _workbufs(self, B: int, Fr: int) -> dict:
        key = (B, Fr)
        if key not in self._bufs:
            dev, H, N = self.device, self.hidden, B * Fr
            T = self.n_patches + 1
            M = N * T
            bf, f32 = dict(dtype=torch.bfloat16, device=dev), dict(dtype=torch.float32, device=dev)
            self._bufs[key] = {"patches": torch.empty(N * self.n_patches, 3 * self.patch ** 2, **bf),
                               "pe": torch.empty(N * self.n_patches, H, **f32), "xf": torch.empty(M, H, **f32),
                               "hb": torch.empty(M, H, **bf), "qkv": torch.empty(M, 3 * H, **bf),
                               "ctx": torch.empty(M, H, **bf), "m": torch.empty(M, self.inter, **bf),
                               "pooled": torch.empty(N, H, **bf), "e": torch.empty(N, self.proj, **f32),
                               "feat": torch.empty(B, self.proj, **f32)}
            p1 = L.lib().ufnd_gemm_bf16_stat_parts(M, H, H)
            p2 = L.lib().ufnd_gemm_bf16_stat_parts(M, H, self.inter)
            if self.fold_ln and p1 > 0 and p1 == p2 and p1 % 2 == 0:
                # st0: the assembled embeddings' statistics; st[2i] / st[2i+1]: layer i's attention / feed-forward half
                self._bufs[key].update({"st0": torch.zeros(M, 2, 2, **f32), "st": torch.zeros(2 * self.layers, M, p1, 2, **f32)})
                self._guard_buf(dev)
        return self._bufs[key]

    @torch.no_grad()
    def image_embeds(self, frames: torch.Tensor) -> torch.Tensor:
        """frames (N,3,S,S) fp32 -> un-normalised projected embeddings (N, 512)
        (CLIPVisionModelWithProjection.image_embeds)."""
        return self._run(frames[:, None])[0]

    @torch.no_grad()
    def hidden_state(self, frames: torch.Tensor, n_layers: Optional[int] = None) -> torch.Tensor:
        """The residual stream after `n_layers` layers (all by default), (N, 50, H) fp32: CLIPVisionModel's hidden_states[n_layers]
        (per-layer localisation in the parity tests)."""
        if frames.dim() == 4:
            frames = frames[:, None]
        _, b = self._run(frames, n_layers)
        N = frames.shape[0] * frames.shape[1]
        return b["xf"].view(N, self.n_patches + 1, self.hidden)

    def _run(self, frames5: torch.Tensor, n_layers: Optional[int] = None):
        self._require_hip()
        dev = self.device
        B, Fr = frames5.shape[:2]
        if tuple(frames5.shape[2:]) != (3, self.image, self.image):
            raise RuntimeError(f"frames: expected (B,F,3,{self.image},{self.image}), got {tuple(frames5.shape)}")
        fr = L.f32c(frames5.to(dev)).view(B * Fr, 3, self.image, self.image)
        p, b, w, V = self._pack(), self._workbufs(B, Fr), self._w, "vision_model."
        if n_layers is not None:
            p = dict(p, layers=p["layers"][:max(1, int(n_layers))])
        N, H, T = B * Fr, self.hidden, self.n_patches + 1
        M = N * T
        s = L.stream_ptr(dev)
        L.check(L.lib().ufnd_vit_patchify(fr.data_ptr(), b["patches"].data_ptr(), N, self.image, self.patch, s), "ufnd_vit_patchify")
        self._gemm(b["patches"], p["wpatch"], None, out_f32=b["pe"])
        L.check(L.lib().ufnd_vit_assemble(b["pe"].data_ptr(), w[V + "embeddings.class_embedding"].data_ptr(),
                                          w[V + "embeddings.position_embedding.weight"].data_ptr(),
                                          w[V + "pre_layrnorm.weight"].data_ptr(), w[V + "pre_layrnorm.bias"].data_ptr(),
                                          None if ("st0" in b and self.residual_dtype == "bf16") else b["xf"].data_ptr(),      # (bf16 stream: hb is the stream)
                                          L.ptr(b["hb"]) if "st0" in b else None, L.ptr(b.get("st0")),
                                          N, self.n_patches, H, self.eps, s), "ufnd_vit_assemble")
        if "st0" in b:
            # pre-LN blocks without LayerNorm kernels: hb is the bf16 rounding of the residual stream xf,
            # st* the row statistics its LayerNorms need (module docstring)
            eps, stA, st = self.eps, b["st0"], b["st"]
            for i, ly in enumerate(p["layers"]):
                st1, st2 = st[2 * i], st[2 * i + 1]
                self._gemm_ln(b["hb"], ly["wqkvf"], ly["bqkvf"], out_bf16=b["qkv"], a_stats=stA, colsum=ly["csqkv"], eps=eps, which="qkv")
                self._attn(b["qkv"], None, b["ctx"], N, T, self.heads)
                rb = self.residual_dtype == "bf16"       # the stream is hb itself, updated in place (a tile reads exactly what it rewrites)
                last = i == len(p["layers"]) - 1
                self._gemm_ln(b["ctx"], ly["wo"], ly["bo"], out_f32=None if rb else b["xf"], out_bf16=b["hb"], residual=None if rb else b["xf"],
                              residual_bf16=b["hb"] if rb else None, out_stats=st1, eps=eps, which="out")
                self._gemm_ln(b["hb"], ly["w1f"], ly["bif"], out_bf16=b["m"], act=ACT_QUICK_GELU, a_stats=st1, colsum=ly["cs1"], eps=eps, which="ffn1")
                self._gemm_ln(b["m"], ly["w2"], ly["b2"], out_f32=b["xf"] if (last or not rb) else None, out_bf16=b["hb"],
                              residual=None if rb else b["xf"], residual_bf16=b["hb"] if rb else None, out_stats=st2, eps=eps, which="ffn2")
                stA = st2
            self._ln(b["xf"], T * H, w[V + "post_layernorm.weight"], w[V + "post_layernorm.bias"], b["pooled"], None, N, H, self.eps)
            self._gemm(b["pooled"], p["wproj"], None, out_f32=b["e"])
            return b["e"], b
        for ly in p["layers"]:
            self._ln(b["xf"], H, ly["g1"], ly["b1"], b["hb"], None, M, H, self.eps)
            self._gemm(b["hb"], ly["wqkv"], ly["bqkv"], out_bf16=b["qkv"], which="qkv")
            self._attn(b["qkv"], None, b["ctx"], N, T, self.heads)
            self._gemm(b["ctx"], ly["wo"], ly["bo"], out_f32=b["xf"], residual=b["xf"], which="out")
            self._ln(b["xf"], H, ly["g2"], ly["b2n"], b["hb"], None, M, H, self.eps)
            self._gemm(b["hb"], ly["w1"], ly["bi"], out_bf16=b["m"], act=ACT_QUICK_GELU, which="ffn1")
            self._gemm(b["m"], ly["w2"], ly["b2"], out_f32=b["xf"], residual=b["xf"], which="ffn2")
        # post-LN on the CLS rows (row stride T*H), bias-free projection
        self._ln(b["xf"], T * H, w[V + "post_layernorm.weight"], w[V + "post_layernorm.bias"], b["pooled"], None, N, H, self.eps)
        self._gemm(b["pooled"], p["wproj"], None, out_f32=b["e"])
        return b["e"], b

    @torch.no_grad()
    def forward(self, frames: torch.Tensor, strict: bool = False) -> torch.Tensor:
        """frames (B,F,3,224,224) or (B,3,224,224) -> (B,512) features.  strict: as BertTextEncoder.forward."""
        if frames.dim() == 4:
            frames = frames[:, None]
        if strict:      # (fp32 residual stream for offline feature building: see BertTextEncoder.forward)
            rd, self.residual_dtype = self.residual_dtype, "fp32"
            try:
                if self.fold_ln:
                    out = self.forward(frames)
                    if not self.check_fold():
                        return out
                return self.forward(frames)
            finally:
                self.residual_dtype = rd
        B, Fr = frames.shape[:2]
        e, b = self._run(frames)
        L.check(L.lib().ufnd_l2norm_frames(e.data_ptr(), b["feat"].data_ptr(), B, Fr, self.proj, L.stream_ptr(self.device)),
                "ufnd_l2norm_frames")
        return b["feat"]
