"""TEST INFRASTRUCTURE (see oracle/__init__.py) -- Tier-B encoder oracle.

torch-fp32 CPU restatement of the two frozen, forward-only encoders that feed
the fusion step:

  text   src/core_blocks/text_blocks.py:63-106 -- `self.model(**enc).last_hidden_state`
         (:79, third-party BertModel) -> masked mean-pool (:82-86) -> L2-norm (:100);
         field averaging :108-128.
  visual no neural arithmetic in the reference (visual_blocks.py is classical CV);
         the geometry pointer is configs/model_configs/semantic.yaml:2
         (CLIP ViT-B/32 @224) -> pooled 768 -> bias-free proj 512 -> L2-norm
         (SURVEY.md 8 row a11).

The transformer arithmetic is third-party (`transformers`, requirements.txt:7,
unpinned; 5.15.0 installed): embeddings + LayerNorm(eps 1e-12), softmax(QK^T d^-1/2
+ mask) V, post-LN residual blocks, exact-erf GELU (models/bert/modeling_bert.py);
CLIP vision: patch conv, CLS, pre/post LN(eps 1e-5), pre-LN blocks, quick_gelu,
bias-free visual_projection (models/clip/modeling_clip.py).  State-dict keys are
the third-party names so real checkpoints would load unchanged.  PARITY
UNPINNED by the reference's own tests; pinned only against the local third-party
classes with seeded random weights (tests/golden/make_golden.py).
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------
# geometry + seeded weights
# --------------------------------------------------------------------------
def bert_shapes(layers=12, hidden=768, inter=3072, vocab=30522, max_pos=512, type_vocab=2):
    s: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    s["embeddings.word_embeddings.weight"] = (vocab, hidden)
    s["embeddings.position_embeddings.weight"] = (max_pos, hidden)
    s["embeddings.token_type_embeddings.weight"] = (type_vocab, hidden)
    s["embeddings.LayerNorm.weight"] = (hidden,)
    s["embeddings.LayerNorm.bias"] = (hidden,)
    for i in range(layers):
        L = f"encoder.layer.{i}."
        for n in ("query", "key", "value"):
            s[L + f"attention.self.{n}.weight"] = (hidden, hidden)
            s[L + f"attention.self.{n}.bias"] = (hidden,)
        s[L + "attention.output.dense.weight"] = (hidden, hidden)
        s[L + "attention.output.dense.bias"] = (hidden,)
        s[L + "attention.output.LayerNorm.weight"] = (hidden,)
        s[L + "attention.output.LayerNorm.bias"] = (hidden,)
        s[L + "intermediate.dense.weight"] = (inter, hidden)
        s[L + "intermediate.dense.bias"] = (inter,)
        s[L + "output.dense.weight"] = (hidden, inter)
        s[L + "output.dense.bias"] = (hidden,)
        s[L + "output.LayerNorm.weight"] = (hidden,)
        s[L + "output.LayerNorm.bias"] = (hidden,)
    return s


def vit_shapes(layers=12, hidden=768, inter=3072, patch=32, image=224, proj=512):
    n_tok = (image // patch) ** 2 + 1
    s: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    V = "vision_model."
    s[V + "embeddings.class_embedding"] = (hidden,)
    s[V + "embeddings.patch_embedding.weight"] = (hidden, 3, patch, patch)
    s[V + "embeddings.position_embedding.weight"] = (n_tok, hidden)
    s[V + "pre_layrnorm.weight"] = (hidden,)
    s[V + "pre_layrnorm.bias"] = (hidden,)
    for i in range(layers):
        L = V + f"encoder.layers.{i}."
        for n in ("k_proj", "v_proj", "q_proj", "out_proj"):
            s[L + f"self_attn.{n}.weight"] = (hidden, hidden)
            s[L + f"self_attn.{n}.bias"] = (hidden,)
        s[L + "layer_norm1.weight"] = (hidden,)
        s[L + "layer_norm1.bias"] = (hidden,)
        s[L + "mlp.fc1.weight"] = (inter, hidden)
        s[L + "mlp.fc1.bias"] = (inter,)
        s[L + "mlp.fc2.weight"] = (hidden, inter)
        s[L + "mlp.fc2.bias"] = (hidden,)
        s[L + "layer_norm2.weight"] = (hidden,)
        s[L + "layer_norm2.bias"] = (hidden,)
    s[V + "post_layernorm.weight"] = (hidden,)
    s[V + "post_layernorm.bias"] = (hidden,)
    s["visual_projection.weight"] = (proj, hidden)
    return s


def seeded_weights(shapes, seed: int) -> "OrderedDict[str, torch.Tensor]":
    """Random-init weights of the right architecture (no pretrained weights offline).
    std 0.02 like both models' initializer_range; LayerNorm weight ~1, biases small
    but non-zero so every term is exercised."""
    g = torch.Generator().manual_seed(seed)
    out: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for k, shp in shapes.items():
        low = k.lower()
        if ("layernorm" in low or "layer_norm" in low or "layrnorm" in low) and k.endswith("weight"):
            out[k] = 1.0 + 0.1 * torch.randn(shp, generator=g)
        elif k.endswith("bias"):
            out[k] = 0.02 * torch.randn(shp, generator=g)
        elif len(shp) >= 2:
            fan_in = 1
            for d in shp[1:]:
                fan_in *= d
            # larger than 0.02 so attention logits / activations are not degenerate
            out[k] = torch.randn(shp, generator=g) * (1.0 / math.sqrt(fan_in)) if "embedding" not in low \
                else torch.randn(shp, generator=g) * 0.05
        else:
            out[k] = 0.05 * torch.randn(shp, generator=g)
    return out


def synthetic_tokens(seed: int, B: int, L: int, vocab: int = 30522, min_len: int = 16):
    """ids ~ U[0,vocab) with 101 ([CLS]) in column 0; per-row valid
    length ~ U{min_len..L}, remainder masked."""
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(0, vocab, (B, L), generator=g)
    ids[:, 0] = min(101, vocab - 1)
    lens = torch.randint(min(min_len, L), L + 1, (B,), generator=g)
    mask = (torch.arange(L)[None, :] < lens[:, None]).long()
    return ids, mask


def synthetic_frames(seed: int, B: int, F_: int = 1, image: int = 224):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(B, F_, 3, image, image, generator=g)


# --------------------------------------------------------------------------
# BERT encoder (post-LN)
# --------------------------------------------------------------------------
def _mha(x, wq, bq, wk, bk, wv, bv, heads: int, add_mask: Optional[torch.Tensor]):
    B, L, H = x.shape
    d = H // heads
    q = F.linear(x, wq, bq).view(B, L, heads, d).transpose(1, 2)
    k = F.linear(x, wk, bk).view(B, L, heads, d).transpose(1, 2)
    v = F.linear(x, wv, bv).view(B, L, heads, d).transpose(1, 2)
    s = (q @ k.transpose(-1, -2)) * (d ** -0.5)
    if add_mask is not None:
        s = s + add_mask
    p = torch.softmax(s, dim=-1)
    return (p @ v).transpose(1, 2).reshape(B, L, H)


def bert_last_hidden_state(w: Dict[str, torch.Tensor], input_ids, attention_mask,
                           heads: int = 12, eps: float = 1e-12, collect: Optional[dict] = None) -> torch.Tensor:
    """BertModel(...).last_hidden_state -- text_blocks.py:79.  collect (optional dict): filled with the hidden states after
    each layer, {1: ..., 2: ...} (BertModel's output_hidden_states[1:]), for per-layer localisation in the tests."""
    B, L = input_ids.shape
    H = w["embeddings.word_embeddings.weight"].shape[1]
    x = (w["embeddings.word_embeddings.weight"][input_ids]
         + w["embeddings.position_embeddings.weight"][:L][None]
         + w["embeddings.token_type_embeddings.weight"][0][None, None])
    x = F.layer_norm(x, (H,), w["embeddings.LayerNorm.weight"], w["embeddings.LayerNorm.bias"], eps)
    add_mask = (1.0 - attention_mask[:, None, None, :].float()) * torch.finfo(torch.float32).min
    i = 0
    while f"encoder.layer.{i}.attention.self.query.weight" in w:
        P = f"encoder.layer.{i}."
        ctx = _mha(x, w[P + "attention.self.query.weight"], w[P + "attention.self.query.bias"],
                   w[P + "attention.self.key.weight"], w[P + "attention.self.key.bias"],
                   w[P + "attention.self.value.weight"], w[P + "attention.self.value.bias"],
                   heads, add_mask)
        y = F.linear(ctx, w[P + "attention.output.dense.weight"], w[P + "attention.output.dense.bias"])
        x = F.layer_norm(y + x, (H,), w[P + "attention.output.LayerNorm.weight"],
                         w[P + "attention.output.LayerNorm.bias"], eps)
        h = F.gelu(F.linear(x, w[P + "intermediate.dense.weight"], w[P + "intermediate.dense.bias"]))
        y = F.linear(h, w[P + "output.dense.weight"], w[P + "output.dense.bias"])
        x = F.layer_norm(y + x, (H,), w[P + "output.LayerNorm.weight"], w[P + "output.LayerNorm.bias"], eps)
        i += 1
        if collect is not None:
            collect[i] = x
    return x


def masked_meanpool_l2(hidden: torch.Tensor, attention_mask: torch.Tensor) -> torch.Tensor:
    """text_blocks.py:82-86 (mean-pool, clamp_min 1e-6) and :100 (v / (||v|| + 1e-9))."""
    m = attention_mask.unsqueeze(-1).float()
    rep = (hidden * m).sum(dim=1) / m.sum(dim=1).clamp_min(1e-6)
    return rep / (rep.norm(dim=-1, keepdim=True) + 1e-9)


def text_features(w, input_ids, attention_mask, heads: int = 12) -> torch.Tensor:
    """BERTContextEncoder.encode batched: (B,L) ids/mask -> (B,768)."""
    return masked_meanpool_l2(bert_last_hidden_state(w, input_ids, attention_mask, heads), attention_mask)


def field_mean_l2(parts: torch.Tensor) -> torch.Tensor:
    """encode_fields, text_blocks.py:126-128: mean of part vectors then L2-norm.
    parts: (..., m, D) -> (..., D)."""
    v = parts.mean(dim=-2)
    return v / (v.norm(dim=-1, keepdim=True) + 1e-9)


# --------------------------------------------------------------------------
# CLIP ViT-B/32 visual encoder (pre-LN)
# --------------------------------------------------------------------------
def vit_pooled(w: Dict[str, torch.Tensor], pixels: torch.Tensor, heads: int = 12,
               eps: float = 1e-5, collect: Optional[dict] = None) -> torch.Tensor:
    """CLIPVisionModel(...).pooler_output: (N,3,224,224) -> (N,768).  collect (optional dict): the residual stream after each
    layer, {1: ..., 2: ...} (CLIPVisionModel's hidden_states[1:])."""
    V = "vision_model."
    pw = w[V + "embeddings.patch_embedding.weight"]
    H, patch = pw.shape[0], pw.shape[-1]
    N = pixels.shape[0]
    x = F.conv2d(pixels.float(), pw, bias=None, stride=patch).flatten(2).transpose(1, 2)
    cls = w[V + "embeddings.class_embedding"].expand(N, 1, H)
    x = torch.cat([cls, x], dim=1) + w[V + "embeddings.position_embedding.weight"][None]
    x = F.layer_norm(x, (H,), w[V + "pre_layrnorm.weight"], w[V + "pre_layrnorm.bias"], eps)
    i = 0
    while V + f"encoder.layers.{i}.self_attn.q_proj.weight" in w:
        P = V + f"encoder.layers.{i}."
        h = F.layer_norm(x, (H,), w[P + "layer_norm1.weight"], w[P + "layer_norm1.bias"], eps)
        ctx = _mha(h, w[P + "self_attn.q_proj.weight"], w[P + "self_attn.q_proj.bias"],
                   w[P + "self_attn.k_proj.weight"], w[P + "self_attn.k_proj.bias"],
                   w[P + "self_attn.v_proj.weight"], w[P + "self_attn.v_proj.bias"], heads, None)
        x = x + F.linear(ctx, w[P + "self_attn.out_proj.weight"], w[P + "self_attn.out_proj.bias"])
        h = F.layer_norm(x, (H,), w[P + "layer_norm2.weight"], w[P + "layer_norm2.bias"], eps)
        h = F.linear(h, w[P + "mlp.fc1.weight"], w[P + "mlp.fc1.bias"])
        h = h * torch.sigmoid(1.702 * h)                               # quick_gelu
        x = x + F.linear(h, w[P + "mlp.fc2.weight"], w[P + "mlp.fc2.bias"])
        i += 1
        if collect is not None:
            collect[i] = x
    return F.layer_norm(x[:, 0], (H,), w[V + "post_layernorm.weight"], w[V + "post_layernorm.bias"], eps)


def visual_features(w, frames: torch.Tensor, heads: int = 12) -> torch.Tensor:
    """frames (B,F,3,224,224) or (B,3,224,224) -> (B,512): pooled -> bias-free
    projection -> L2-norm per frame; multi-frame = mean over F then L2-norm (the
    reference's only pooling idiom, text_blocks.py:126-128)."""
    if frames.dim() == 4:
        frames = frames[:, None]
    B, Fr = frames.shape[:2]
    pooled = vit_pooled(w, frames.reshape(B * Fr, *frames.shape[2:]), heads)
    e = F.linear(pooled, w["visual_projection.weight"])
    e = e / (e.norm(dim=-1, keepdim=True) + 1e-9)
    e = e.view(B, Fr, -1)
    if Fr == 1:
        return e[:, 0]
    return field_mean_l2(e)


# --------------------------------------------------------------------------
# Gradients (Tier-B backward).  The reference never trains its encoders (text_blocks.py:52,63), so these are the
# autograd gradients of the restatements above -- checked against the installed third-party classes' own autograd by
# tests/golden/make_golden.py tier_b_grads ("parity unpinned by the reference").
# --------------------------------------------------------------------------
def probe_loss(features: torch.Tensor, seed: int) -> torch.Tensor:
    """A fixed scalar functional of the features: sum(features * R), R ~ N(0, 1) seeded; d loss / d features = R."""
    g = torch.Generator().manual_seed(seed)
    return (features * torch.randn(features.shape, generator=g)).sum()


def text_feature_grads(w: Dict[str, torch.Tensor], input_ids, attention_mask, seed: int, heads: int = 12):
    """(features, {name: d probe_loss / d w[name]}) by autograd over text_features."""
    wl = {k: v.detach().clone().requires_grad_(True) for k, v in w.items()}
    feat = text_features(wl, input_ids, attention_mask, heads)
    probe_loss(feat, seed).backward()
    return feat.detach(), {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in wl.items()}


def visual_feature_grads(w: Dict[str, torch.Tensor], frames: torch.Tensor, seed: int, heads: int = 12):
    wl = {k: v.detach().clone().requires_grad_(True) for k, v in w.items()}
    feat = visual_features(wl, frames, heads)
    probe_loss(feat, seed).backward()
    return feat.detach(), {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in wl.items()}
