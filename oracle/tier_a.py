"""TEST INFRASTRUCTURE (see oracle/__init__.py) -- Tier-A oracle.

Functional torch-fp32 CPU restatement of the reference's trainable hot path:

  CrossModalTransformer.forward   src/models/fusion/cross_modal_transformer.py:134-210
  ForensicCoAttention.forward     src/models/fusion/cross_modal_transformer.py:39-55
  DeepTruthClassifier.forward     src/models/fusion/deep_truth_classifier.py:148-171
  _ObliviousTree / NODEEnsemble   src/models/fusion/deep_truth_classifier.py:54-74,88-90
  loss / clip / AdamW / StepLR    src/training/forensic_trainer.py:287-298,173-177,341

Parameters are plain dicts keyed by the reference's `state_dict` names
(SURVEY.md 8c).  Backward is torch autograd over this restatement, which is
also how the reference obtains its gradients.  The optimizer and the clip are
restated by hand (no torch.optim) so that the HIP kernels have a formula to be
compared with, and are themselves pinned against torch.optim.AdamW /
clip_grad_norm_ by tests/golden/make_golden.py.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------
# geometry (configs/model_configs/fusion.yaml:2-7, classifier.yaml:2-17)
# --------------------------------------------------------------------------
HIDDEN = 512
TEXT_DIM, AUDIO_DIM, VISUAL_DIM, TEMPORAL_DIM, GNN_DIM = 768, 128, 512, 256, 128
AUX_DIM = 2
TREES, DEPTH, TAU = 6, 4, 10.0
NUM_CLASSES = 2


def fusion_shapes(hidden: int = HIDDEN, gnn_dim: int = GNN_DIM, use_gnn: bool = True) -> "OrderedDict[str, Tuple[int, ...]]":
    """state_dict key -> shape, in the reference's registration order
    (cross_modal_transformer.py:96-130; semantic.* from semantic_forgery.py:73-82).  use_gnn=False is fusion.yaml's
    `use_gnn: false`: no gnn_proj, fuse_mlp.0 is 15H wide (:88,101-102,114-120)."""
    H = hidden
    s: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    for name, d in (("text", TEXT_DIM), ("audio", AUDIO_DIM), ("visual", VISUAL_DIM),
                    ("temporal", TEMPORAL_DIM)) + ((("gnn", gnn_dim),) if use_gnn else ()):
        s[f"{name}_proj.weight"] = (H, d)
        s[f"{name}_proj.bias"] = (H,)
    for name in ("text_proj", "vision_proj"):
        s[f"semantic.{name}.0.weight"] = (512, 512)
        s[f"semantic.{name}.0.bias"] = (512,)
    for blk in ("attn_tv", "attn_ta", "attn_vu"):
        for p in ("q", "k", "v"):
            s[f"{blk}.{p}.weight"] = (H, H)
            s[f"{blk}.{p}.bias"] = (H,)
        s[f"{blk}.evidence_proj.0.weight"] = (H, 3)
        s[f"{blk}.evidence_proj.0.bias"] = (H,)
        s[f"{blk}.evidence_proj.2.weight"] = (1, H)
        s[f"{blk}.evidence_proj.2.bias"] = (1,)
    s["fuse_mlp.0.weight"] = (2 * H, (16 if use_gnn else 15) * H)
    s["fuse_mlp.0.bias"] = (2 * H,)
    s["fuse_mlp.3.weight"] = (H, 2 * H)
    s["fuse_mlp.3.bias"] = (H,)
    s["classifier.weight"] = (2, H)
    s["classifier.bias"] = (2,)
    return s


def clf_shapes(hidden: int = HIDDEN, in_dim: int = HIDDEN, aux_dim: int = AUX_DIM,
               trees: int = TREES, depth: int = DEPTH, classes: int = NUM_CLASSES):
    """deep_truth_classifier.py:115-138."""
    H = hidden
    s: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    s["temperature"] = ()
    s["pre.0.weight"] = (H, in_dim + aux_dim)
    s["pre.0.bias"] = (H,)
    s["pre.3.weight"] = (H, H)
    s["pre.3.bias"] = (H,)
    for t in range(trees):
        s[f"node.trees.{t}.tau"] = ()
        s[f"node.trees.{t}.leaf_logits"] = (1 << depth, classes)
        for k in range(depth):
            s[f"node.trees.{t}.gates.{k}"] = (H,)
        for k in range(depth):
            s[f"node.trees.{t}.thresh.{k}"] = (1,)
    s["bypass.weight"] = (classes, H)
    s["bypass.bias"] = (classes,)
    return s


# keys that never receive a gradient under the trainer's loss
def no_grad_keys_fusion():
    return [k for k in fusion_shapes() if k.startswith("semantic.") or k.startswith("classifier.")]


def no_grad_keys_clf():
    return ["temperature"] + [f"node.trees.{t}.tau" for t in range(TREES)]


def seeded_params(seed: int, use_gnn: bool = True) -> Tuple[Dict[str, torch.Tensor], Dict[str, torch.Tensor]]:
    """Deterministic, *perturbed* parameters (NODE leaves/gates non-zero so the
    tree path is exercised -- SURVEY.md 8c).  Regenerated from the seed on both
    sides of every comparison so fixtures stay small; a checksum of the result is
    stored in each fixture to catch generator drift."""
    g = torch.Generator().manual_seed(seed)

    def gen(key: str, shape) -> torch.Tensor:
        if key.endswith("tau"):
            return torch.tensor(TAU)
        if key == "temperature":
            return torch.tensor(1.3)
        if ".gates." in key:
            return torch.randn(shape, generator=g) * 1.5
        if ".thresh." in key:
            return torch.randn(shape, generator=g) * 0.02
        if key.endswith("leaf_logits"):
            return torch.randn(shape, generator=g) * 0.5
        if len(shape) == 2:
            return torch.randn(shape, generator=g) * (1.0 / math.sqrt(shape[1]))
        return torch.randn(shape, generator=g) * 0.05

    fus = OrderedDict((k, gen(k, s)) for k, s in fusion_shapes(use_gnn=use_gnn).items())
    clf = OrderedDict((k, gen(k, s)) for k, s in clf_shapes().items())
    return fus, clf


def seeded_batch(seed: int, B: int) -> Dict[str, torch.Tensor]:
    """FakeSV-shaped synthetic feature batch (SURVEY.md 8c 'Golden vectors')."""
    g = torch.Generator().manual_seed(seed)
    return {
        "text_features": torch.randn(B, TEXT_DIM, generator=g),
        "audio_features": torch.randn(B, AUDIO_DIM, generator=g),
        "visual_features": torch.randn(B, VISUAL_DIM, generator=g),
        "temporal_features": torch.randn(B, TEMPORAL_DIM, generator=g),
        "gnn_feat": torch.randn(B, GNN_DIM, generator=g),
        "aux": torch.rand(B, AUX_DIM, generator=g),
        "label": torch.randint(0, 2, (B,), generator=g),
    }


# --------------------------------------------------------------------------
# forward
# --------------------------------------------------------------------------
def _co_attention(p: Dict[str, torch.Tensor], blk: str, x, y, evidence):
    """ForensicCoAttention.forward, cross_modal_transformer.py:39-55."""
    H = x.shape[1]
    q = F.linear(x, p[f"{blk}.q.weight"], p[f"{blk}.q.bias"])
    k = F.linear(y, p[f"{blk}.k.weight"], p[f"{blk}.k.bias"])
    v = F.linear(y, p[f"{blk}.v.weight"], p[f"{blk}.v.bias"])
    score = (q * k).sum(dim=-1, keepdim=True) / (H ** 0.5)
    attn = torch.sigmoid(score)
    e = F.linear(evidence, p[f"{blk}.evidence_proj.0.weight"], p[f"{blk}.evidence_proj.0.bias"])
    e = F.linear(F.gelu(e), p[f"{blk}.evidence_proj.2.weight"], p[f"{blk}.evidence_proj.2.bias"])
    gated = torch.sigmoid(e)
    return gated * (attn * v) + (1.0 - gated) * (0.5 * (x + y))


def _drop(x, p: float, train: bool):
    return F.dropout(x, p, training=train) if (train and p > 0) else x


def fusion_forward(p: Dict[str, torch.Tensor], feats: Dict[str, torch.Tensor],
                   dropout: float = 0.1, train: bool = False) -> Dict[str, torch.Tensor]:
    """CrossModalTransformer.forward, cross_modal_transformer.py:134-210."""
    t = F.linear(feats["text_features"].float(), p["text_proj.weight"], p["text_proj.bias"])
    a = F.linear(feats["audio_features"].float(), p["audio_proj.weight"], p["audio_proj.bias"])
    v = F.linear(feats["visual_features"].float(), p["visual_proj.weight"], p["visual_proj.bias"])
    u = F.linear(feats["temporal_features"].float(), p["temporal_proj.weight"], p["temporal_proj.bias"])

    with torch.no_grad():  # :153-164
        def cos01(x1, x2):
            c = (F.normalize(x1, dim=-1) * F.normalize(x2, dim=-1)).sum(dim=-1, keepdim=True)
            return 0.5 * (c.clamp(-1, 1) + 1.0)
        semantic_conflict = 1.0 - cos01(t, v)
        emo = t.abs().mean(dim=-1, keepdim=True).tanh()
        delay = 1.0 - cos01(t, u)
        z = torch.zeros_like(emo)

    tv = _co_attention(p, "attn_tv", t, v, torch.cat([semantic_conflict, emo, z], dim=-1))
    ta = _co_attention(p, "attn_ta", t, a, torch.cat([emo, z, z], dim=-1))
    vu = _co_attention(p, "attn_vu", v, u, torch.cat([delay, z, z], dim=-1))

    pairs = [t + a, t * a, (t - a).abs(), t + v, t * v, (t - v).abs(), t + u, v + u]  # :172-178
    cat = [t, a, v, u, *pairs, tv, ta, vu]
    if feats.get("gnn_feat") is not None and "gnn_proj.weight" in p:  # :184-187 (`use_gnn: false`: no gnn_proj, gnn_feat ignored)
        cat.append(F.linear(feats["gnn_feat"].float(), p["gnn_proj.weight"], p["gnn_proj.bias"]))
    fused_cat = torch.cat(cat, dim=-1)

    h = _drop(F.gelu(F.linear(fused_cat, p["fuse_mlp.0.weight"], p["fuse_mlp.0.bias"])), dropout, train)
    fused = _drop(F.gelu(F.linear(h, p["fuse_mlp.3.weight"], p["fuse_mlp.3.bias"])), dropout, train)
    logits = F.linear(fused, p["classifier.weight"], p["classifier.bias"])
    return {"fused": fused, "logits": logits, "fused_cat": fused_cat,
            "forensic": {"emotion_intensity": emo.squeeze(-1),
                         "semantic_conflict": semantic_conflict.squeeze(-1),
                         "temporal_delay": delay.squeeze(-1)}}


def classifier_forward(p: Dict[str, torch.Tensor], fused, aux: Optional[torch.Tensor],
                       dropout: float = 0.1, node_dropout: float = 0.3, train: bool = False,
                       trees: int = TREES, depth: int = DEPTH) -> Dict[str, torch.Tensor]:
    """DeepTruthClassifier.forward, deep_truth_classifier.py:148-171."""
    x = fused.float()
    if aux is not None:
        x = torch.cat([x, aux.float()], dim=-1)
    h = _drop(F.gelu(F.linear(x, p["pre.0.weight"], p["pre.0.bias"])), dropout, train)
    h = _drop(F.gelu(F.linear(h, p["pre.3.weight"], p["pre.3.bias"])), dropout, train)

    outs = []
    for t in range(trees):  # _ObliviousTree.forward :54-74
        probs = h.new_ones((h.shape[0], 1))
        tau = p[f"node.trees.{t}.tau"]
        for k in range(depth):
            alpha = torch.softmax(p[f"node.trees.{t}.gates.{k}"], dim=0)
            feat = (h * alpha).sum(dim=-1, keepdim=True)
            s = torch.sigmoid(tau * (feat - p[f"node.trees.{t}.thresh.{k}"]))
            probs = torch.cat([probs * (1.0 - s), probs * s], dim=1)
        outs.append(_drop(probs @ p[f"node.trees.{t}.leaf_logits"], node_dropout, train))
    logits = torch.stack(outs, 0).mean(0) + F.linear(h, p["bypass.weight"], p["bypass.bias"])
    T = torch.clamp(p["temperature"], min=0.5, max=5.0)
    return {"logits": logits, "probs": F.softmax(logits / T, dim=-1), "temperature": T, "h": h}


def forward_batch(fus, clf, batch, train: bool = False, dropout: float = 0.1):
    """ForensicTrainer._forward_batch (forensic_trainer.py:238-271) on a ready batch."""
    fo = fusion_forward(fus, batch, dropout=dropout, train=train)
    co = classifier_forward(clf, fo["fused"], batch["aux"], dropout=dropout,
                            node_dropout=0.3 if dropout > 0 else 0.0, train=train)
    return {"logits": co["logits"], "probs": co["probs"], "y": batch["label"],
            "forensic": fo["forensic"], "fused": fo["fused"], "fusion_logits": fo["logits"]}


# --------------------------------------------------------------------------
# train step: CE -> backward -> clip -> AdamW          forensic_trainer.py:287-298
# --------------------------------------------------------------------------
class AdamWState:
    """Hand-restated torch.optim.AdamW (lr 2e-4, wd 1e-4, betas (0.9,0.999), eps 1e-8;
    forensic_trainer.py:176) + StepLR(3, 0.7) (:177,341)."""

    def __init__(self, lr=2e-4, weight_decay=1e-4, betas=(0.9, 0.999), eps=1e-8):
        self.base_lr, self.lr, self.wd, self.betas, self.eps = lr, lr, weight_decay, betas, eps
        self.m: Dict[str, torch.Tensor] = {}
        self.v: Dict[str, torch.Tensor] = {}
        self.t: Dict[str, int] = {}
        self.epoch = 0

    def scheduler_step(self, step_size=3, gamma=0.7):
        self.epoch += 1
        self.lr = self.base_lr * (gamma ** (self.epoch // step_size))

    @torch.no_grad()
    def step(self, params: Dict[str, torch.Tensor], grads: Dict[str, Optional[torch.Tensor]]):
        b1, b2 = self.betas
        for k, p in params.items():
            g = grads.get(k)
            if g is None:           # AdamW skips params whose .grad is None (no decay either)
                continue
            if k not in self.m:
                self.m[k] = torch.zeros_like(p)
                self.v[k] = torch.zeros_like(p)
                self.t[k] = 0
            self.t[k] += 1
            t = self.t[k]
            p.mul_(1.0 - self.lr * self.wd)
            self.m[k].mul_(b1).add_(g, alpha=1 - b1)
            self.v[k].mul_(b2).addcmul_(g, g, value=1 - b2)
            bc1 = 1 - b1 ** t
            bc2 = 1 - b2 ** t
            denom = (self.v[k].sqrt() / math.sqrt(bc2)).add_(self.eps)
            p.addcdiv_(self.m[k], denom, value=-(self.lr / bc1))


def clip_grads_(grads: Dict[str, Optional[torch.Tensor]], max_norm: float) -> float:
    """torch.nn.utils.clip_grad_norm_ (global L2, grads that are None skipped);
    forensic_trainer.py:292-297.  Returns the pre-clip total norm."""
    gs = [g for g in grads.values() if g is not None]
    total = torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(g) for g in gs]))
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    for g in gs:
        g.mul_(coef)
    return float(total)


def loss_and_grads(fus, clf, batch, train: bool = False, dropout: float = 0.1):
    """Forward + F.cross_entropy(mean) + autograd backward.  Returns
    (out, loss, grads_fusion, grads_clf) with None for keys that get no grad."""
    fl = {k: v.detach().clone().requires_grad_(v.dtype.is_floating_point) for k, v in fus.items()}
    cl = {k: v.detach().clone().requires_grad_(v.dtype.is_floating_point and not k.endswith("tau"))
          for k, v in clf.items()}
    out = forward_batch(fl, cl, batch, train=train, dropout=dropout)
    loss = F.cross_entropy(out["logits"], batch["label"])
    loss.backward()
    gf = {k: v.grad for k, v in fl.items()}
    gc = {k: v.grad for k, v in cl.items()}
    return out, loss.detach(), gf, gc


def train_step(fus, clf, batch, opt: AdamWState, grad_clip: float = 5.0,
               train: bool = False, dropout: float = 0.1):
    """One iteration of ForensicTrainer._epoch_loop's train branch
    (forensic_trainer.py:285-298).  Mutates fus/clf in place."""
    out, loss, gf, gc = loss_and_grads(fus, clf, batch, train=train, dropout=dropout)
    grads = {**{"fusion." + k: g for k, g in gf.items()}, **{"clf." + k: g for k, g in gc.items()}}
    total = clip_grads_(grads, grad_clip) if grad_clip and grad_clip > 0 else float("nan")
    params = {**{"fusion." + k: v for k, v in fus.items()}, **{"clf." + k: v for k, v in clf.items()}}
    opt.step(params, grads)
    return out, float(loss), total
