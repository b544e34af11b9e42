"""CrossModalTransformer -- MI355X-native mirror of the reference's fusion module
(src/models/fusion/cross_modal_transformer.py:17-55,62-210).

Same constructor, `forward(feats) -> {"fused", "logits", "forensic"}` contract, YAML keys and
`state_dict` names (including the four dead `semantic.*` tensors the reference carries,
SURVEY.md 8c), same parameter initialisation order (so an identical torch seed yields identical
initial weights).  All arithmetic runs in libultrafnd_hip.so; there is no CPU path.  Unlike the
reference (which pins `mps|cpu` inside forward, :93,139) the module follows its parameters'
device.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn

from . import _lib as L
from .arena import ArenaModule, Group, rehome
from .config_utils import ConfigManager
from .state import StepStateBuffer


class ForensicCoAttention(nn.Module):
    """Parameter container for one evidence-gated co-attention block
    (cross_modal_transformer.py:17-38).  Its arithmetic is fused into the parent's kernels."""

    def __init__(self, hidden_dim: int, evidence_dim: int = 3):
        super().__init__()
        self.h = hidden_dim
        self.q = nn.Linear(hidden_dim, hidden_dim)
        self.k = nn.Linear(hidden_dim, hidden_dim)
        self.v = nn.Linear(hidden_dim, hidden_dim)
        self.evidence_proj = nn.Sequential(nn.Linear(evidence_dim, hidden_dim), nn.GELU(), nn.Linear(hidden_dim, 1))

    def forward(self, x, y, evidence):  # pragma: no cover - fused into CrossModalTransformer
        raise RuntimeError("ForensicCoAttention.forward: this class is a parameter container here -- the three blocks are evaluated inside "
                           "CrossModalTransformer's fused HIP kernels (one stacked (9H, H) GEMM + coattn_pairs); call the parent module "
                           "(INTEGRATION.md section A lists this difference from cross_modal_transformer.py:39-55)")


class _SemanticParams(nn.Module):
    """The two unused Linear(512,512) the reference registers through SemanticForgeryAnalyzer
    (src/models/semantic_forgery.py:73-82): kept only so checkpoints interchange."""

    def __init__(self):
        super().__init__()
        self.text_proj = nn.Sequential(nn.Linear(512, 512))
        self.vision_proj = nn.Sequential(nn.Linear(512, 512))


_QKV_ORDER = [("attn_tv", "q"), ("attn_ta", "q"), ("attn_tv", "k"), ("attn_tv", "v"), ("attn_vu", "q"),
              ("attn_ta", "k"), ("attn_ta", "v"), ("attn_vu", "k"), ("attn_vu", "v")]
_BLOCKS = ("attn_tv", "attn_ta", "attn_vu")
_PROJ = (("text", 768), ("audio", 128), ("visual", 512), ("temporal", 256))


class CrossModalTransformer(ArenaModule):
    def __init__(self, config_path: str = "configs/model_configs/fusion.yaml"):
        super().__init__()
        cfg = ConfigManager().load_config(config_path)
        self.hidden = int(cfg.get("hidden_dim", 512))
        self.dropout = float(cfg.get("dropout", 0.3))
        self.use_gnn = bool(cfg.get("use_gnn", True))
        self.gnn_dim = int(cfg.get("gnn_dim", 128))
        self.dtype = torch.float32
        if self.hidden not in (256, 512, 1024):      # (the reference's YAML takes any width, cross_modal_transformer.py:87; the kernels do not)
            raise ValueError(f"fusion.yaml: hidden_dim={self.hidden}: the HIP kernels support hidden_dim in {{256, 512, 1024}}")
        H = self.hidden
        # construction order == the reference's (:96-130) so the RNG stream matches
        self.text_proj = nn.Linear(768, H)
        self.audio_proj = nn.Linear(128, H)
        self.visual_proj = nn.Linear(512, H)
        self.temporal_proj = nn.Linear(256, H)
        if self.use_gnn:           # `use_gnn: false` (:88,101-102): no gnn_proj, the concat is 15H wide, gnn_feat is ignored
            self.gnn_proj = nn.Linear(self.gnn_dim, H)
        self.semantic = _SemanticParams()
        self.attn_tv = ForensicCoAttention(H, 3)
        self.attn_ta = ForensicCoAttention(H, 3)
        self.attn_vu = ForensicCoAttention(H, 3)
        self.include_pairs = True
        self.fused_dim = (16 if self.use_gnn else 15) * H        # (:114-120)
        self.fuse_mlp = nn.Sequential(nn.Linear(self.fused_dim, 2 * H), nn.GELU(), nn.Dropout(self.dropout),
                                      nn.Linear(2 * H, H), nn.GELU(), nn.Dropout(self.dropout))
        self.classifier = nn.Linear(H, 2)
        self._ws: Dict[Tuple[int, bool], torch.Tensor] = {}
        self._gen: Dict[int, int] = {}
        self._ptab = None
        self._gtab = None
        self._rng: Optional[StepStateBuffer] = None
        rehome([self], [""])

    # ------------------------------------------------------------------ arena layout
    def _arena_groups(self) -> Tuple[List[Group], List[Group]]:
        H = self.hidden
        fuse = [("fuse_mlp.0.weight", (2 * H, self.fused_dim)), ("fuse_mlp.0.bias", (2 * H,)),
                ("fuse_mlp.3.weight", (H, 2 * H)), ("fuse_mlp.3.bias", (H,))]
        qkv_w = [(f"{b}.{p}.weight", (H, H)) for b, p in _QKV_ORDER]
        qkv_b = [(f"{b}.{p}.bias", (H,)) for b, p in _QKV_ORDER]
        ev = []
        for b in _BLOCKS:
            ev += [(f"{b}.evidence_proj.0.weight", (H, 3)), (f"{b}.evidence_proj.0.bias", (H,)),
                   (f"{b}.evidence_proj.2.weight", (1, H)), (f"{b}.evidence_proj.2.bias", (1,))]
        proj = []
        for n, d in _PROJ + ((("gnn", self.gnn_dim),) if self.use_gnn else ()):
            proj += [(f"{n}_proj.weight", (H, d)), (f"{n}_proj.bias", (H,))]
        # gradient-ready order of backward: fuse_mlp first, then attention, then projections
        grad = [fuse, qkv_w, qkv_b] + [[e] for e in ev] + [[p] for p in proj]
        nograd = [[("semantic.text_proj.0.weight", (512, 512))], [("semantic.text_proj.0.bias", (512,))],
                  [("semantic.vision_proj.0.weight", (512, 512))], [("semantic.vision_proj.0.bias", (512,))],
                  [("classifier.weight", (2, H))], [("classifier.bias", (2,))]]
        return grad, nograd

    def _on_rehome(self) -> None:
        self._ptab = self._gtab = None
        self._ws.clear()
        self._rng = None

    # ------------------------------------------------------------------ C tables
    def dims(self, clf=None) -> L.Dims:
        d = L.Dims()
        d.hidden, d.text_dim, d.audio_dim, d.visual_dim, d.temporal_dim = self.hidden, 768, 128, 512, 256
        d.gnn_dim, d.aux_dim, d.trees, d.depth, d.classes = (self.gnn_dim if self.use_gnn else 0), 2, 6, 4, 2      # gnn_dim 0 = no GNN slot
        d.fusion_dropout, d.clf_dropout, d.node_dropout = self.dropout, 0.1, 0.3
        return d

    def _table(self, getter) -> L.FusionParams:
        t = L.FusionParams()
        for n in ("text", "audio", "visual", "temporal") + (("gnn",) if self.use_gnn else ()):
            setattr(t, f"{n}_w", getter(f"{n}_proj.weight").data_ptr())
            setattr(t, f"{n}_b", getter(f"{n}_proj.bias").data_ptr())
        t.qkv_w = getter("attn_tv.q.weight").data_ptr()
        t.qkv_b = getter("attn_tv.q.bias").data_ptr()
        for i, b in enumerate(_BLOCKS):
            t.ev0_w[i] = getter(f"{b}.evidence_proj.0.weight").data_ptr()
            t.ev0_b[i] = getter(f"{b}.evidence_proj.0.bias").data_ptr()
            t.ev2_w[i] = getter(f"{b}.evidence_proj.2.weight").data_ptr()
            t.ev2_b[i] = getter(f"{b}.evidence_proj.2.bias").data_ptr()
        t.fuse0_w, t.fuse0_b = getter("fuse_mlp.0.weight").data_ptr(), getter("fuse_mlp.0.bias").data_ptr()
        t.fuse3_w, t.fuse3_b = getter("fuse_mlp.3.weight").data_ptr(), getter("fuse_mlp.3.bias").data_ptr()
        return t

    def param_table(self) -> L.FusionParams:
        if self._ptab is None:
            t = self._table(self.aview)
            t.cls_w, t.cls_b = self.aview("classifier.weight").data_ptr(), self.aview("classifier.bias").data_ptr()
            self._ptab = t
        return self._ptab

    def grad_table(self) -> L.FusionParams:
        """Gradient pointers (arena layout).  The aux head lives in the no-grad region: it gets
        a side buffer that is only written when a gradient arrives at the fusion logits."""
        if self._gtab is None:
            t = self._table(self.gview)
            self._cls_grad = torch.zeros(2 * self.hidden + 64, dtype=torch.float32, device=self._arena.device)
            t.cls_w, t.cls_b = self._cls_grad.data_ptr(), self._cls_grad[2 * self.hidden:].data_ptr()
            self._gtab = t
        return self._gtab

    def workspace(self, B: int, train: bool) -> torch.Tensor:
        key = (B, bool(train))
        if key not in self._ws:
            d = self.dims()
            n = L.lib().ufnd_fusion_workspace_floats(C.byref(d), B)
            self._ws[key] = torch.empty(n, dtype=torch.float32, device=self._arena.device)
        return self._ws[key]

    def rng(self) -> StepStateBuffer:
        """Private dropout key/counter used when the module runs outside the fused trainer."""
        if self._rng is None:
            self._rng = StepStateBuffer(self._arena.device, seed=torch.initial_seed() + 0x5F5)
        return self._rng

    # ------------------------------------------------------------------ forward
    def forward(self, feats: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        dev = self._arena.device
        if dev.type != "cuda":
            raise L.UltrafndHipError("CrossModalTransformer runs on a HIP device only: call .to('cuda') "
                                     "(there is no CPU fallback; the reference's CPU path is not part of this package)")
        if self.use_gnn and feats.get("gnn_feat") is None:
            raise RuntimeError("gnn_feat is required: fuse_mlp expects the 16*hidden concat "
                               "(the reference fails the same way without it, cross_modal_transformer.py:184-197)")
        xs = [L.f32c(feats[k].to(dev)) for k in ("text_features", "audio_features", "visual_features", "temporal_features")]
        # use_gnn false: gnn_feat is ignored as in the reference (:184); a zero-width placeholder keeps the Function's arity
        xs.append(L.f32c(feats["gnn_feat"].to(dev)) if self.use_gnn else torch.empty(xs[0].shape[0], 0, dtype=torch.float32, device=dev))
        for x, (n, dim) in zip(xs, _PROJ + (("gnn", self.gnn_dim if self.use_gnn else 0),)):
            if x.dim() != 2 or x.shape[1] != dim or x.shape[0] != xs[0].shape[0]:
                raise RuntimeError(f"{n} features: expected (B,{dim}), got {tuple(x.shape)}")
        from .functional import FusionFunction
        needs_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        fused, logits, forensic = FusionFunction.apply(self, self.training, needs_grad, *xs,
                                                       *[p for p in self.parameters() if p.requires_grad])
        return {"fused": fused, "logits": logits,
                "forensic": {"emotion_intensity": forensic[0], "semantic_conflict": forensic[1],
                             "temporal_delay": forensic[2]}}
