"""ForensicTrainer -- MI355X-native mirror of the reference's training step API
(src/training/forensic_trainer.py:60-107,139-396).

Kept signature-for-signature: `TrainConfig` fields and defaults (:90-107), `CachedTensorDataset`
item schema (:60-83), `ForensicTrainer(cfg)` with `.fusion .clf .gnn .optim .scheduler
.train_loader .val_loader .test_loader .cache .ckpt_path .best_val_auc`, `fit()`, `test()`,
`_forward_batch`, `_epoch_loop`, `_build_dataloaders`, the `best.pt` dict (:355-360).

What is different, on purpose (SURVEY.md 3 "hot loops today", 8e):
  * the feature cache lives in HBM; a batch is an index gather on the device (no per-sample
    Python collate, no per-step H2D copies);
  * one step = the fused C-ABI sequence fusion fwd -> classifier fwd -> CE -> classifier bwd ->
    fusion bwd -> [all-reduce] -> global-norm clip + AdamW over one flat arena, optionally
    replayed from a captured hipGraph; ~30 launches instead of ~7,700 ATen dispatches;
  * loss / probabilities / forensic scalars stay on the device until the epoch ends (the
    reference syncs 5x per step, :301-313);
  * data parallel: batches are sharded over ranks, gradients all-reduced (dp.py);
  * optional `encode_inline`: raw token ids / frames go through the native BERT / ViT encoders
    inside the step (the north-star's "text+vision" step) instead of precomputed features.
Where things live: data.py (device-resident split + loader), head_step.py (the head's forward / loss / backward on static
buffers, graphs), pipeline.py (encoder streams, graphs, lookahead groups, pipelined steps), dp.py (gradient exchange, metric
gather, checkpoint), this file (the reference's trainer API: construction, epoch loop, fit / test).
Out of scope: FakeSVRawDataset / build_gnn_cache_from_raw_dataset (dataset
preprocessing needing the FakeSV corpus): the trainer takes the cache dict they would have produced.  When that
cache carries `ocr_sets` instead of `gnn_Z`, the graph side of the reference's construction (OCR-Jaccard
adjacency, SimpleGCN, its two pre-training steps; forensic_trainer.py:184-224) runs here too (gcn.py).
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from . import _lib as L
from .arena import rehome
from .classifier import DeepTruthClassifier
from .data import CachedTensorDataset, DeviceBatchLoader, IndexedBatch, _batch_size, synthetic_cache  # noqa: F401  (re-exported)
from .dp import FactorExchange, GradReducer, as_comm, broadcast_from_rank0, gather_epoch_outputs, save_checkpoint
from .fusion import CrossModalTransformer
from .head_step import HeadStep
from .metrics import aggregate_epoch_metrics, pretty_print
from .optim import CosineAnnealingLR, FusedAdamW, StepLR
from .pipeline import EncoderPipeline


@dataclass
class TrainConfig:
    data_root: str
    ocr_phrase_pkl: Optional[str]
    out_dir: str = "outputs"
    batch_size: int = 16
    epochs: int = 8
    lr: float = 2e-4
    weight_decay: float = 1e-4
    gnn_dim: int = 128
    gnn_overlap_thresh: float = 0.12
    seed: int = 42
    use_mps: bool = True          # kept for signature parity; ignored (the device is HIP)
    use_gnn: bool = True
    save_best: bool = True
    grad_clip: float = 5.0
    early_stop_patience: int = 3
    # ---- MI355X additions (all optional)
    device: str = "cuda"
    use_graph: bool = True        # replay the step from a captured hipGraph
    encode_inline: bool = False   # run the native text / visual encoders inside the step
    # criterion / schedule of the integrated variant (forensic_trainer_integrated.py:30-46,151-166); defaults = the main trainer
    label_smoothing: float = 0.0
    class_weighting: bool = False  # CE weights 0.5 * total / count(class) from the cache's labels
    use_cosine: bool = False       # CosineAnnealingLR(T_max = epochs, eta_min = lr * min_lr_scale) instead of StepLR(3, 0.7)
    min_lr_scale: float = 0.1
    # compute units of the (text encoder, visual encoder) streams when the encoders run inside the step (streams.py);
    # None = ordinary streams.  The head -> exchange -> optimizer chain keeps the whole chip at high priority.
    cu_split: Optional[Tuple[int, int]] = None
    # the integrated variant's in-graph GNN (forensic_trainer_integrated.py:132-138,203-224): gnn_feat of a mini-batch comes from
    # a GNNModel over the batch's own OCR-Jaccard graph and is trained WITH the head (the main trainer's gnn_feat is a detached table)
    gnn_in_graph: bool = False
    # encode_inline: the frozen encoders run over this many consecutive batches per pass (train_group_pipelined); the head,
    # the exchange and the optimizer still step batch by batch, bit-identical to one batch per pass.  1 = off.
    encoder_lookahead: int = 4
    # experiments / diagnostics that used to be environment variables
    head_graph: bool = True        # False: the head's forward / backward eagerly (dW kernels on a side stream) even with use_graph
    # the loader rotates a small fixed set of device input buffers (bench.py: four): encoder graphs read them in place, one graph
    # per buffer set (pipeline.py).  False: every batch that is not one of the trainer's own group buffers is staged.
    persistent_inputs: bool = False
    # gradient exchange variants (dp.py): payload "fp32" | "bf16", algorithm "all_reduce" | "rs_ag" | "factors" (the head's Linear
    # gradients formed from all-gathered factor panels instead of being reduced: 20 x fewer bytes per rank; fp32 payload only)
    grad_payload: str = "fp32"
    grad_exchange: str = "all_reduce"
    # fine-tune the encoders with the head (encoder_train.py: forward with saved activations + hand-written backward; needs
    # encode_inline and both encoders).  The reference keeps them frozen (text_blocks.py:52,63): False reproduces it.
    train_encoders: bool = False
    # the head's forward + CE and its backward as two C-ABI calls over both modules (21 launches, same bits); False: the five module-level calls (26)
    fused_head: bool = True


class ForensicTrainer:
    def __init__(self, cfg: TrainConfig, cache: Optional[Dict] = None, text_encoder=None, visual_encoder=None,
                 group=None, temporal_net=None, force_exchange: bool = False):
        """`group`: a torch.distributed process group (None = the default one) or a dp.Collectives.  `force_exchange` runs the
        gradient exchange at world size 1 too (the RCCL path on one GPU: tests, bench --gpus 1 under torchrun)."""
        self.cfg = cfg
        os.makedirs(cfg.out_dir, exist_ok=True)
        self.device = torch.device(cfg.device)
        if self.device.type != "cuda":
            raise L.UltrafndHipError("ForensicTrainer needs a HIP device (TrainConfig.device='cuda'); no CPU path")
        if self.device.index is None:           # "cuda" -> the current device, spelled out (tensors report cuda:N)
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.dtype = torch.float32
        self.comm = as_comm(group)
        self.group = self.comm
        self.world, self.rank = self.comm.world, self.comm.rank
        torch.manual_seed(cfg.seed)
        np.random.seed(cfg.seed)

        if cache is None:
            path = os.path.join(cfg.data_root, "feature_cache.npz")
            if not os.path.exists(path):
                raise FileNotFoundError(
                    f"{path} not found.  Building the cache from the raw FakeSV corpus (FakeSVRawDataset + "
                    "build_gnn_cache_from_raw_dataset) is dataset preprocessing outside this package's scope; pass "
                    "`cache=` (keys ids, labels, text, audio, visual, temporal, aux, gnn_Z, split) or save one there.")
            z = np.load(path, allow_pickle=False)
            cache = {k: z[k] for k in z.files if k not in ("split_train", "split_val", "split_test")}
            cache["split"] = (z["split_train"], z["split_val"], z["split_test"])
        if not cfg.use_gnn:
            raise ValueError("use_gnn=False: the reference's trainer builds its fusion head from fusion.yaml (use_gnn: true, 16H concat) "
                             "whatever this flag says and then feeds it no gnn_feat -- a 7,680-wide concat into an 8,192-wide Linear, "
                             "which raises (cross_modal_transformer.py:184-197).  Refused here as well; a head WITHOUT the GNN slot is "
                             "fusion.yaml's `use_gnn: false` (CrossModalTransformer supports it).")
        self.gnn = None   # the GCN that produced gnn_Z is not part of the step (its output is detached: :209-211)
        self.gnn_model = None
        if cfg.gnn_in_graph:
            if "ocr_sets" not in cache:
                raise KeyError("gnn_in_graph=True needs cache['ocr_sets'] (the mini-batch graph is built from them, "
                               "forensic_trainer_integrated.py:207-213)")
            cache = dict(cache)
            cache.setdefault("gnn_Z", np.zeros((len(cache["labels"]), cfg.gnn_dim), dtype=np.float32))    # (placeholder: never read)
        if "gnn_Z" not in cache:
            if "ocr_sets" not in cache:
                raise KeyError("cache needs either 'gnn_Z' (N, gnn_dim) or 'ocr_sets' (N phrase sets) to build it from "
                               "(forensic_trainer.py:184-211)")
            # ForensicTrainer._build_gnn: node features, OCR-Jaccard adjacency, SimpleGCN, two pre-training steps
            from .gcn import build_gnn_embeddings
            self.gnn, self.X, self.Adj, Z = build_gnn_embeddings(cache, cfg.gnn_dim, cfg.gnn_overlap_thresh, self.device)
            cache = dict(cache)
            cache["gnn_Z"] = Z.detach()
        self.cache = cache
        (self.tr_idx, self.va_idx, self.te_idx) = cache["split"]

        self.train_loader, self.val_loader, self.test_loader = self._build_dataloaders()

        self.fusion = CrossModalTransformer(config_path="configs/model_configs/fusion.yaml").to(self.device)
        self.clf = DeepTruthClassifier(config_path="configs/model_configs/classifier.yaml").to(self.device)
        if not self.fusion.use_gnn:
            raise ValueError("configs/model_configs/fusion.yaml says use_gnn: false while TrainConfig.use_gnn is true: the trainer would feed "
                             "gnn_feat to a head without the GNN slot (the reference silently ignores it; set use_gnn in both places alike)")
        # trainable encoders: their masters join the arena behind the head's, in gradient-ready order (text, then visual)
        self.text_bp = self.vis_bp = None
        self._enc_side = None      # second stream of the trainable-encoder step (visual encoder beside the text encoder)
        extra = []
        if cfg.train_encoders:
            if not cfg.encode_inline or text_encoder is None or visual_encoder is None or cfg.gnn_in_graph:
                raise ValueError("train_encoders=True needs encode_inline=True with text_encoder= and visual_encoder= (and not gnn_in_graph)")
            from .encoder_train import TextBackprop, VisualBackprop
            text_encoder._require_hip(); visual_encoder._require_hip()
            self.text_bp, self.vis_bp = TextBackprop(text_encoder), VisualBackprop(visual_encoder)
            extra = [[("text." + k, s) for k, s in g] for g in self.text_bp.groups()] + [[("vis." + k, s) for k, s in g] for g in self.vis_bp.groups()]
        # one flat arena for both modules: clf first (its gradients are ready first in backward)
        if cfg.gnn_in_graph:
            from .gnn_model import GNNModel
            self.gnn_model = GNNModel(in_dim=416, hid=256, out_dim=cfg.gnn_dim, dropout=0.1).to(self.device)   # :136
            self.arena = rehome([self.clf, self.fusion, self.gnn_model], ["clf.", "fusion.", "gnn."])        # its gradients are ready last
            self._epoch = 0
        else:
            self.arena = rehome([self.clf, self.fusion], ["clf.", "fusion."], extra_grad_groups=extra)
        # buckets in gradient-ready order: [classifier | fuse_mlp] is complete after the first phase of backward, the rest of the
        # head after the second; with trainable encoders, the text encoder's and the visual encoder's gradients follow
        bounds = [self.arena.offsets["fusion.attn_tv.q.weight"][0]]
        if self.text_bp is not None:
            self.text_bp.bind(self.arena, "text.")
            self.vis_bp.bind(self.arena, "vis.")
            bounds += [self.arena.offsets[extra[0][0][0]][0], self.arena.offsets["vis." + self.vis_bp.groups()[0][0][0]][0]]
            self._enc_dirty = False
        if cfg.grad_exchange == "factors":
            if cfg.grad_payload != "fp32":
                raise ValueError('grad_exchange="factors" moves fp32 factor panels: grad_payload must stay "fp32"')
            self.reducer = FactorExchange(self.arena.ensure_grad(), group=self.comm, bounds=bounds, force=force_exchange,
                                          linear_ranges=self._linear_grad_ranges())
        else:
            self.reducer = GradReducer(self.arena.ensure_grad(), group=self.comm, bounds=bounds,
                                       force=force_exchange, payload=cfg.grad_payload, algorithm=cfg.grad_exchange)
        if self.text_bp is not None and self.world > 1:
            # the encoders are built by the caller, before this constructor seeds anything: replicas may start from different RNG
            # states.  They only ever exchange gradients, so unequal masters would silently stay unequal.
            broadcast_from_rank0(self.arena.data, self.comm)
            self._sync_encoder_operands()
        self.optim = FusedAdamW(self.arena, lr=cfg.lr, weight_decay=cfg.weight_decay,
                                max_norm=cfg.grad_clip if cfg.grad_clip and cfg.grad_clip > 0 else 0.0,
                                seed=cfg.seed + 1000 * self.rank, grad_scale=self.reducer.grad_scale)
        if cfg.use_cosine:
            self.scheduler = CosineAnnealingLR(self.optim, T_max=cfg.epochs, eta_min=cfg.lr * cfg.min_lr_scale)
        else:
            self.scheduler = StepLR(self.optim, step_size=3, gamma=0.7)
        # criterion: plain mean CE (forensic_trainer.py:287) unless the integrated variant's options are set
        ce_w = (1.0, 1.0)
        if cfg.class_weighting:
            y = np.asarray(self.cache["labels"])
            pos, neg = float((y == 1).sum()), float((y == 0).sum())
            total = max(1.0, pos + neg)
            ce_w = (0.5 * total / max(1.0, neg), 0.5 * total / max(1.0, pos))
        self.text_encoder, self.visual_encoder = text_encoder, visual_encoder
        self.temporal_net = temporal_net    # optional TemporalSyncNet: temporal = align(text, visual) inside the step
        if cfg.encode_inline and (text_encoder is None or visual_encoder is None):
            raise ValueError("encode_inline=True needs text_encoder= and visual_encoder=")
        self.head = HeadStep(cfg, self.device, self.fusion, self.clf, self.optim, self.reducer, ce_w,
                             gnn_dims=416 if self.gnn_model is not None else None)
        self.pipe = EncoderPipeline(cfg, self.device, self.head, self.reducer, self.optim, text_encoder, visual_encoder, temporal_net)

        self.best_val_auc = -1.0
        self.no_improve = 0
        self.ckpt_path = os.path.join(cfg.out_dir, "best.pt")

    def _linear_grad_ranges(self):
        """[begin, end) float ranges of the gradient arena written by the head's Linear dW / db products (what the factor exchange
        forms instead of reducing); a range runs to the next tensor's offset, i.e. over the alignment gap behind it."""
        linear = {"clf.pre.0.weight", "clf.pre.0.bias", "clf.pre.3.weight", "clf.pre.3.bias",
                  "fusion.fuse_mlp.0.weight", "fusion.fuse_mlp.0.bias", "fusion.fuse_mlp.3.weight", "fusion.fuse_mlp.3.bias"}
        for k in self.arena.grad_keys:
            if k.startswith("fusion.") and (k.endswith("_proj.weight") or k.endswith("_proj.bias")) and "evidence" not in k and "semantic" not in k:
                linear.add(k)
            if k.startswith("fusion.attn_") and k.rsplit(".", 2)[-2] in ("q", "k", "v"):
                linear.add(k)
        order = sorted(self.arena.grad_keys, key=lambda k: self.arena.offsets[k][0])
        out = []
        for i, k in enumerate(order):
            if k in linear:
                lo = self.arena.offsets[k][0]
                hi = self.arena.offsets[order[i + 1]][0] if i + 1 < len(order) else self.arena.n_grad
                if out and out[-1][1] == lo:
                    out[-1] = (out[-1][0], hi)
                else:
                    out.append((lo, hi))
        n_lin = 2 + 2 + 9 + (5 if self.fusion.use_gnn else 4)          # pre.0/.3, fuse_mlp.0/.3, nine q/k/v, the input projections
        if len(linear) != 2 * n_lin:
            raise RuntimeError(f"factor exchange: expected {2 * n_lin} Linear weight / bias tensors in the head, found {len(linear)}")
        return out

    # ------------------------------------------------------------------ data
    def _build_dataloaders(self):
        tr = CachedTensorDataset(self.cache, self.tr_idx, self.device)
        va = CachedTensorDataset(self.cache, self.va_idx, self.device)
        te = CachedTensorDataset(self.cache, self.te_idx, self.device)
        bs = self.cfg.batch_size
        return (DeviceBatchLoader(tr, bs, shuffle=True, seed=self.cfg.seed, group=self.comm),
                DeviceBatchLoader(va, bs, shuffle=False, group=self.comm),
                DeviceBatchLoader(te, bs, shuffle=False, group=self.comm))

    def _dataset(self, split: str) -> CachedTensorDataset:
        return {"train": self.train_loader, "val": self.val_loader}.get(split, self.test_loader).dataset

    # ------------------------------------------------------------------ the step
    def _load_batch(self, b: dict, batch: Dict[str, torch.Tensor], split: str) -> None:
        """Fill the step's static buffers from a batch; features come from the cache or from the encoders."""
        if type(batch) is IndexedBatch and batch.ds.G is not None and not dict.__contains__(batch, "gnn_feat") and \
                not (self.cfg.encode_inline and "input_ids" in batch):
            self.head.gather_cached(b, batch.ds, dict.__getitem__(batch, "index"))
            return
        inline = self.cfg.encode_inline and "input_ids" in batch
        if inline:
            if self.text_bp is not None and self._enc_dirty:      # the masters moved since the frozen path last packed its operands
                for enc in (self.text_encoder, self.visual_encoder):
                    enc._packed = None
                    enc.weights_version += 1
                self._enc_dirty = False
            self.pipe.prefetch_features(batch, 0)
            for ev in self.pipe.feat_ready[0]:
                torch.cuda.current_stream(self.device).wait_event(ev)
            self.pipe.feat_ready[0] = None
            slot0 = self.head.bufs(_batch_size(batch), True, 0)       # the encoders write input slot 0 of the TRAIN buffers
            if slot0 is not b:                                        # (forward-only batches have buffers of their own)
                b["text"].copy_(slot0["text"])
                b["visual"].copy_(slot0["visual"])
        else:
            b["text"].copy_(batch["text_features"])
            b["visual"].copy_(batch["visual_features"])
        b["audio"].copy_(batch["audio_features"])
        if self.temporal_net is not None and inline:
            # the reference's cache builder draws align()'s dropout ONCE per sample (fakesv_dataset.py:176, module in train mode);
            # recomputed per step here, the mask is redrawn per TRAINING step, and validation / test use the deterministic
            # projection, so that epoch metrics, early stopping and best.pt selection do not depend on a mask draw
            self.temporal_net.align_batch(b["text"], b["visual"], out=b["temporal"], training=None if split == "train" else False)
        else:
            b["temporal"].copy_(batch["temporal_features"])
        b["aux"].copy_(batch["aux"])
        b["label"].copy_(batch["label"])
        ds = self._dataset(split)
        if "gnn_feat" in batch and batch["gnn_feat"] is not None:
            b["gnn"].copy_(batch["gnn_feat"])
        else:   # forensic_trainer.py:240-252: local index -> gnn_Z row
            idx = batch["index"]
            idx = idx.to(self.device) if isinstance(idx, torch.Tensor) else torch.as_tensor(idx, device=self.device)
            torch.index_select(ds.G, 0, idx, out=b["gnn"])

    # ---- the integrated variant's in-graph GNN (forensic_trainer_integrated.py:203-224)
    def _batch_ocr_sets(self, batch, split: str) -> list:
        if "ocr_sets" in batch and not isinstance(batch.get("ocr_sets"), torch.Tensor):
            return list(batch["ocr_sets"])
        idx = dict.__getitem__(batch, "index") if type(batch) is IndexedBatch else batch["index"]
        local = idx.cpu().tolist() if isinstance(idx, torch.Tensor) else list(idx)
        gi = self._dataset(split).global_idx.cpu()
        return [self.cache["ocr_sets"][int(gi[i])] for i in local]

    def _gnn_forward(self, b: dict, batch, B: int, split: str, train: bool) -> None:
        """gnn_feat of this mini-batch = GNNModel(node features, weighted OCR-Jaccard adjacency of the batch) -> b["gnn"].
        The overlap threshold anneals per epoch: max(0.05, thresh * 0.95^epoch) (:210-212)."""
        from .gnn_model import batch_node_features
        from .gcn import sets_to_csr
        batch_node_features(b["text"], b["audio"], b["visual"], b["temporal"], out=b["gnn_x"])
        thr = max(0.05, self.cfg.gnn_overlap_thresh * (0.95 ** self._epoch))
        offs, toks = sets_to_csr(self._batch_ocr_sets(batch, split))
        o = torch.from_numpy(offs).to(self.device)
        t = torch.from_numpy(toks if toks.size else np.zeros(1, dtype=np.int32)).to(self.device)
        L.check(L.lib().ufnd_ocr_adjacency_weighted(o.data_ptr(), t.data_ptr(), B, float(thr), b["gnn_adj"].data_ptr(), B,
                                                    L.stream_ptr(self.device)), "ufnd_ocr_adjacency_weighted")
        o.record_stream(torch.cuda.current_stream(self.device)); t.record_stream(torch.cuda.current_stream(self.device))
        self.gnn_model.train(train)
        self.gnn_model(b["gnn_x"], b["gnn_adj"], state=self.optim.state, out=b["gnn"])

    def _gnn_backward(self, b: dict, B: int) -> None:
        """d loss / d gnn_feat out of the fusion's workspace, then the GNN's parameter gradients (the arena's last range)."""
        L.check(L.lib().ufnd_fusion_gnn_input_grad(C.byref(b["dims"]), C.byref(self.fusion.param_table()), b["fws"].data_ptr(), B,
                                                   b["dgnn"].data_ptr(), self.optim.state.ptr, L.stream_ptr(self.device)),
                "ufnd_fusion_gnn_input_grad")
        self.gnn_model.backward(b["dgnn"])

    def _train_step_encoders(self, batch: Dict[str, torch.Tensor]) -> dict:
        """train_step with trainable encoders: encoder forwards that keep their activations -> head forward / loss / backward
        -> feature gradients -> encoder backwards (text, then visual; each closes its bucket of the exchange) -> one global-norm
        clip + AdamW over the joint arena -> the encoders' bf16 operands re-cast from the updated masters.  No lookahead: every
        step changes the encoders' weights.  The visual encoder's forward and backward run on a second stream beside the text
        encoder's (at 32 samples their launches are 150-430 workgroups each: neither chain fills the chip alone); the head sits
        between the two joins."""
        B = _batch_size(batch)
        b = self.head.bufs(B, True)
        if "dtext" not in b:
            b["dtext"] = torch.empty(B, 768, dtype=torch.float32, device=self.device)
            b["dvis"] = torch.empty(B, 512, dtype=torch.float32, device=self.device)
        main = torch.cuda.current_stream(self.device)
        if self._enc_side is None:
            self._enc_side = torch.cuda.Stream(device=self.device)
        side = self._enc_side
        side.wait_stream(main)
        with torch.cuda.stream(side):
            b["visual"].copy_(self.vis_bp.forward_train(batch["frames"]))
        b["text"].copy_(self.text_bp.forward_train(batch["input_ids"], batch["attention_mask"]))
        b["audio"].copy_(batch["audio_features"])
        main.wait_stream(side)
        if self.temporal_net is not None:       # (temporal = align(text, visual) is data, as in the reference's cache: no gradient through it)
            self.temporal_net.align_batch(b["text"], b["visual"], out=b["temporal"])
        else:
            b["temporal"].copy_(batch["temporal_features"])
        b["aux"].copy_(batch["aux"])
        b["label"].copy_(batch["label"])
        if "gnn_feat" in batch and batch["gnn_feat"] is not None:
            b["gnn"].copy_(batch["gnn_feat"])
        else:
            idx = batch["index"]
            torch.index_select(self._dataset("train").G, 0, idx.to(self.device), out=b["gnn"])

        def tail():
            self.head.feature_grads(b, B, b["dtext"], b["dvis"])
            side.wait_stream(main)
            with torch.cuda.stream(side):
                self.vis_bp.backward(b["dvis"])
            self.text_bp.backward(b["dtext"])
            if self.reducer.active:
                self.reducer.start(2)           # (the text encoder's bucket leaves while the visual backward is still running)
            main.wait_stream(side)
            if self.reducer.active:
                self.reducer.start(3)
        self.head.fwd_bwd(b, B, tail=tail)
        self.reducer.finish()
        self.optim.clip_and_step()
        side.wait_stream(main)
        with torch.cuda.stream(side):
            self.vis_bp.refresh_operands()
        self.text_bp.refresh_operands()
        main.wait_stream(side)
        self._enc_dirty = True
        return {"loss": self.optim.state.float_view("loss"), "probs": b["probs"], "y": b["label"],
                "forensic": b["forensic"], "logits": b["logits"]}

    def train_step(self, batch: Dict[str, torch.Tensor], split: str = "train") -> dict:
        """One iteration of the reference's train loop body (forensic_trainer.py:285-298):
        forward, CE, backward, [all-reduce], clip_grad_norm_, AdamW.step.  Returns device tensors."""
        if self.text_bp is not None and "input_ids" in batch:
            return self._train_step_encoders(batch)
        B = _batch_size(batch)
        b = self.head.bufs(B, True)
        self._load_batch(b, batch, split)
        if self.gnn_model is not None:
            self._gnn_forward(b, batch, B, split, True)
        self.head.fwd_bwd(b, B, post=(lambda: self._gnn_backward(b, B)) if self.gnn_model is not None else None)
        self.reducer.finish()
        self.optim.clip_and_step()
        return {"loss": self.optim.state.float_view("loss"), "probs": b["probs"], "y": b["label"],
                "forensic": b["forensic"], "logits": b["logits"]}

    # ---- encode_inline: the scheduler's entry points under the trainer's name (bench.py, tests, tools call them here)
    def prefetch_features(self, batch, slot: Optional[int] = None, inputs_ready=None, group: bool = False) -> None:
        self.pipe.prefetch_features(batch, slot, inputs_ready, group)

    def train_step_pipelined(self, batch, next_batch) -> dict:
        return self.pipe.train_step_pipelined(batch, next_batch)

    def train_group_pipelined(self, group, next_group, steps: Optional[int] = None, on_step=None) -> dict:
        return self.pipe.train_group_pipelined(group, next_group, steps, on_step)

    def measure_gemm_time(self, batch, steps: int = 3) -> Tuple[float, int]:
        out = self.pipe.measure_gemm_time(batch, steps)
        self.last_marker_us, self.last_raw_interval_us, self.last_gemm_by_shape = \
            self.pipe.last_marker_us, self.pipe.last_raw_interval_us, self.pipe.last_gemm_by_shape
        return out

    def _forward_batch(self, batch, split: str) -> Dict[str, torch.Tensor]:
        """Forward only (forensic_trainer.py:238-271); dropout follows the split like .train(is_train)."""
        B = _batch_size(batch)
        train = split == "train"
        b = self.head.bufs(B, False)
        self._load_batch(b, batch, split)
        if self.gnn_model is not None:
            self._gnn_forward(b, batch, B, split, train)
        self.head.enqueue_forward(b, B, train, False)
        f = b["forensic"]
        return {"logits": b["logits"], "probs": b["probs"], "y": b["label"],
                "forensic": {"emotion_intensity": f[0], "semantic_conflict": f[1], "temporal_delay": f[2]}}

    def _epoch_loop(self, loader, split: str) -> Tuple[float, Dict[str, float]]:
        is_train = split == "train"
        self.fusion.train(is_train)
        self.clf.train(is_train)
        losses: List[torch.Tensor] = []
        ys: List[torch.Tensor] = []
        p1s: List[torch.Tensor] = []
        fors: List[torch.Tensor] = []
        lookahead = (is_train and self.cfg.encode_inline and int(self.cfg.encoder_lookahead) > 1 and self.cfg.use_graph and self.text_bp is None and
                     self.gnn_model is None and isinstance(loader, DeviceBatchLoader) and loader.dataset.ids_tok is not None and
                     loader.dataset.frames is not None and loader.dataset.G is not None)
        if lookahead:
            def sink(b):
                losses.append(self.optim.state.float_view("loss").clone())
                ys.append(b["label"].clone())
                p1s.append(b["probs"][:, 1].clone())
                fors.append(b["forensic"].clone())
            self._train_epoch_lookahead(loader, sink)
        for batch in (() if lookahead else loader):
            if is_train:
                out = self.train_step(batch, split)
                f = out["forensic"]
                loss = out["loss"]
            else:
                o = self._forward_batch(batch, split)
                out = {"probs": o["probs"], "y": o["y"]}
                f = self.head.bufs(_batch_size(batch), False)["forensic"]
                loss = self.optim.state.float_view("loss")
            # device-side clones; ONE host sync per epoch instead of five per step
            losses.append(loss.clone())
            ys.append(out["y"].clone())
            p1s.append(out["probs"][:, 1].clone())
            fors.append(f.clone())
        if self.cfg.encode_inline:
            self.pipe.guard_flush()      # every pass evaluated the fold guard on the device; act on what has not been looked at yet
        if not losses and self.world == 1:
            return 0.0, aggregate_epoch_metrics(np.array([], dtype=int), np.array([], dtype=float))
        if not losses:      # an empty evaluation shard still takes part in the gather
            dev = self.device
            losses, ys = [torch.zeros((), device=dev)], [torch.zeros(0, dtype=torch.int64, device=dev)]
            p1s, fors = [torch.zeros(0, device=dev)], [torch.zeros(3, 0, device=dev)]
            n_loss = 0
        else:
            n_loss = len(losses)
        y_cat, p1_cat, f_cat, loss_mean_local = gather_epoch_outputs(torch.cat(ys), torch.cat(p1s), torch.cat(fors, dim=1),
                                                                      torch.stack(losses).sum(), n_loss, self.comm)
        loss_mean = float(loss_mean_local.cpu())
        f_np = f_cat.cpu().numpy()
        forensic = {"emotion_intensity": f_np[0], "semantic_conflict": f_np[1], "temporal_delay": f_np[2]}
        metrics = aggregate_epoch_metrics(y_true=y_cat.cpu().numpy(), y_score=p1_cat.cpu().numpy().astype(float),
                                          forensic=forensic, threshold=0.5, include_cm=False)
        return loss_mean, metrics

    def _train_epoch_lookahead(self, loader: "DeviceBatchLoader", sink) -> None:
        """One training epoch over a device-resident split of RAW inputs with the encoders inside the step: the loader's
        (sharded, shuffled) index list is cut into groups of encoder_lookahead batches -- the frozen encoders run once per
        group, the optimizer steps batch by batch (train_group_pipelined) -- then a shorter group of the remaining whole
        batches, then the ragged last batch (drop_last=False) as a plain train_step.  Same batches in the same order, same
        parameter updates as the plain loop, bit for bit (test)."""
        G, B, ds = int(self.cfg.encoder_lookahead), int(self.cfg.batch_size), loader.dataset
        idx = loader._indices().to(self.device)
        loader.epoch += 1
        n = int(idx.numel())
        cuts, s = [], 0
        while n - s >= B:
            g = min(G, (n - s) // B)
            cuts.append((s, s + g * B))
            s += g * B
        cur = None
        for i, (lo, hi) in enumerate(cuts):
            if cur is None:
                cur = self.pipe.group_inputs(ds, idx[lo:hi], self.pipe.gslot)
                self.pipe.prefetch_features(cur, group=True)
            nxt = self.pipe.group_inputs(ds, idx[cuts[i + 1][0]:cuts[i + 1][1]], self.pipe.gslot ^ 1) if i + 1 < len(cuts) else None
            self.pipe.train_group_pipelined(cur, nxt, on_step=sink)
            cur = nxt
        if s < n:
            out = self.train_step(IndexedBatch(ds, idx[s:]), "train")
            sink({"label": out["y"], "probs": out["probs"], "forensic": out["forensic"]})

    # ------------------------------------------------------------------ fit / test (forensic_trainer.py:332-396)
    def fit(self):
        self.no_improve = 0
        for epoch in range(1, self.cfg.epochs + 1):
            self._epoch = epoch - 1        # (zero-based: anneals the in-graph GNN's overlap threshold, forensic_trainer_integrated.py:259)
            tr_loss, tr_metrics = self._epoch_loop(self.train_loader, "train")
            va_loss, va_metrics = self._epoch_loop(self.val_loader, "val")
            self.scheduler.step()
            if self.rank == 0:
                print(f"[Epoch {epoch:02d}] train_loss={tr_loss:.4f} | ", end="")
                pretty_print("train", tr_metrics)
                print(f"           val_loss={va_loss:.4f} | ", end="")
                pretty_print("val", va_metrics)
            val_auc = float(va_metrics.get("auc", 0.5))
            improved = val_auc > (self.best_val_auc + 1e-4)
            if improved and self.cfg.save_best:
                self.best_val_auc = val_auc
                self.no_improve = 0
                # rank 0 writes the file atomically; every rank leaves save_checkpoint only when it is complete
                save_checkpoint(self._checkpoint_state(), self.ckpt_path, self.comm)
                if self.rank == 0:
                    print(f"  ↳ saved best checkpoint to {self.ckpt_path} (val_auc={self.best_val_auc:.3f})")
            else:
                self.no_improve += 1
                if self.no_improve >= self.cfg.early_stop_patience:
                    if self.rank == 0:
                        print(f"↳ Early stopping (no val AUC improvement for {self.cfg.early_stop_patience} epochs)")
                    break
        return self.best_val_auc

    def _checkpoint_state(self) -> dict:
        """The `best.pt` dict (forensic_trainer.py:355-360): fusion, clf, gnn, cfg -- and, when the encoders are trained with the
        head (train_encoders), their masters under keys of their own: the head of the best epoch is only meaningful together with
        the encoders of that epoch."""
        st = {"fusion": {k: v.cpu() for k, v in self.fusion.state_dict().items()},
              "clf": {k: v.cpu() for k, v in self.clf.state_dict().items()},
              "gnn": ({k: v.cpu() for k, v in self.gnn_model.state_dict().items()} if self.gnn_model is not None
                      else self.gnn.state_dict() if self.gnn is not None else None), "cfg": dict(self.cfg.__dict__)}
        if self.text_bp is not None:
            st["text_encoder"] = {k: v.cpu() for k, v in self.text_encoder.state_dict().items()}
            st["visual_encoder"] = {k: v.cpu() for k, v in self.visual_encoder.state_dict().items()}
        return st

    def _load_checkpoint(self) -> bool:
        """Rank 0 reads best.pt (weights_only); every rank then continues with rank 0's parameters (one broadcast of the arena:
        head and -- when they are trained -- encoder masters live in it).  Returns whether a file was read on rank 0."""
        found = False
        if self.rank == 0 and os.path.exists(self.ckpt_path):
            found = True
            ck = torch.load(self.ckpt_path, map_location="cpu", weights_only=True)
            self.fusion.load_state_dict(ck["fusion"])
            self.clf.load_state_dict(ck["clf"])
            if self.gnn_model is not None and ck.get("gnn") is not None:
                self.gnn_model.load_state_dict(ck["gnn"])
            if self.text_bp is not None and ck.get("text_encoder") is not None:      # (the masters are arena views: loaded in place)
                self.text_encoder.load_state_dict(ck["text_encoder"])
                self.visual_encoder.load_state_dict(ck["visual_encoder"])
        broadcast_from_rank0(self.arena.data, self.comm)
        if self.text_bp is not None:      # every operand copy derived from the masters is stale now
            self._sync_encoder_operands()
        return found

    def _sync_encoder_operands(self) -> None:
        for enc in (self.text_encoder, self.visual_encoder):
            enc._packed = None
            enc.weights_version += 1
        self.text_bp.refresh_operands()
        self.vis_bp.refresh_operands()
        self._enc_dirty = False

    def test(self) -> Dict[str, float]:
        # rank 0 reads the checkpoint; every rank then continues with rank 0's parameters (one broadcast of the arena),
        # so no rank ever evaluates its shard with a stale or half-written file
        self._load_checkpoint()
        self.fusion.eval()
        self.clf.eval()
        ts_loss, ts_metrics = self._epoch_loop(self.test_loader, "test")
        if self.rank == 0:
            print(f"[Test] loss={ts_loss:.4f} | ", end="")
            pretty_print("test", ts_metrics)
        return {"test_loss": ts_loss, "test_acc": ts_metrics.get("accuracy", 0.0), "test_auc": ts_metrics.get("auc", 0.5),
                "test_precision": ts_metrics.get("precision", 0.0), "test_recall": ts_metrics.get("recall", 0.0),
                "test_f1": ts_metrics.get("f1", 0.0), "test_cmcs": ts_metrics.get("cmcs", 0.0),
                "test_dfdr": ts_metrics.get("dfdr", 0.0)}

