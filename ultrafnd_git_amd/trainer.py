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
Out of scope: FakeSVRawDataset / build_gnn_cache_from_raw_dataset (dataset
preprocessing needing the FakeSV corpus): the trainer takes the cache dict they would have produced.  When that
cache carries `ocr_sets` instead of `gnn_Z`, the graph side of the reference's construction (OCR-Jaccard
adjacency, SimpleGCN, its two pre-training steps; forensic_trainer.py:184-224) runs here too (gcn.py).
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import Dict, Iterator, List, Optional, Tuple

import numpy as np
import torch

from . import _lib as L
from .arena import rehome
from .classifier import DeepTruthClassifier
from .dp import GradReducer, broadcast_from_rank0, gather_epoch_outputs, save_checkpoint, shard_indices, world_info
from .fusion import CrossModalTransformer
from .metrics import aggregate_epoch_metrics, pretty_print
from .optim import CosineAnnealingLR, FusedAdamW, StepLR

FEATS = ("text_features", "audio_features", "visual_features", "temporal_features")
_CACHE_KEY = {"text_features": "text", "audio_features": "audio", "visual_features": "visual",
              "temporal_features": "temporal"}


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


class CachedTensorDataset(torch.utils.data.Dataset):
    """Tensorised view of one split of the cache (forensic_trainer.py:60-83), device-resident."""

    def __init__(self, cache: Dict, indices: np.ndarray, device: Optional[torch.device] = None):
        indices = np.asarray(indices, dtype=np.int64)
        self.ids = cache["ids"][indices] if "ids" in cache else indices
        self.global_idx = torch.from_numpy(indices)

        def take(key, dtype):
            a = cache[key]
            t = a if isinstance(a, torch.Tensor) else torch.from_numpy(np.asarray(a))
            t = t[torch.from_numpy(indices).to(t.device)].to(dtype)
            return t.to(device) if device is not None else t
        self.T = take("text", torch.float32)
        self.A = take("audio", torch.float32)
        self.V = take("visual", torch.float32)
        self.U = take("temporal", torch.float32)
        self.AUX = take("aux", torch.float32)
        self.y = take("labels", torch.int64)
        self.G = take("gnn_Z", torch.float32) if "gnn_Z" in cache else None
        # raw inputs for encode_inline
        self.ids_tok = take("input_ids", torch.int64) if "input_ids" in cache else None
        self.mask_tok = take("attention_mask", torch.int32) if "attention_mask" in cache else None
        self.frames = take("frames", torch.float32) if "frames" in cache else None
        if device is not None:
            self.global_idx = self.global_idx.to(device)

    def __len__(self):
        return self.T.shape[0]

    def __getitem__(self, i):
        return {"text_features": self.T[i], "audio_features": self.A[i], "visual_features": self.V[i],
                "temporal_features": self.U[i], "aux": self.AUX[i], "label": self.y[i], "index": i}

    def gather(self, idx: torch.Tensor) -> Dict[str, torch.Tensor]:
        """Default-collate equivalent for a whole index vector, as one gather per tensor."""
        b = {"text_features": self.T[idx], "audio_features": self.A[idx], "visual_features": self.V[idx],
             "temporal_features": self.U[idx], "aux": self.AUX[idx], "label": self.y[idx], "index": idx}
        if self.ids_tok is not None:
            b["input_ids"], b["attention_mask"] = self.ids_tok[idx], self.mask_tok[idx]
        if self.frames is not None:
            b["frames"] = self.frames[idx]
        return b


class IndexedBatch(dict):
    """A batch of a device-resident split, named by its row indices (`batch["index"]`).  It is the dict the reference's
    default collate would build (same keys); a tensor is gathered when it is first asked for.  The trainer never asks:
    it sends the indices to `ufnd_gather_rows`, which fills the step's static buffers in one launch."""
    _SRC = {"text_features": "T", "audio_features": "A", "visual_features": "V", "temporal_features": "U", "aux": "AUX",
            "label": "y", "input_ids": "ids_tok", "attention_mask": "mask_tok", "frames": "frames"}

    def __init__(self, ds: "CachedTensorDataset", idx: torch.Tensor):
        super().__init__(index=idx)
        self.ds = ds

    def _lazy(self, k) -> bool:
        return k in self._SRC and getattr(self.ds, self._SRC[k]) is not None

    def __missing__(self, k):
        if not self._lazy(k):
            raise KeyError(k)
        v = getattr(self.ds, self._SRC[k])[dict.__getitem__(self, "index")]
        self[k] = v
        return v

    def __contains__(self, k):
        return dict.__contains__(self, k) or self._lazy(k)

    def get(self, k, default=None):
        return self[k] if k in self else default

    def keys(self):
        return [k for k in self._SRC if self._lazy(k)] + [k for k in dict.keys(self) if k not in self._SRC]

    def __iter__(self):
        return iter(self.keys())

    def __len__(self):
        return len(self.keys())

    def items(self):
        return [(k, self[k]) for k in self.keys()]

    def values(self):
        return [self[k] for k in self.keys()]


def _batch_size(batch) -> int:
    if type(batch) is IndexedBatch:
        return int(dict.__getitem__(batch, "index").numel())
    return int(batch["label"].shape[0])


class DeviceBatchLoader:
    """DataLoader(dataset, batch_size, shuffle, drop_last=False) over a device-resident split,
    sharded across data-parallel ranks (DistributedSampler semantics)."""

    def __init__(self, dataset: CachedTensorDataset, batch_size: int, shuffle: bool, seed: int = 0, group=None, pad: Optional[bool] = None):
        self.dataset, self.batch_size, self.shuffle, self.seed, self.group = dataset, int(batch_size), shuffle, seed, group
        # training shards are wrapped to equal length (every rank takes the same number of steps: one collective per
        # step); evaluation shards are not, so that no sample enters the epoch metrics twice
        self.pad = shuffle if pad is None else pad
        self.epoch = 0

    def _indices(self) -> torch.Tensor:
        n = len(self.dataset)
        world, rank = world_info(self.group)
        perm = None
        if self.shuffle:
            g = torch.Generator().manual_seed(self.seed + self.epoch)
            perm = torch.randperm(n, generator=g)
        return shard_indices(n, world, rank, perm, pad=self.pad)

    def __len__(self):
        n = self._indices().numel()
        return (n + self.batch_size - 1) // self.batch_size

    def __iter__(self) -> Iterator[Dict[str, torch.Tensor]]:
        idx = self._indices().to(self.dataset.T.device)
        self.epoch += 1
        for s in range(0, idx.numel(), self.batch_size):
            yield IndexedBatch(self.dataset, idx[s:s + self.batch_size])


class ForensicTrainer:
    def __init__(self, cfg: TrainConfig, cache: Optional[Dict] = None, text_encoder=None, visual_encoder=None,
                 group=None, temporal_net=None):
        self.cfg = cfg
        os.makedirs(cfg.out_dir, exist_ok=True)
        self.device = torch.device(cfg.device)
        if self.device.type != "cuda":
            raise L.UltrafndHipError("ForensicTrainer needs a HIP device (TrainConfig.device='cuda'); no CPU path")
        self.dtype = torch.float32
        self.group = group
        self.world, self.rank = world_info(group)
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
            raise ValueError("use_gnn=False: the reference's fusion head is sized for the concat WITH the GNN slot and "
                             "fails in forward without gnn_feat (cross_modal_transformer.py:184-197); not supported")
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
        # one flat arena for both modules: clf first (its gradients are ready first in backward)
        if cfg.gnn_in_graph:
            from .gnn_model import GNNModel
            self.gnn_model = GNNModel(in_dim=416, hid=256, out_dim=cfg.gnn_dim, dropout=0.1).to(self.device)   # :136
            self.arena = rehome([self.clf, self.fusion, self.gnn_model], ["clf.", "fusion.", "gnn."])        # its gradients are ready last
            self._epoch = 0
        else:
            self.arena = rehome([self.clf, self.fusion], ["clf.", "fusion."])
        # two buckets in gradient-ready order: [classifier | fuse_mlp] is complete after the first phase of backward
        self.reducer = GradReducer(self.arena.ensure_grad(), group=group, bounds=[self.arena.offsets["fusion.attn_tv.q.weight"][0]])
        self.optim = FusedAdamW(self.arena, lr=cfg.lr, weight_decay=cfg.weight_decay,
                                max_norm=cfg.grad_clip if cfg.grad_clip and cfg.grad_clip > 0 else 0.0,
                                seed=cfg.seed + 1000 * self.rank, grad_scale=self.reducer.grad_scale)
        if cfg.use_cosine:
            self.scheduler = CosineAnnealingLR(self.optim, T_max=cfg.epochs, eta_min=cfg.lr * cfg.min_lr_scale)
        else:
            self.scheduler = StepLR(self.optim, step_size=3, gamma=0.7)
        # criterion: plain mean CE (forensic_trainer.py:287) unless the integrated variant's options are set
        self._ce_w = (1.0, 1.0)
        if cfg.class_weighting:
            y = np.asarray(self.cache["labels"])
            pos, neg = float((y == 1).sum()), float((y == 0).sum())
            total = max(1.0, pos + neg)
            self._ce_w = (0.5 * total / max(1.0, neg), 0.5 * total / max(1.0, pos))
        self.text_encoder, self.visual_encoder = text_encoder, visual_encoder
        self.temporal_net = temporal_net    # optional TemporalSyncNet: temporal = align(text, visual) inside the step
        if cfg.encode_inline and (text_encoder is None or visual_encoder is None):
            raise ValueError("encode_inline=True needs text_encoder= and visual_encoder=")

        self.best_val_auc = -1.0
        self.no_improve = 0
        self.ckpt_path = os.path.join(cfg.out_dir, "best.pt")
        self._step_bufs: Dict[Tuple[int, bool, int], dict] = {}
        self._enc_bufs: Dict[Tuple[int, int, int], dict] = {}
        self._enc_streams = None
        self._hp_stream: Optional[torch.cuda.Stream] = None
        self._dw_stream: Optional[torch.cuda.Stream] = None
        # the head's fwd/bwd as a graph (default with use_graph) or eager with the dW side stream
        self._head_graph = cfg.use_graph and os.environ.get("UFND_HEAD_GRAPH", "1") != "0"
        self._slot = 0
        self._feat_ready = [None, None]
        self._slot_free = [None, None]
        self._gslot = 0                      # encoder lookahead (train_group_pipelined): group feature buffers, two slots
        self._grp_bufs: Dict[Tuple[int, int], dict] = {}
        self._grp_in: Dict[Tuple[int, int], dict] = {}     # persistent input buffers of the epoch loop's lookahead groups
        self._grp_ready = [None, None]
        self._grp_free = [None, None]

    # ------------------------------------------------------------------ data
    def _build_dataloaders(self):
        tr = CachedTensorDataset(self.cache, self.tr_idx, self.device)
        va = CachedTensorDataset(self.cache, self.va_idx, self.device)
        te = CachedTensorDataset(self.cache, self.te_idx, self.device)
        bs = self.cfg.batch_size
        return (DeviceBatchLoader(tr, bs, shuffle=True, seed=self.cfg.seed, group=self.group),
                DeviceBatchLoader(va, bs, shuffle=False, group=self.group),
                DeviceBatchLoader(te, bs, shuffle=False, group=self.group))

    def _dataset(self, split: str) -> CachedTensorDataset:
        return {"train": self.train_loader, "val": self.val_loader}.get(split, self.test_loader).dataset

    # ------------------------------------------------------------------ the step
    def _bufs(self, B: int, train: bool, slot: int = 0) -> dict:
        """Static buffers of one batch size (graph replay needs fixed addresses).  Two slots exist so
        that the encoders can fill step i+1's inputs while step i's backward still reads its own."""
        key = (B, train, slot)
        if key not in self._step_bufs:
            dev, f32 = self.device, torch.float32
            dims = self.clf.dims()
            dims.fusion_dropout = self.fusion.dropout
            n_f = L.lib().ufnd_fusion_workspace_floats(C.byref(dims), B)
            n_c = L.lib().ufnd_clf_workspace_floats(C.byref(dims), B)
            fws = torch.empty(n_f, dtype=f32, device=dev)
            cws = torch.empty(n_c, dtype=f32, device=dev)
            ld = C.c_int(0)
            xin = L.lib().ufnd_clf_input_panel(C.byref(dims), cws.data_ptr(), B, C.byref(ld))
            self._step_bufs[key] = {
                "dims": dims, "fws": fws, "cws": cws, "xin": xin, "ldx": ld.value,
                "text": torch.empty(B, 768, dtype=f32, device=dev), "audio": torch.empty(B, 128, dtype=f32, device=dev),
                "visual": torch.empty(B, 512, dtype=f32, device=dev), "temporal": torch.empty(B, 256, dtype=f32, device=dev),
                "gnn": torch.empty(B, self.fusion.gnn_dim, dtype=f32, device=dev),
                "aux": torch.empty(B, 2, dtype=f32, device=dev), "label": torch.empty(B, dtype=torch.int64, device=dev),
                "logits": torch.empty(B, 2, dtype=f32, device=dev), "probs": torch.empty(B, 2, dtype=f32, device=dev),
                "forensic": torch.empty(3, B, dtype=f32, device=dev), "dlogits": torch.empty(B, 2, dtype=f32, device=dev),
                "dfused": torch.empty(B, self.fusion.hidden, dtype=f32, device=dev), "graph": None}
            if self.gnn_model is not None:
                self._step_bufs[key].update({"gnn_x": torch.empty(B, 416, dtype=f32, device=dev), "gnn_adj": torch.zeros(B, B, dtype=f32, device=dev),
                                             "dgnn": torch.empty(B, self.fusion.gnn_dim, dtype=f32, device=dev)})
        return self._step_bufs[key]

    def _load_batch(self, b: dict, batch: Dict[str, torch.Tensor], split: str) -> None:
        """Copy a batch into the static buffers; features come from the cache or from the encoders."""
        if type(batch) is IndexedBatch and batch.ds.G is not None and not dict.__contains__(batch, "gnn_feat") and \
                not (self.cfg.encode_inline and "input_ids" in batch):
            # cached features named by row index: one launch gathers every tensor (and gnn_Z, forensic_trainer.py:240-252)
            ds, idx = batch.ds, dict.__getitem__(batch, "index")
            idx = idx.to(self.device, torch.int64).contiguous()
            pairs = [(ds.T, b["text"]), (ds.A, b["audio"]), (ds.V, b["visual"]), (ds.U, b["temporal"]), (ds.AUX, b["aux"]),
                     (ds.y, b["label"]), (ds.G, b["gnn"])]
            items = (L.GatherItem * len(pairs))()
            for it, (src, dst) in zip(items, pairs):
                rb = src[0].numel() * src.element_size()
                if not src.is_contiguous() or src.dtype != dst.dtype or rb != dst[0].numel() * dst.element_size() or src.device != dst.device:
                    raise RuntimeError(f"cached tensor {tuple(src.shape)} {src.dtype} does not match its batch buffer {tuple(dst.shape)} {dst.dtype}")
                it.src, it.dst, it.row_bytes, it.src_rows = src.data_ptr(), dst.data_ptr(), rb, src.shape[0]
            L.check(L.lib().ufnd_gather_rows(idx.data_ptr(), idx.numel(), items, len(pairs), L.stream_ptr(self.device)), "ufnd_gather_rows")
            return
        if self.cfg.encode_inline and "input_ids" in batch:
            self.prefetch_features(batch, 0)
            for ev in self._feat_ready[0]:
                torch.cuda.current_stream(self.device).wait_event(ev)
            self._feat_ready[0] = None
        else:
            b["text"].copy_(batch["text_features"])
            b["visual"].copy_(batch["visual_features"])
        b["audio"].copy_(batch["audio_features"])
        if self.temporal_net is not None and self.cfg.encode_inline and "input_ids" in batch:
            b["temporal"].copy_(self.temporal_net.align_batch(b["text"], b["visual"]))
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

    def _enqueue_forward(self, b: dict, B: int, train: bool, with_loss_grad: bool) -> None:
        lib, s, st = L.lib(), L.stream_ptr(self.device), self.optim.state.ptr
        d = b["dims"]
        L.check(lib.ufnd_fusion_forward(C.byref(d), C.byref(self.fusion.param_table()), b["text"].data_ptr(),
                                        b["audio"].data_ptr(), b["visual"].data_ptr(), b["temporal"].data_ptr(),
                                        b["gnn"].data_ptr(), B, int(train), b["fws"].data_ptr(), b["xin"], b["ldx"], None,
                                        b["forensic"].data_ptr(), st, s), "ufnd_fusion_forward")
        L.check(lib.ufnd_classifier_forward(C.byref(d), C.byref(self.clf.param_table()), b["xin"], b["ldx"],
                                            b["aux"].data_ptr(), B, int(train), b["cws"].data_ptr(), b["logits"].data_ptr(),
                                            b["probs"].data_ptr(), st, s), "ufnd_classifier_forward")
        if self.cfg.label_smoothing > 0.0 or self.cfg.class_weighting:
            L.check(lib.ufnd_softmax_ce_weighted(b["logits"].data_ptr(), b["label"].data_ptr(), B, self._ce_w[0], self._ce_w[1],
                                                 float(self.cfg.label_smoothing), None,
                                                 b["dlogits"].data_ptr() if with_loss_grad else None, st, s), "ufnd_softmax_ce_weighted")
        else:
            L.check(lib.ufnd_softmax_ce(b["logits"].data_ptr(), b["label"].data_ptr(), B, None,
                                        b["dlogits"].data_ptr() if with_loss_grad else None, st, s), "ufnd_softmax_ce")

    def _enqueue_backward(self, b: dict, B: int, part: int = 0) -> None:
        """part 0: the whole backward; 1: classifier backward + the fuse_mlp phase of the fusion backward (bucket 0 of
        the gradient exchange is complete afterwards); 2: the rest of the fusion backward."""
        lib, s, st = L.lib(), L.stream_ptr(self.device), self.optim.state.ptr
        d = b["dims"]
        # Eager launches: dW / parameter-gradient kernels run beside the dX chain on a second stream (joined
        # at the end).  Inside a captured hipGraph the fork/join costs more than it hides (ROCm 7 replays
        # multi-branch graphs almost serially -- measured), so the graph keeps one stream.
        side = None
        if not self._head_graph:
            if self._dw_stream is None:
                self._dw_stream = torch.cuda.Stream(device=self.device)
            side = self._dw_stream.cuda_stream
        if part != 2:
            L.check(lib.ufnd_classifier_backward(C.byref(d), C.byref(self.clf.param_table()), C.byref(self.clf.grad_table()), B, 1,
                                                 b["cws"].data_ptr(), b["dlogits"].data_ptr(), b["dfused"].data_ptr(),
                                                 self.fusion.hidden, st, s, side, 0), "ufnd_classifier_backward")
        L.check(lib.ufnd_fusion_backward_phase(C.byref(d), C.byref(self.fusion.param_table()), C.byref(self.fusion.grad_table()),
                                               b["text"].data_ptr(), b["audio"].data_ptr(), b["visual"].data_ptr(),
                                               b["temporal"].data_ptr(), b["gnn"].data_ptr(), B, 1, b["fws"].data_ptr(),
                                               b["dfused"].data_ptr(), self.fusion.hidden, None, st, s, side, 1,
                                               (L.BWD_ALL, L.BWD_FUSE_MLP, L.BWD_REST)[part]), "ufnd_fusion_backward_phase")

    def _fwd_bwd(self, b: dict, B: int) -> None:
        """fusion fwd -> clf fwd -> CE -> clf bwd -> fusion bwd, eager or replayed from a hipGraph.  With a gradient
        exchange (data parallel) the backward is cut after the fuse_mlp phase: bucket 0 of the exchange starts there and
        runs beside the rest of backward, bucket 1 follows it (dp.py); the caller's reducer.finish() joins both."""
        dp = self.reducer.active

        def first():
            self._enqueue_forward(b, B, True, True)
            self._enqueue_backward(b, B, 1 if dp else 0)

        def second():
            self._enqueue_backward(b, B, 2)
        post = (lambda: self._gnn_backward(b, B)) if self.gnn_model is not None else (lambda: None)
        if not self._head_graph:
            first()
            if dp:
                self.reducer.start(0)
                second()
                post()
                self.reducer.start(1)
            else:
                post()
            return
        key = "graph_dp" if dp else "graph"
        if b.get(key) is None:
            side = torch.cuda.Stream(device=self.device)
            side.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(side):          # warm-up outside capture (lazy module loads etc.)
                first()
                if dp:
                    second()
            torch.cuda.current_stream(self.device).wait_stream(side)
            graphs = []
            for fn in ((first, second) if dp else (first,)):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    fn()
                graphs.append(g)
            b[key] = graphs
        b[key][0].replay()
        if dp:
            self.reducer.start(0)
            b[key][1].replay()
            post()                      # (eager, behind the graph: the GNN's gradients close the arena's last bucket)
            self.reducer.start(1)
        else:
            post()

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

    def train_step(self, batch: Dict[str, torch.Tensor], split: str = "train") -> dict:
        """One iteration of the reference's train loop body (forensic_trainer.py:285-298):
        forward, CE, backward, [all-reduce], clip_grad_norm_, AdamW.step.  Returns device tensors."""
        B = _batch_size(batch)
        b = self._bufs(B, True)
        self._load_batch(b, batch, split)
        if self.gnn_model is not None:
            self._gnn_forward(b, batch, B, split, True)
        self._fwd_bwd(b, B)
        self.reducer.finish()
        self.optim.clip_and_step()
        return {"loss": self.optim.state.float_view("loss"), "probs": b["probs"], "y": b["label"],
                "forensic": b["forensic"], "logits": b["logits"]}

    # ---- software-pipelined variant for encode_inline: all-reduce(i) overlaps encoders(i+1)
    # ---- encoders: two independent chains on two streams, each captured as its own hipGraph
    def _enc_state(self, B: int, Lq: int, Fr: int, S: int) -> dict:
        key = (B, Lq, Fr)
        if key not in self._enc_bufs:
            dev = self.device
            self._enc_bufs[key] = {
                "ids": torch.empty(B, Lq, dtype=torch.int64, device=dev), "mask": torch.empty(B, Lq, dtype=torch.int32, device=dev),
                "frames": torch.empty(B, Fr, 3, S, S, dtype=torch.float32, device=dev),
                "text_out": torch.empty(B, 768, dtype=torch.float32, device=dev),
                "vis_out": torch.empty(B, 512, dtype=torch.float32, device=dev), "g_text": None, "g_vis": None}
        if self._enc_streams is None:
            split = self.cfg.cu_split
            if os.environ.get("UFND_CU_SPLIT"):          # experiments: "192,64" / "0" = off
                v = [int(x) for x in os.environ["UFND_CU_SPLIT"].split(",")]
                split = tuple(v) if len(v) == 2 and v[0] > 0 else None
            if split:
                from .streams import MaskedStream, partition_bits
                bt, bv = partition_bits(split)
                self._enc_streams = (MaskedStream(self.device, bt), MaskedStream(self.device, bv))
            else:
                # (default priorities: a high-priority text stream bought 0.5 % at one GPU and cost 2x under data
                #  parallelism, where the all-reduce of step i must get CUs while the encoders of step i+1 run)
                self._enc_streams = (torch.cuda.Stream(device=self.device), torch.cuda.Stream(device=self.device))
        return self._enc_bufs[key]

    def _encode_text(self, e: dict) -> None:
        e["text_out"].copy_(self.text_encoder(e["ids"], e["mask"]))

    def _encode_vis(self, e: dict) -> None:
        e["vis_out"].copy_(self.visual_encoder(e["frames"]))

    _PINNED_GRAPHS = 8      # captured encoder graphs per encoder that read the CALLER's input buffers in place

    def _encode_pinned(self, e: dict, which: str, enc, inputs: tuple, dtypes: tuple, out: torch.Tensor) -> bool:
        """Encode straight from the caller's buffers into the step's slot buffer, from a graph captured for exactly these
        addresses: a loader that rotates a few persistent device buffers (bench.py: four) then pays NO restaging copy (the frames
        alone were a 19 MB device-to-device copy per step) and no copy of the features.  False when the inputs do not qualify
        or the cache is full: the caller then stages them into the encoder's static buffers (one graph for any address)."""
        if not self.cfg.use_graph:
            return False
        for t, dt in zip(inputs, dtypes):
            if not (isinstance(t, torch.Tensor) and t.device == self.device and t.dtype == dt and t.is_contiguous()):
                return False
        key = (which,) + tuple(t.data_ptr() for t in inputs) + (out.data_ptr(),)
        cache = e.setdefault("pinned", {})
        ent = cache.get(key)
        if ent is None:
            if sum(1 for k in cache if k[0] == which) >= self._PINNED_GRAPHS:
                return False
            cur = torch.cuda.current_stream(self.device)
            out.copy_(enc(*inputs))                     # warm-up: packs weights, allocates buffers
            cur.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=cur, capture_error_mode="thread_local"):
                out.copy_(enc(*inputs))
            ent = cache[key] = (g, inputs)              # (the graph names these buffers: keep them alive)
        ent[0].replay()
        return True

    def _replay_or_capture(self, e: dict, which: str, fn) -> None:
        """Run `fn(e)` on the current stream: eagerly, or (use_graph) from a graph captured once."""
        if not self.cfg.use_graph:
            fn(e)
            return
        if e[which] is None:
            fn(e)                                   # warm-up: packs weights, allocates buffers
            torch.cuda.current_stream(self.device).synchronize()
            g = torch.cuda.CUDAGraph()
            # thread_local: the RCCL watchdog thread may poll events while this thread captures
            with torch.cuda.graph(g, stream=torch.cuda.current_stream(self.device), capture_error_mode="thread_local"):
                fn(e)
            e[which] = g
        e[which].replay()

    def prefetch_features(self, batch: Dict[str, torch.Tensor], slot: Optional[int] = None,
                          inputs_ready: Optional[torch.cuda.Event] = None, group: bool = False) -> None:
        """Encode `batch` on the two encoder streams (text || visual) into input slot `slot` of the step
        buffers.  Runs concurrently with whatever the compute stream is doing (the fusion head of the
        previous batch); `_feat_ready[slot]` is recorded when both features have landed.
        group=True: `batch` is a lookahead group (G x batch_size rows, train_group_pipelined): the features go to the group
        feature buffers of group slot `slot`, and `_grp_ready[slot]` is recorded."""
        slot = (self._gslot if group else self._slot) if slot is None else slot
        ids, frames = batch["input_ids"], batch["frames"]
        if frames.dim() == 4:
            frames = frames[:, None]
        B, Lq, Fr = int(ids.shape[0]), int(ids.shape[1]), int(frames.shape[1])
        b = self._group_bufs(B, slot) if group else self._bufs(B, True, slot)
        free = self._grp_free if group else self._slot_free
        e = self._enc_state(B, Lq, Fr, int(frames.shape[-1]))
        main = torch.cuda.current_stream(self.device)
        st, sv = self._enc_streams
        for strm in (st, sv):
            # the batch tensors were produced on the compute stream: wait for THEM, not for later work
            if inputs_ready is not None:
                strm.wait_event(inputs_ready)
            else:
                strm.wait_stream(main)
            if free[slot] is not None:
                strm.wait_event(free[slot])   # the head(s) that last read this slot are done with it
        with torch.cuda.stream(st):
            self._mark("text0", st)
            mask = batch["attention_mask"]
            e["last_text"] = (ids, mask)
            if not self._encode_pinned(e, "text", self.text_encoder, (ids, mask), (torch.int64, torch.int32), b["text"]):
                e["ids"].copy_(ids)
                e["mask"].copy_(mask)
                e["last_text"] = (e["ids"], e["mask"])
                self._replay_or_capture(e, "g_text", self._encode_text)
                b["text"].copy_(e["text_out"])
            ev_t = torch.cuda.Event()
            ev_t.record(st)
            self._mark("text1", st)
        with torch.cuda.stream(sv):
            self._mark("vis0", sv)
            e["last_vis"] = (frames,)
            if not self._encode_pinned(e, "vis", self.visual_encoder, (frames,), (torch.float32,), b["visual"]):
                e["frames"].copy_(frames)
                e["last_vis"] = (e["frames"],)
                self._replay_or_capture(e, "g_vis", self._encode_vis)
                b["visual"].copy_(e["vis_out"])
            ev_v = torch.cuda.Event()
            ev_v.record(sv)
            self._mark("vis1", sv)
        for t in (ids, batch["attention_mask"], frames):
            t.record_stream(st)
            t.record_stream(sv)
        if group:
            self._grp_ready[slot] = (ev_t, ev_v)
        else:
            self._feat_ready[slot] = (ev_t, ev_v)

    # ---- encoder lookahead: the frozen encoders run over G consecutive batches per pass, the head steps batch by batch
    def _group_bufs(self, GB: int, gslot: int) -> dict:
        key = (GB, gslot)
        if key not in self._grp_bufs:
            self._grp_bufs[key] = {"text": torch.empty(GB, 768, dtype=torch.float32, device=self.device),
                                   "visual": torch.empty(GB, 512, dtype=torch.float32, device=self.device)}
        return self._grp_bufs[key]

    def train_group_pipelined(self, group: Dict[str, torch.Tensor], next_group: Optional[Dict[str, torch.Tensor]],
                              steps: Optional[int] = None, on_step=None) -> dict:
        """`steps` (default: all G) optimizer steps over a lookahead group: a dict of raw inputs with G x batch_size rows whose
        features prefetch_features(group=True) has started.  The encoders are frozen (as in the reference, where the features
        are a precomputed cache), so encoding G batches in ONE pass changes no value -- a row's features do not depend on the
        batch it is encoded in, bit for bit (test) -- while every GEMM launch gets G times the rows: fewer, larger launches.
        The head, the loss, the gradient exchange, the clip and AdamW run per batch of batch_size rows, in order, exactly as
        train_step does: G optimizer steps.  The next group's encoders are enqueued behind the first head.
        `on_step(b)` (optional) is called after every optimizer step with the step's static buffers (the epoch loop clones
        what its metrics need)."""
        if self._hp_stream is None:
            self._hp_stream = torch.cuda.Stream(device=self.device, priority=-1)
        caller = torch.cuda.current_stream(self.device)
        self._hp_stream.wait_stream(caller)
        with torch.cuda.stream(self._hp_stream):
            out = self._train_group_pipelined(group, next_group, steps, on_step)
        caller.wait_stream(self._hp_stream)
        return out

    def _train_group_pipelined(self, group, next_group, steps, on_step=None) -> dict:
        B = int(self.cfg.batch_size)
        GB = int(group["input_ids"].shape[0])
        if GB % B:
            raise RuntimeError(f"lookahead group of {GB} rows is not a multiple of batch_size {B}")
        G = GB // B
        steps = G if steps is None else int(steps)
        gslot = self._gslot
        grp = self._group_bufs(GB, gslot)
        main = torch.cuda.current_stream(self.device)
        if self._grp_ready[gslot] is None:
            raise RuntimeError("train_group_pipelined: call prefetch_features(group, group=True) for the first group")
        inputs_ready = torch.cuda.Event()
        inputs_ready.record(main)
        self._mark("step0", main)
        for ev in self._grp_ready[gslot]:
            main.wait_event(ev)
        self._grp_ready[gslot] = None
        self._mark("head0", main)
        b = self._bufs(B, True, 0)
        losses = []
        started_next = next_group is None
        for k in range(steps):
            self._stage_group_rows(b, group, grp, k, B)
            if self.temporal_net is not None:
                self.temporal_net.align_batch(b["text"], b["visual"], out=b["temporal"])
            else:
                b["temporal"].copy_(group["temporal_features"][k * B:(k + 1) * B])
            if not started_next and self.reducer.active:      # (before the collectives: see _train_step_pipelined)
                self.prefetch_features(next_group, gslot ^ 1, inputs_ready, group=True)
                started_next = True
            self._fwd_bwd(b, B)
            if not started_next:
                self.prefetch_features(next_group, gslot ^ 1, inputs_ready, group=True)
                started_next = True
            self.reducer.finish()
            self.optim.clip_and_step()
            losses.append(self.optim.state.float_view("loss").clone())
            if on_step is not None:
                on_step(b)
        done = torch.cuda.Event()
        done.record(main)
        self._grp_free[gslot] = done
        self._mark("opt1", main)
        self._gslot ^= 1
        return {"loss": self.optim.state.float_view("loss"), "losses": losses, "probs": b["probs"], "y": b["label"],
                "forensic": b["forensic"], "logits": b["logits"]}

    def _stage_group_rows(self, b: dict, group, grp: dict, k: int, B: int) -> None:
        """Rows [kB, (k+1)B) of the group's features and small inputs into the step's static buffers: ONE gather launch."""
        pairs = [(grp["text"], b["text"]), (grp["visual"], b["visual"]), (group["audio_features"], b["audio"]), (group["aux"], b["aux"]),
                 (group["label"], b["label"]), (group["gnn_feat"], b["gnn"])]
        for src, dst in pairs:
            if not (src.device == dst.device and src.dtype == dst.dtype and src.is_contiguous() and tuple(src.shape[1:]) == tuple(dst.shape[1:]) and
                    (src[0].numel() * src.element_size()) % 8 == 0):
                raise RuntimeError(f"lookahead group tensor {tuple(src.shape)} {src.dtype} does not match its step buffer {tuple(dst.shape)} {dst.dtype}")
        if self._iota is None or self._iota.numel() < B:
            self._iota = torch.arange(max(B, 256), dtype=torch.int64, device=self.device)
        items = (L.GatherItem * len(pairs))()
        for it, (src, dst) in zip(items, pairs):
            rb = src[0].numel() * src.element_size()
            it.src, it.dst, it.row_bytes, it.src_rows = src.data_ptr() + k * B * rb, dst.data_ptr(), rb, B
        L.check(L.lib().ufnd_gather_rows(self._iota.data_ptr(), B, items, len(pairs), L.stream_ptr(self.device)), "ufnd_gather_rows")

    def _stage_small_inputs(self, b: dict, batch, B: int) -> None:
        """audio / aux / label / gnn rows of a raw batch into the step's static buffers: ONE ufnd_gather_rows launch (identity
        index) instead of four copy kernels on the head -> exchange -> optimizer chain; torch copies when a tensor does not
        have the buffer's dtype / layout."""
        pairs = [(batch["audio_features"], b["audio"]), (batch["aux"], b["aux"]), (batch["label"], b["label"]), (batch["gnn_feat"], b["gnn"])]
        ok = all(isinstance(src, torch.Tensor) and src.device == dst.device and src.dtype == dst.dtype and src.is_contiguous() and
                 tuple(src.shape) == tuple(dst.shape) and (src[0].numel() * src.element_size()) % 8 == 0 for src, dst in pairs)
        if not ok:
            for src, dst in pairs:
                dst.copy_(src)
            return
        if self._iota is None or self._iota.numel() < B:
            self._iota = torch.arange(max(B, 256), dtype=torch.int64, device=self.device)
        items = (L.GatherItem * len(pairs))()
        for it, (src, dst) in zip(items, pairs):
            it.src, it.dst, it.row_bytes, it.src_rows = src.data_ptr(), dst.data_ptr(), src[0].numel() * src.element_size(), src.shape[0]
        L.check(L.lib().ufnd_gather_rows(self._iota.data_ptr(), B, items, len(pairs), L.stream_ptr(self.device)), "ufnd_gather_rows")

    _iota: Optional[torch.Tensor] = None
    _timeline: Optional[list] = None     # tools/step_timeline.py sets a list: (tag, timing event) pairs are appended

    def _mark(self, tag: str, stream) -> None:
        if self._timeline is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(stream)
            self._timeline.append((tag, ev))

    def train_step_pipelined(self, batch: Dict[str, torch.Tensor], next_batch: Optional[Dict[str, torch.Tensor]]) -> dict:
        """train_step for raw batches whose features were started by prefetch_features():
          compute stream : wait features(i) -> head fwd/bwd(i) -> [all-reduce(i)] -> clip + AdamW(i)
          encoder streams: text(i+1) || visual(i+1), launched right after the head of i is enqueued
        so the head, the exchange and the optimizer of step i all hide behind the (frozen) encoders of
        step i+1.  Same arithmetic and order of parameter updates as train_step (bit-identical)."""
        # The head / exchange / optimizer chain is short but serial (45 small kernels) and shares the GPU with two
        # encoder graphs full of whole-CU GEMM blocks: on a normal-priority queue every one of its kernels waits for
        # CUs (0.4 ms alone -> 0.8 ms beside one encoder, 2.3 ms beside both).  It runs on a high-priority stream.
        if self._hp_stream is None:
            self._hp_stream = torch.cuda.Stream(device=self.device, priority=-1)
        caller = torch.cuda.current_stream(self.device)
        self._hp_stream.wait_stream(caller)
        with torch.cuda.stream(self._hp_stream):
            out = self._train_step_pipelined(batch, next_batch)
        caller.wait_stream(self._hp_stream)
        return out

    def _train_step_pipelined(self, batch, next_batch) -> dict:
        B = _batch_size(batch)
        slot = self._slot
        b = self._bufs(B, True, slot)
        main = torch.cuda.current_stream(self.device)
        if self._feat_ready[slot] is None:
            raise RuntimeError("train_step_pipelined: call prefetch_features(batch) for the first batch")
        inputs_ready = torch.cuda.Event()
        inputs_ready.record(main)                  # next_batch (if any) exists on the device by now
        self._mark("step0", main)
        for ev in self._feat_ready[slot]:
            main.wait_event(ev)
        self._feat_ready[slot] = None
        self._mark("head0", main)
        self._stage_small_inputs(b, batch, B)
        if self.temporal_net is not None:      # fakesv_dataset.py:176: U = tsync.align(T, V), written straight into the step's buffer
            self.temporal_net.align_batch(b["text"], b["visual"], out=b["temporal"])
        else:
            b["temporal"].copy_(batch["temporal_features"])
        # With a gradient exchange, the next batch's encoders are enqueued BEFORE the head: the RCCL launches inside
        # _fwd_bwd hold the host until the work they depend on has run (measured: encoders enqueued after a collective
        # reached the GPU 40 us after the head's end and the step degenerated into head -> encoders -> optimizer in
        # series).  They read the other input slot, so the order of enqueueing changes no value.
        early = next_batch is not None and self.reducer.active
        if early:
            self.prefetch_features(next_batch, slot ^ 1, inputs_ready)
        self._fwd_bwd(b, B)
        done = torch.cuda.Event()
        done.record(main)
        self._slot_free[slot] = done
        self._mark("head1", main)
        if next_batch is not None and not early:
            self.prefetch_features(next_batch, slot ^ 1, inputs_ready)
        self.reducer.finish()
        self._mark("reduce1", main)
        self.optim.clip_and_step()
        self._mark("opt1", main)
        self._slot ^= 1
        return {"loss": self.optim.state.float_view("loss"), "probs": b["probs"], "y": b["label"],
                "forensic": b["forensic"], "logits": b["logits"]}

    def measure_gemm_time(self, batch: Dict[str, torch.Tensor], steps: int = 3) -> Tuple[float, int]:
        """(ms of ufnd_gemm_bf16 per step, launches per step): HIP events recorded on the launch
        stream around every GEMM launch of both encoders (an instrumented pass, not the timed one)."""
        events: List[Tuple[torch.cuda.Event, torch.cuda.Event]] = []
        shapes: List[Tuple[int, int, int]] = []
        originals = []
        passes: List[Tuple[int, int]] = []
        for enc in (self.text_encoder, self.visual_encoder):
            for name in ("_gemm", "_gemm_ln", "_qkv_attn"):       # plain, LayerNorm-aware and fused-attention entries: one kernel family
                orig = getattr(enc, name)

                def timed(A, W, *a, _orig=orig, **kw):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    _orig(A, W, *a, **kw)
                    e1.record()
                    events.append((e0, e1))
                    shapes.append((int(A.shape[0]), int(W.shape[0]), int(W.shape[1])))
                originals.append((enc, name, orig))
                setattr(enc, name, timed)
        try:
            for _ in range(steps):
                # park the GPU behind a ~12 ms spin so the host has every launch and event of the pass queued
                # before the GPU reaches them: event deltas then measure GPU time, not host enqueue latency
                for enc, args in ((self.text_encoder, (batch["input_ids"], batch["attention_mask"])), (self.visual_encoder, (batch["frames"],))):
                    torch.cuda._sleep(24_000_000)
                    mark = len(events)
                    enc(*args)
                    passes.append((mark, len(events)))
            torch.cuda.synchronize(self.device)
        finally:
            for enc, name, orig in originals:
                setattr(enc, name, orig)
        # An event costs the queue a marker packet.  Its price is taken IN SITU: where two GEMMs follow each other
        # with nothing in between, (end event of the first -> start event of the second) is exactly one
        # marker-to-marker interval of the busy queue; the lower quartile of all such gaps of a pass is that price
        # (the other gaps contain an attention / LayerNorm kernel).  launch duration = (end - start) - price.
        gaps = []
        for lo, hi in passes:
            gaps += [events[k][1].elapsed_time(events[k + 1][0]) for k in range(lo, hi - 1)]
        gaps.sort()
        marker = gaps[len(gaps) // 4] if gaps else 0.0
        self.last_marker_us = marker * 1e3
        total = sum(max(0.0, e0.elapsed_time(e1) - marker) for e0, e1 in events)
        self.last_raw_interval_us = sum(e0.elapsed_time(e1) for e0, e1 in events) / max(1, len(events)) * 1e3
        self.last_gemm_by_shape = {}
        for (e0, e1), shp in zip(events, shapes):
            d = self.last_gemm_by_shape.setdefault("x".join(map(str, shp)), [0, 0.0])
            d[0] += 1
            d[1] += max(0.0, e0.elapsed_time(e1) - marker)
        return total / steps, len(events) // steps

    def _forward_batch(self, batch, split: str) -> Dict[str, torch.Tensor]:
        """Forward only (forensic_trainer.py:238-271); dropout follows the split like .train(is_train)."""
        B = _batch_size(batch)
        train = split == "train"
        b = self._bufs(B, False)
        self._load_batch(b, batch, split)
        if self.gnn_model is not None:
            self._gnn_forward(b, batch, B, split, train)
        self._enqueue_forward(b, B, train, False)
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
        lookahead = (is_train and self.cfg.encode_inline and int(self.cfg.encoder_lookahead) > 1 and self.cfg.use_graph and
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
                f = self._bufs(_batch_size(batch), False)["forensic"]
                loss = self.optim.state.float_view("loss")
            # device-side clones; ONE host sync per epoch instead of five per step
            losses.append(loss.clone())
            ys.append(out["y"].clone())
            p1s.append(out["probs"][:, 1].clone())
            fors.append(f.clone())
        if is_train and self.cfg.encode_inline:
            # fold guard of the in-step encoders (encoders.py): one eager, guarded pass per epoch over the inputs of the
            # last batch; a row outside the folded LayerNorm's accuracy range switches that encoder to materialised
            # LayerNorms, and its captured graph is rebuilt on the next step
            for e in list(self._enc_bufs.values()):
                for enc, key, tag, args in ((self.text_encoder, "g_text", "text", e.get("last_text")), (self.visual_encoder, "g_vis", "vis", e.get("last_vis"))):
                    if enc is not None and args is not None and enc.guarded_pass(*args):
                        for e2 in self._enc_bufs.values():
                            e2[key] = None
                            for k in [k for k in e2.get("pinned", {}) if k[0] == tag]:
                                del e2["pinned"][k]
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
                                                                      torch.stack(losses).sum(), n_loss, self.group)
        loss_mean = float(loss_mean_local.cpu())
        f_np = f_cat.cpu().numpy()
        forensic = {"emotion_intensity": f_np[0], "semantic_conflict": f_np[1], "temporal_delay": f_np[2]}
        metrics = aggregate_epoch_metrics(y_true=y_cat.cpu().numpy(), y_score=p1_cat.cpu().numpy().astype(float),
                                          forensic=forensic, threshold=0.5, include_cm=False)
        return loss_mean, metrics

    def _group_inputs(self, ds: CachedTensorDataset, rows: torch.Tensor, slot: int) -> dict:
        """Raw inputs and small per-sample inputs of a lookahead group, gathered into PERSISTENT buffers (one set per group
        size and slot): the encoder graphs are captured per input address, so a group's inputs must not move."""
        n = int(rows.numel())
        key = (n, slot)
        src = {"input_ids": ds.ids_tok, "attention_mask": ds.mask_tok, "frames": ds.frames, "audio_features": ds.A, "aux": ds.AUX,
               "label": ds.y, "gnn_feat": ds.G, "temporal_features": ds.U}
        buf = self._grp_in.get(key)
        if buf is None:
            buf = self._grp_in[key] = {k: torch.empty((n,) + tuple(t.shape[1:]), dtype=t.dtype, device=self.device) for k, t in src.items()}
        for k, t in src.items():
            torch.index_select(t, 0, rows, out=buf[k])
        out = dict(buf)
        out["index"] = rows
        return out

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
                cur = self._group_inputs(ds, idx[lo:hi], self._gslot)
                self.prefetch_features(cur, group=True)
            nxt = self._group_inputs(ds, idx[cuts[i + 1][0]:cuts[i + 1][1]], self._gslot ^ 1) if i + 1 < len(cuts) else None
            self.train_group_pipelined(cur, nxt, on_step=sink)
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
                save_checkpoint({"fusion": {k: v.cpu() for k, v in self.fusion.state_dict().items()},
                                 "clf": {k: v.cpu() for k, v in self.clf.state_dict().items()},
                                 "gnn": ({k: v.cpu() for k, v in self.gnn_model.state_dict().items()} if self.gnn_model is not None
                                         else self.gnn.state_dict() if self.gnn is not None else None), "cfg": dict(self.cfg.__dict__)},
                                self.ckpt_path, self.group)
                if self.rank == 0:
                    print(f"  ↳ saved best checkpoint to {self.ckpt_path} (val_auc={self.best_val_auc:.3f})")
            else:
                self.no_improve += 1
                if self.no_improve >= self.cfg.early_stop_patience:
                    if self.rank == 0:
                        print(f"↳ Early stopping (no val AUC improvement for {self.cfg.early_stop_patience} epochs)")
                    break
        return self.best_val_auc

    def test(self) -> Dict[str, float]:
        # rank 0 reads the checkpoint; every rank then continues with rank 0's parameters (one broadcast of the arena),
        # so no rank ever evaluates its shard with a stale or half-written file
        if self.rank == 0 and os.path.exists(self.ckpt_path):
            ck = torch.load(self.ckpt_path, map_location="cpu", weights_only=True)
            self.fusion.load_state_dict(ck["fusion"])
            self.clf.load_state_dict(ck["clf"])
            if self.gnn_model is not None and ck.get("gnn") is not None:
                self.gnn_model.load_state_dict(ck["gnn"])
        broadcast_from_rank0(self.arena.data, self.group)
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


# ---------------------------------------------------------------------------------------------
def synthetic_cache(n: int, seed: int = 0, gnn_dim: int = 128, with_raw: bool = False, seq_len: int = 128,
                    frames: int = 1, vocab: int = 30522) -> Dict:
    """FakeSV-shaped synthetic cache: the reference's own smoke test feeds randn
    features (scripts/smoke_test_v2.py:43-45).  70/15/15 split."""
    g = torch.Generator().manual_seed(seed)

    def l2(x):
        return x / x.norm(dim=1, keepdim=True)
    cache = {"ids": np.array([f"syn{i}" for i in range(n)]), "labels": torch.randint(0, 2, (n,), generator=g).numpy(),
             "text": l2(torch.randn(n, 768, generator=g)).numpy(), "audio": l2(torch.randn(n, 128, generator=g)).numpy(),
             "visual": l2(torch.randn(n, 512, generator=g)).numpy(), "temporal": torch.randn(n, 256, generator=g).numpy(),
             "aux": torch.rand(n, 2, generator=g).numpy(), "gnn_Z": torch.randn(n, gnn_dim, generator=g).numpy()}
    perm = torch.randperm(n, generator=g).numpy()
    a, b = int(0.7 * n), int(0.85 * n)
    cache["split"] = (np.sort(perm[:a]), np.sort(perm[a:b]), np.sort(perm[b:]))
    if with_raw:
        ids = torch.randint(0, vocab, (n, seq_len), generator=g)
        ids[:, 0] = min(101, vocab - 1)
        lens = torch.randint(min(16, seq_len), seq_len + 1, (n,), generator=g)
        cache["input_ids"] = ids.numpy()
        cache["attention_mask"] = (torch.arange(seq_len)[None] < lens[:, None]).to(torch.int32).numpy()
        cache["frames"] = torch.randn(n, frames, 3, 224, 224, generator=g).numpy()
    return cache
