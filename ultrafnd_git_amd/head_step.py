"""The fusion + classifier part of one train step on static device buffers -- the body of the reference's loop,
src/training/forensic_trainer.py:285-291 (forward, F.cross_entropy, backward), as the C-ABI sequence

    ufnd_fusion_forward -> ufnd_classifier_forward -> ufnd_softmax_ce -> ufnd_classifier_backward -> ufnd_fusion_backward_phase

eager or replayed from captured hipGraphs.  With a gradient exchange the backward is cut after the fuse_mlp phase, so that
bucket 0 of the exchange (dp.py) starts inside backward.  `HeadStep` owns the per-batch-size static buffers (graph replay
needs fixed addresses; two input slots, so that the encoders can fill step i+1's inputs while step i's backward still
reads its own) and the one-launch gathers that fill them."""
from __future__ import annotations

import ctypes as C
from typing import Callable, Dict, Optional, Tuple

import torch

from . import _lib as L
from .data import CachedTensorDataset


class HeadStep:
    def __init__(self, cfg, device: torch.device, fusion, clf, optim, reducer, ce_weights=(1.0, 1.0), gnn_dims: Optional[int] = None):
        self.cfg, self.device, self.fusion, self.clf, self.optim, self.reducer = cfg, device, fusion, clf, optim, reducer
        self.ce_w = ce_weights
        self.gnn_node_dim = gnn_dims            # width of the in-graph GNN's node features (integrated variant), else None
        self.step_bufs: Dict[Tuple[int, bool, int], dict] = {}
        # the head's fwd/bwd as a graph (default with use_graph) or eager with the dW side stream
        self.head_graph = bool(cfg.use_graph and cfg.head_graph)
        # the two-call fused form of the step (ufnd_head_forward_loss / ufnd_head_backward: 22 launches instead of 26, same bits) whenever
        # the criterion is the plain mean CE; the weighted / label-smoothed criterion keeps the five module-level calls
        self.fused_entries = not (cfg.label_smoothing > 0.0 or cfg.class_weighting) and bool(getattr(cfg, "fused_head", True))
        self._dw_stream: Optional[torch.cuda.Stream] = None
        self._iota: Optional[torch.Tensor] = None

    # ------------------------------------------------------------------ buffers
    def bufs(self, B: int, train: bool, slot: int = 0) -> dict:
        key = (B, train, slot)
        if key not in self.step_bufs:
            dev, f32 = self.device, torch.float32
            dims = self.clf.dims()
            dims.fusion_dropout = self.fusion.dropout
            n_f = L.lib().ufnd_fusion_workspace_floats(C.byref(dims), B)
            n_c = L.lib().ufnd_clf_workspace_floats(C.byref(dims), B)
            fws = torch.empty(n_f, dtype=f32, device=dev)
            cws = torch.empty(n_c, dtype=f32, device=dev)
            ld = C.c_int(0)
            xin = L.lib().ufnd_clf_input_panel(C.byref(dims), cws.data_ptr(), B, C.byref(ld))
            self.step_bufs[key] = {
                "dims": dims, "fws": fws, "cws": cws, "xin": xin, "ldx": ld.value,
                "text": torch.empty(B, 768, dtype=f32, device=dev), "audio": torch.empty(B, 128, dtype=f32, device=dev),
                "visual": torch.empty(B, 512, dtype=f32, device=dev), "temporal": torch.empty(B, 256, dtype=f32, device=dev),
                "gnn": torch.empty(B, self.fusion.gnn_dim, dtype=f32, device=dev),
                "aux": torch.empty(B, 2, dtype=f32, device=dev), "label": torch.empty(B, dtype=torch.int64, device=dev),
                "logits": torch.empty(B, 2, dtype=f32, device=dev), "probs": torch.empty(B, 2, dtype=f32, device=dev),
                "forensic": torch.empty(3, B, dtype=f32, device=dev), "dlogits": torch.empty(B, 2, dtype=f32, device=dev),
                "dfused": torch.empty(B, self.fusion.hidden, dtype=f32, device=dev), "graph": None}
            if self.gnn_node_dim is not None:
                self.step_bufs[key].update({"gnn_x": torch.empty(B, self.gnn_node_dim, dtype=f32, device=dev),
                                            "gnn_adj": torch.zeros(B, B, dtype=f32, device=dev),
                                            "dgnn": torch.empty(B, self.fusion.gnn_dim, dtype=f32, device=dev)})
        return self.step_bufs[key]

    # ------------------------------------------------------------------ one-launch gathers into the static buffers
    def _gather(self, idx_ptr: int, n: int, pairs, what: str) -> None:
        items = (L.GatherItem * len(pairs))()
        for it, (src_ptr, dst, rb, rows) in zip(items, pairs):
            it.src, it.dst, it.row_bytes, it.src_rows = src_ptr, dst.data_ptr(), rb, rows
        L.check(L.lib().ufnd_gather_rows(idx_ptr, n, items, len(pairs), L.stream_ptr(self.device)), what)

    def _identity(self, B: int) -> torch.Tensor:
        if self._iota is None or self._iota.numel() < B:
            self._iota = torch.arange(max(B, 256), dtype=torch.int64, device=self.device)
        return self._iota

    def gather_cached(self, b: dict, ds: CachedTensorDataset, idx: torch.Tensor) -> None:
        """Cached features named by row index: one launch gathers every tensor (and gnn_Z, forensic_trainer.py:240-252)."""
        idx = idx.to(self.device, torch.int64).contiguous()
        pairs = []
        for src, dst in ((ds.T, b["text"]), (ds.A, b["audio"]), (ds.V, b["visual"]), (ds.U, b["temporal"]), (ds.AUX, b["aux"]),
                         (ds.y, b["label"]), (ds.G, b["gnn"])):
            rb = src[0].numel() * src.element_size()
            if not src.is_contiguous() or src.dtype != dst.dtype or rb != dst[0].numel() * dst.element_size() or src.device != dst.device:
                raise RuntimeError(f"cached tensor {tuple(src.shape)} {src.dtype} does not match its batch buffer {tuple(dst.shape)} {dst.dtype}")
            pairs.append((src.data_ptr(), dst, rb, src.shape[0]))
        self._gather(idx.data_ptr(), idx.numel(), pairs, "ufnd_gather_rows")

    def stage_group_rows(self, b: dict, group, grp: dict, k: int, B: int) -> None:
        """Rows [kB, (k+1)B) of a lookahead group's features and small inputs into the step's static buffers: ONE launch."""
        pairs = []
        for src, dst in ((grp["text"], b["text"]), (grp["visual"], b["visual"]), (group["audio_features"], b["audio"]), (group["aux"], b["aux"]),
                         (group["label"], b["label"]), (group["gnn_feat"], b["gnn"])):
            if not (src.device == dst.device and src.dtype == dst.dtype and src.is_contiguous() and tuple(src.shape[1:]) == tuple(dst.shape[1:]) and
                    (src[0].numel() * src.element_size()) % 8 == 0):
                raise RuntimeError(f"lookahead group tensor {tuple(src.shape)} {src.dtype} does not match its step buffer {tuple(dst.shape)} {dst.dtype}")
            rb = src[0].numel() * src.element_size()
            pairs.append((src.data_ptr() + k * B * rb, dst, rb, B))
        self._gather(self._identity(B).data_ptr(), B, pairs, "ufnd_gather_rows")

    def stage_small_inputs(self, b: dict, batch, B: int) -> None:
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
        self._gather(self._identity(B).data_ptr(), B,
                     [(src.data_ptr(), dst, src[0].numel() * src.element_size(), src.shape[0]) for src, dst in pairs], "ufnd_gather_rows")

    # ------------------------------------------------------------------ launches
    def _io(self, b: dict) -> "L.HeadIO":
        if "io" not in b:
            io = L.HeadIO()
            io.text, io.audio, io.visual, io.temporal = (b[k].data_ptr() for k in ("text", "audio", "visual", "temporal"))
            io.gnn, io.aux, io.labels = b["gnn"].data_ptr(), b["aux"].data_ptr(), b["label"].data_ptr()
            io.fusion_workspace, io.clf_workspace = b["fws"].data_ptr(), b["cws"].data_ptr()
            io.logits, io.probs, io.forensic, io.d_logits = (b[k].data_ptr() for k in ("logits", "probs", "forensic", "dlogits"))
            b["io"] = io
        return b["io"]

    def enqueue_forward(self, b: dict, B: int, train: bool, with_loss_grad: bool) -> None:
        lib, s, st = L.lib(), L.stream_ptr(self.device), self.optim.state.ptr
        d = b["dims"]
        if self.fused_entries and train and with_loss_grad:
            L.check(lib.ufnd_head_forward_loss(C.byref(d), C.byref(self.fusion.param_table()), C.byref(self.clf.param_table()), C.byref(self._io(b)),
                                               B, 1, st, s), "ufnd_head_forward_loss")
            return
        L.check(lib.ufnd_fusion_forward(C.byref(d), C.byref(self.fusion.param_table()), b["text"].data_ptr(),
                                        b["audio"].data_ptr(), b["visual"].data_ptr(), b["temporal"].data_ptr(),
                                        b["gnn"].data_ptr(), B, int(train), b["fws"].data_ptr(), b["xin"], b["ldx"], None,
                                        b["forensic"].data_ptr(), st, s), "ufnd_fusion_forward")
        L.check(lib.ufnd_classifier_forward(C.byref(d), C.byref(self.clf.param_table()), b["xin"], b["ldx"],
                                            b["aux"].data_ptr(), B, int(train), b["cws"].data_ptr(), b["logits"].data_ptr(),
                                            b["probs"].data_ptr(), st, s), "ufnd_classifier_forward")
        if self.cfg.label_smoothing > 0.0 or self.cfg.class_weighting:
            L.check(lib.ufnd_softmax_ce_weighted(b["logits"].data_ptr(), b["label"].data_ptr(), B, self.ce_w[0], self.ce_w[1],
                                                 float(self.cfg.label_smoothing), None,
                                                 b["dlogits"].data_ptr() if with_loss_grad else None, st, s), "ufnd_softmax_ce_weighted")
        else:
            L.check(lib.ufnd_softmax_ce(b["logits"].data_ptr(), b["label"].data_ptr(), B, None,
                                        b["dlogits"].data_ptr() if with_loss_grad else None, st, s), "ufnd_softmax_ce")

    def enqueue_backward(self, b: dict, B: int, part: int = 0, linear_grads: bool = True) -> None:
        """part 0: the whole backward; 1: classifier backward + the fuse_mlp phase of the fusion backward (bucket 0 of
        the gradient exchange is complete afterwards); 2: the rest of the fusion backward.  linear_grads=False leaves the
        Linear layers' dW / db products out (factor exchange: linear_grads_from_factors forms them over all ranks' rows)."""
        lib, s, st = L.lib(), L.stream_ptr(self.device), self.optim.state.ptr
        d = b["dims"]
        skip = 0 if linear_grads else L.BWD_NO_LINEAR_GRADS
        if self.fused_entries:
            side_f = None
            if not self.head_graph:
                if self._dw_stream is None:
                    self._dw_stream = torch.cuda.Stream(device=self.device)
                side_f = self._dw_stream.cuda_stream
            L.check(lib.ufnd_head_backward(C.byref(d), C.byref(self.fusion.param_table()), C.byref(self.fusion.grad_table()), C.byref(self.clf.param_table()),
                                           C.byref(self.clf.grad_table()), C.byref(self._io(b)), B, 1, st, s, side_f, 1,
                                           (L.BWD_ALL, L.BWD_FUSE_MLP, L.BWD_REST)[part] | skip), "ufnd_head_backward")
            return
        # Eager launches: dW / parameter-gradient kernels run beside the dX chain on a second stream (joined
        # at the end).  Inside a captured hipGraph the fork/join costs more than it hides (ROCm 7 replays
        # multi-branch graphs almost serially -- measured), so the graph keeps one stream.
        side = None
        if not self.head_graph:
            if self._dw_stream is None:
                self._dw_stream = torch.cuda.Stream(device=self.device)
            side = self._dw_stream.cuda_stream
        if part != 2:
            L.check(lib.ufnd_classifier_backward_ex(C.byref(d), C.byref(self.clf.param_table()), C.byref(self.clf.grad_table()), B, 1,
                                                    b["cws"].data_ptr(), b["dlogits"].data_ptr(), b["dfused"].data_ptr(),
                                                    self.fusion.hidden, st, s, side, 0, skip), "ufnd_classifier_backward")
        L.check(lib.ufnd_fusion_backward_phase(C.byref(d), C.byref(self.fusion.param_table()), C.byref(self.fusion.grad_table()),
                                               b["text"].data_ptr(), b["audio"].data_ptr(), b["visual"].data_ptr(),
                                               b["temporal"].data_ptr(), b["gnn"].data_ptr(), B, 1, b["fws"].data_ptr(),
                                               b["dfused"].data_ptr(), self.fusion.hidden, None, st, s, side, 1,
                                               (L.BWD_ALL, L.BWD_FUSE_MLP, L.BWD_REST)[part] | skip), "ufnd_fusion_backward_phase")

    # ------------------------------------------------------------------ factor form of the gradient exchange (dp.FactorExchange)
    def factor_pack(self, b: dict, B: int) -> torch.Tensor:
        """The rank's factor panels of this step's backward (run with linear_grads=False) in one contiguous buffer: ONE launch."""
        if "pack" not in b:
            b["pack"] = torch.empty(L.lib().ufnd_head_factor_floats(C.byref(b["dims"]), B), dtype=torch.float32, device=self.device)
        L.check(L.lib().ufnd_head_pack_factors(C.byref(b["dims"]), b["text"].data_ptr(), b["audio"].data_ptr(), b["visual"].data_ptr(),
                                               b["temporal"].data_ptr(), b["gnn"].data_ptr(), B, b["fws"].data_ptr(), b["cws"].data_ptr(),
                                               b["pack"].data_ptr(), L.stream_ptr(self.device)), "ufnd_head_pack_factors")
        return b["pack"]

    def linear_grads_from_factors(self, b: dict, B: int, packs: torch.Tensor, stride: int, ranks: int) -> None:
        """The summed dW / db of the head's 13 Linear layers over ranks x B rows, from the gathered packs: ONE grouped launch."""
        if packs.device != self.device or packs.dtype != torch.float32 or packs.numel() < stride * ranks:
            raise RuntimeError(f"gathered factor packs: {packs.numel()} {packs.dtype} values on {packs.device} for {ranks} ranks of {stride}")
        L.check(L.lib().ufnd_head_linear_grads_from_factors(C.byref(b["dims"]), C.byref(self.fusion.grad_table()), C.byref(self.clf.grad_table()),
                                                            packs.data_ptr(), stride, ranks, B, L.stream_ptr(self.device)),
                "ufnd_head_linear_grads_from_factors")

    def _fwd_bwd_factors(self, b: dict, B: int, post: Callable[[], None], tail: Callable[[], None]) -> None:
        """fwd_bwd with the factor exchange: the backward leaves the Linear products out and packs its factor panels; the packs are
        all-gathered while `tail()` runs; reducer.finish() forms the summed Linear gradients (one launch over world x B rows)."""
        def body():
            self.enqueue_forward(b, B, True, True)
            self.enqueue_backward(b, B, 0, linear_grads=False)
            self.factor_pack(b, B)
        if not self.head_graph:
            body()
        else:
            if b.get("graph_factors") is None:
                side = torch.cuda.Stream(device=self.device)
                side.wait_stream(torch.cuda.current_stream(self.device))
                with torch.cuda.stream(side):          # warm-up outside capture (lazy module loads etc.)
                    body()
                torch.cuda.current_stream(self.device).wait_stream(side)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    body()
                b["graph_factors"] = g
            b["graph_factors"].replay()
        post()                          # (the integrated variant's GNN backward: its gradients are in the all-reduced ranges)
        self.reducer.start_factors(b["pack"], lambda packs, stride, ranks: self.linear_grads_from_factors(b, B, packs, stride, ranks))
        tail()

    def feature_grads(self, b: dict, B: int, d_text: Optional[torch.Tensor], d_visual: Optional[torch.Tensor]) -> None:
        """d loss / d text_features, d loss / d visual_features out of the workspace the fusion backward has filled (where the
        trainable encoders' backward starts; the reference's trainer treats both as cached data, forensic_trainer.py:60-83)."""
        L.check(L.lib().ufnd_fusion_feature_grads(C.byref(b["dims"]), C.byref(self.fusion.param_table()), b["fws"].data_ptr(), B, L.ptr(d_text),
                                                  L.ptr(d_visual), self.optim.state.ptr, L.stream_ptr(self.device)), "ufnd_fusion_feature_grads")

    def fwd_bwd(self, b: dict, B: int, post: Optional[Callable[[], None]] = None, tail: Optional[Callable[[], None]] = None) -> None:
        """fusion fwd -> clf fwd -> CE -> clf bwd -> fusion bwd, eager or replayed from a hipGraph.  With a gradient
        exchange (data parallel) the backward is cut after the fuse_mlp phase: bucket 0 of the exchange starts there and
        runs beside the rest of backward, bucket 1 follows it (dp.py); the caller's reducer.finish() joins both.
        `post()` (optional) is enqueued eagerly behind the backward and in front of the head's last bucket (the integrated
        variant's GNN backward: its gradients close that bucket); `tail()` behind that bucket's start (the trainable encoders'
        backward, which starts its own buckets)."""
        dp = self.reducer.active
        post = post or (lambda: None)
        tail = tail or (lambda: None)
        if dp and getattr(self.reducer, "factors", False):
            return self._fwd_bwd_factors(b, B, post, tail)

        def first():
            self.enqueue_forward(b, B, True, True)
            self.enqueue_backward(b, B, 1 if dp else 0)

        def second():
            self.enqueue_backward(b, B, 2)
        if not self.head_graph:
            first()
            if dp:
                self.reducer.start(0)
                second()
                post()
                self.reducer.start(1)
            else:
                post()
            tail()
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
            post()                      # (eager, behind the graph)
            self.reducer.start(1)
        else:
            post()
        tail()
