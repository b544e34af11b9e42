"""Trainable encoders: forward with saved activations + hand-written backward for BertTextEncoder / ClipVisualEncoder
(Tier-B backward; `TrainConfig.train_encoders`).

The reference keeps its encoder frozen (src/core_blocks/text_blocks.py:52 `.eval()`, :63 `inference_mode`) and trains on cached
features, so nothing here replaces a reference code path: it is the fine-tuning capability north_star's "forward/backward hot
path" of the encoders asks for.  Parity is against torch autograd over oracle/encoders_ref.py ("parity unpinned by the
reference": DESIGN.md section 2).

How it is built.
  * Parameters.  `groups()` lists the encoder's tensors in gradient-ready order (last layer first, embeddings last); the
    trainer lays them out in its ONE flat fp32 arena behind the head's, so the global-norm clip, AdamW and the gradient
    exchange stay single contiguous ranges.  q/k/v weights (and biases) of a layer are adjacent: the stacked (3H, H) operand is
    a view.  `bind()` re-points the encoder's master tensors at the arena.
  * Operands.  Every Linear has two bf16 copies of its fp32 master, W (forward, wgrad shape) and W^T (data gradient), re-cast
    after every optimizer step (`refresh_operands`: ufnd_cast_bf16 / ufnd_transpose_bf16 -- ~6 B per parameter per step).
  * Forward (`forward_train`).  The un-folded layer sequence (one LayerNorm kernel per LayerNorm), with what the backward needs
    kept per layer: the bf16 GEMM inputs, fused q|k|v rows, attention output and per-query log-sum-exp, the pre-LayerNorm sums
    (fp32), FFN1's pre-activations.
  * Backward (`backward`).  Per layer: LayerNorm backward (row kernel, two-stage parameter sums) -> for each Linear: transpose
    dy and x (dy's column sums = the bias gradient fall out of the same pass), weight gradient as the forward's NT kernel over
    the token dimension with split-K slabs, data gradient as the NT kernel on W^T with the residual-branch gradient or the
    activation derivative fused into its epilogue -> flash-style attention backward.  No atomics anywhere: gradients are
    run-to-run identical."""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch

from . import _lib as L

ACT_NONE, ACT_GELU, ACT_QUICK_GELU, ACT_GELU_BWD, ACT_QUICK_GELU_BWD = 0, 1, 2, 3, 4


def _pad64(n: int) -> int:
    return (n + 63) // 64 * 64


class _Backprop:
    """Shared plumbing: parameter binding, operand copies, and the backward of one Linear / LayerNorm."""

    def __init__(self, enc):
        self.enc = enc
        self.arena = None
        self.prefix = ""
        self._ops: Dict[str, Tuple[torch.Tensor, torch.Tensor]] = {}      # linear name -> (W bf16 (N, K), W^T bf16 (K, N))
        self._scratch: Dict[Tuple, dict] = {}
        self.saved: Optional[dict] = None
        self._refresh = None               # (device table, items, tiles, names left to the per-Linear path) of the grouped operand refresh
        self._views: Dict[Tuple, torch.Tensor] = {}
        self._lk_cache: Dict[int, dict] = {}
        self._pending_ln = None            # (job, scratch, buffer index) of a LayerNorm backward's deferred dgamma / dbeta finish (rides in the next Linear's finish launch)
        self._wg_side = None               # the stream of the weight-gradient products
        self._wg_open = False
        # weight-gradient products on a second stream beside the data-gradient chain (they are not on backward's critical path).  Measured
        # A/B/A/B on one box: 9.17 / 9.24 ms per step with it, 9.48 / 9.42 without -- once the host enqueue had come down to 4.6-5.2 ms per step
        # (memoised arena views); at 6.4 ms the events' extra 2-3 ms of host time made the step host-bound and the overlap invisible
        self.overlap_wgrad = True

    # ------------------------------------------------------------------ parameters
    def groups(self) -> List[List[Tuple[str, Tuple[int, ...]]]]:
        raise NotImplementedError

    def linears(self) -> Dict[str, Tuple[List[str], Optional[List[str]]]]:
        """linear name -> (weight keys stacked row-wise, bias keys or None)."""
        raise NotImplementedError

    def bind(self, arena, prefix: str) -> None:
        """Re-point the encoder's fp32 masters at their arena views (values preserved)."""
        self.arena, self.prefix = arena, prefix
        w = self.enc._w
        with torch.no_grad():
            for k in list(w):
                v = arena.view(prefix + k)
                v.copy_(w[k].to(v.device))
                w[k] = v
        self.enc._packed = None
        self.enc.weights_version += 1
        self._ops.clear()
        self._refresh = None
        self._views.clear()

    def _stacked(self, buf: torch.Tensor, keys: List[str]) -> torch.Tensor:
        """View of adjacent arena tensors as one matrix (q/k/v -> (3H, H)) or vector.  Memoised per buffer: a backward asks for ~400 of
        them, and building a view costs more host time than enqueueing the kernel that uses it."""
        ck = (buf.data_ptr(), keys[0], len(keys))
        v = self._views.get(ck)
        if v is None:
            v = self._views[ck] = self._stacked_view(buf, keys)
        return v

    def _stacked_view(self, buf: torch.Tensor, keys: List[str]) -> torch.Tensor:
        o0, s0 = self.arena.offsets[self.prefix + keys[0]]
        n = 0
        for k in keys:
            o, s = self.arena.offsets[self.prefix + k]
            if o != o0 + n:
                raise RuntimeError(f"{keys} are not adjacent in the arena")
            cnt = 1
            for dim in s:
                cnt *= int(dim)
            n += cnt
        rows = sum(self.arena.offsets[self.prefix + k][1][0] for k in keys)
        rest = tuple(s0[1:])
        return buf[o0:o0 + n].view((rows,) + rest)

    def _lk(self, i: int) -> dict:
        """The parameter names of layer i (built once: a dozen formatted strings per layer per pass otherwise)."""
        k = self._lk_cache.get(i)
        if k is None:
            k = self._lk_cache[i] = self._lk_build(i)
        return k

    def master(self, keys: List[str]) -> torch.Tensor:
        return self._stacked(self.arena.data, keys)

    def grad(self, keys: List[str]) -> torch.Tensor:
        return self._stacked(self.arena.ensure_grad(), keys)

    def refresh_operands(self) -> None:
        """bf16 W and W^T of every Linear from the fp32 masters (after an optimizer step; captured-graph safe: fixed buffers): ONE
        grouped launch over all Linears whose shapes are multiples of 64 (every one of BERT-base / ViT-B), the two-launch form for
        the rest."""
        s = L.stream_ptr(self.enc.device)
        if self._refresh is None:
            import ctypes as C
            items, rest, tile0 = [], [], 0
            for name, (wk, _) in self.linears().items():
                m = self.master(wk)
                m2 = m.reshape(m.shape[0], -1)
                if name not in self._ops:
                    self._ops[name] = (torch.empty(m2.shape, dtype=torch.bfloat16, device=m.device),
                                       torch.empty((m2.shape[1], m2.shape[0]), dtype=torch.bfloat16, device=m.device))
                wb, wt = self._ops[name]
                R, Cc = m2.shape
                ok = (R % 64 == 0 and Cc % 64 == 0 and m2.stride(1) == 1 and m2.stride(0) % 4 == 0 and m2.data_ptr() % 16 == 0 and
                      wb.data_ptr() % 16 == 0 and wt.data_ptr() % 16 == 0)
                if not ok:
                    rest.append(name)
                    continue
                it = L.RefreshItem()
                it.master, it.w, it.wt = m2.data_ptr(), wb.data_ptr(), wt.data_ptr()
                it.rows, it.cols, it.ld_master, it.ld_w, it.ld_wt, it.tile0 = R, Cc, m2.stride(0), wb.stride(0), wt.stride(0), tile0
                tile0 += (R // 64) * (Cc // 64)
                items.append(it)
            table = None
            if items:
                arr = (L.RefreshItem * len(items))(*items)
                table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self.enc.device)
            self._refresh = (table, len(items), tile0, rest)
        table, n_items, tiles, rest = self._refresh
        if n_items:
            L.check(L.lib().ufnd_refresh_operands(table.data_ptr(), n_items, tiles, s), "ufnd_refresh_operands")
        for name in rest:
            m = self.master(self.linears()[name][0])
            m2 = m.reshape(m.shape[0], -1)
            wb, wt = self._ops[name]
            L.check(L.lib().ufnd_cast_bf16(m2.data_ptr(), wb.data_ptr(), m2.numel(), s), "ufnd_cast_bf16")
            L.check(L.lib().ufnd_transpose_bf16(m2.data_ptr(), 1, m2.shape[0], m2.shape[1], m2.stride(0), wt.data_ptr(), wt.stride(0), m2.shape[0],
                                                None, None, 0, s), "ufnd_transpose_bf16")

    # ------------------------------------------------------------------ scratch
    def _bwd_scratch(self, M: int, widths: Tuple[int, ...]) -> dict:
        key = (M,) + widths
        if key not in self._scratch:
            dev, Mp, wmax = self.enc.device, _pad64(M), max(widths)
            lib = L.lib()
            ws = max(lib.ufnd_gemm_bf16_wgrad_workspace_floats(a, b, Mp) for a in widths for b in widths if b % 64 == 0)
            two = lambda make: [make(), make()]
            # two buffer sets: the weight-gradient product of one Linear runs on a second stream while the main stream transposes the next
            # Linear's operands into the other set (busy[k] / ln_busy[k]: the event after which set k may be rewritten)
            self._scratch[key] = {
                "t1": two(lambda: torch.zeros(wmax, Mp, dtype=torch.bfloat16, device=dev)), "t2": two(lambda: torch.zeros(wmax, Mp, dtype=torch.bfloat16, device=dev)),
                "wg": two(lambda: torch.empty(max(1, ws), dtype=torch.float32, device=dev)),
                "cs": two(lambda: torch.empty(lib.ufnd_transpose_colsum_workspace_floats(Mp, wmax), dtype=torch.float32, device=dev)),
                "ln": two(lambda: torch.empty(lib.ufnd_layernorm_bwd_workspace_floats(M, self.enc.hidden), dtype=torch.float32, device=dev)),
                "flip": 0, "busy": [None, None], "ln_flip": 0, "ln_busy": [None, None]}
        return self._scratch[key]

    # ------------------------------------------------------------------ building blocks
    def _gemm(self, A, W, bias, out_bf16=None, out_f32=None, residual=None, act=ACT_NONE):
        self.enc._gemm(A, W, bias, out_bf16=out_bf16, out_f32=out_f32, residual=residual, act=act)

    def _dgrad(self, dy, wt, out_bf16=None, out_f32=None, residual=None, aux=None, act=ACT_NONE):
        """out (M, N) = dy (M, K) x wt (N, K)^T [x act'(aux)] [+ residual]."""
        M, K = dy.shape
        N = wt.shape[0]
        L.check(L.lib().ufnd_gemm_bf16_dgrad(dy.data_ptr(), wt.data_ptr(), L.ptr(residual), L.ptr(aux), L.ptr(out_bf16), L.ptr(out_f32), M, N, K,
                                             dy.stride(0), wt.stride(0), residual.stride(0) if residual is not None else 0,
                                             aux.stride(0) if aux is not None else 0, out_bf16.stride(0) if out_bf16 is not None else 0,
                                             out_f32.stride(0) if out_f32 is not None else 0, act, L.stream_ptr(dy.device)), "ufnd_gemm_bf16_dgrad")

    def _wgrad(self, sc: dict, dy, x, dW: torch.Tensor, db: Optional[torch.Tensor]) -> None:
        """dW (N, K) = dy (M, N)^T x (M, K); db (N) = column sums of dy.  (Overwrites: every step writes every gradient.)  Three
        launches (ufnd_linear_wgrad): both transposes on the caller's stream -- dy may be overwritten behind them -- then the sliced NT
        product and one finish pass (which also carries a pending LayerNorm's dgamma / dbeta finish) on a SECOND stream, beside the
        data-gradient chain that continues on the caller's: the weight gradients are not on backward's critical path.  join_wgrad() ends it."""
        import ctypes as C
        M, N = dy.shape
        K = x.shape[1]
        lib, dev = L.lib(), dy.device
        main = torch.cuda.current_stream(dev)
        if not self.overlap_wgrad:         # everything on the caller's stream (one buffer set)
            job, self._pending_ln = self._pending_ln, None
            L.check(lib.ufnd_linear_wgrad(dy.data_ptr(), dy.stride(0), x.data_ptr(), x.stride(0), M, N, K, dW.reshape(N, -1).data_ptr(), L.ptr(db),
                                          sc["t1"][0].data_ptr(), sc["t2"][0].data_ptr(), sc["t1"][0].stride(0), sc["wg"][0].data_ptr(), sc["cs"][0].data_ptr(),
                                          C.byref(job[0]) if job is not None else None, L.WGRAD_ALL, main.cuda_stream), "ufnd_linear_wgrad")
            return
        if self._wg_side is None:
            self._wg_side = torch.cuda.Stream(device=dev)
        side = self._wg_side
        k = sc["flip"]
        sc["flip"] ^= 1
        if sc["busy"][k] is not None:
            main.wait_event(sc["busy"][k])             # the product that read this buffer set last
        t1, t2 = sc["t1"][k], sc["t2"][k]
        dW2 = dW.reshape(N, -1)
        job, self._pending_ln = self._pending_ln, None
        args = (dy.data_ptr(), dy.stride(0), x.data_ptr(), x.stride(0), M, N, K, dW2.data_ptr(), L.ptr(db), t1.data_ptr(), t2.data_ptr(), t1.stride(0),
                sc["wg"][k].data_ptr(), sc["cs"][k].data_ptr(), C.byref(job[0]) if job is not None else None)
        L.check(lib.ufnd_linear_wgrad(*args, L.WGRAD_TRANSPOSE, main.cuda_stream), "ufnd_linear_wgrad")
        ev = torch.cuda.Event()
        ev.record(main)
        side.wait_event(ev)
        L.check(lib.ufnd_linear_wgrad(*args, L.WGRAD_PRODUCT, side.cuda_stream), "ufnd_linear_wgrad")
        done = torch.cuda.Event()
        done.record(side)
        sc["busy"][k] = done
        if job is not None:
            job[1]["ln_busy"][job[2]] = done            # the LayerNorm partials it finished may be overwritten after this
        self._wg_open = True

    def join_wgrad(self) -> None:
        """The caller's stream waits for every weight-gradient product started so far (end of a backward)."""
        if self._wg_open:
            torch.cuda.current_stream(self.enc.device).wait_stream(self._wg_side)
            self._wg_open = False

    def _flush_ln(self) -> None:
        """A deferred LayerNorm finish that no Linear picked up (the next kernel is another LayerNorm backward, or the backward ends)."""
        import ctypes as C
        if self._pending_ln is not None:
            job, self._pending_ln = self._pending_ln, None
            L.check(L.lib().ufnd_row_partials_finish(C.byref(job[0]), 0, L.stream_ptr(self.enc.device)), "ufnd_row_partials_finish")
            job[1]["ln_busy"][job[2]] = None            # (stream order protects the buffer)

    def _ln_bwd(self, sc: dict, x, ldx, gamma, dy, dx_f32, dx_bf16, lddx, dgamma, dbeta, M, add=None):
        H = self.enc.hidden
        self._flush_ln()                   # one pending job at a time: a LayerNorm backward right behind another one finishes the first here
        kk = sc["ln_flip"]
        sc["ln_flip"] ^= 1
        if sc["ln_busy"][kk] is not None:  # the finish (on the weight-gradient stream) that read this workspace last
            torch.cuda.current_stream(x.device).wait_event(sc["ln_busy"][kk])
            sc["ln_busy"][kk] = None
        ws = sc["ln"][kk]
        L.check(L.lib().ufnd_layernorm_bwd(x.data_ptr(), ldx, gamma.data_ptr(), dy.data_ptr(), dy.stride(0), L.ptr(add), add.stride(0) if add is not None else 0,
                                           L.ptr(dx_f32), L.ptr(dx_bf16), lddx, L.ptr(dgamma), L.ptr(dbeta), ws.data_ptr(), L.PARTIALS_DEFER, M, H,
                                           self.enc.eps, L.stream_ptr(x.device)), "ufnd_layernorm_bwd")
        job = L.PartialsJob()
        job.part, job.nblk, job.H, job.out0, job.out1 = ws.data_ptr(), L.lib().ufnd_layernorm_bwd_blocks(M), H, dgamma.data_ptr(), dbeta.data_ptr()
        self._pending_ln = (job, sc, kk)

    def _attn_bwd(self, qkv, ctx, dctx, lse, mask, dqkv, ws, B, Lq):
        L.check(L.lib().ufnd_attention_bf16_bwd(qkv.data_ptr(), ctx.data_ptr(), dctx.data_ptr(), lse.data_ptr(), L.ptr(mask), dqkv.data_ptr(), ws.data_ptr(),
                                                B, Lq, self.enc.heads, L.stream_ptr(qkv.device)), "ufnd_attention_bf16_bwd")


def _act(x: torch.Tensor, out: torch.Tensor, act: int) -> None:
    """bf16 activation over a whole tensor (training forward: FFN1 keeps its pre-activations AND their activation)."""
    L.check(L.lib().ufnd_act_bf16(x.data_ptr(), out.data_ptr(), x.numel(), act, L.stream_ptr(x.device)), "ufnd_act_bf16")


# =============================================================================================
class TextBackprop(_Backprop):
    """BertTextEncoder with a backward: BertModel (post-LN) -> masked mean-pool -> L2."""

    def _lk_build(self, i: int) -> dict:
        P = f"encoder.layer.{i}."
        return {"qkv_w": [P + f"attention.self.{n}.weight" for n in ("query", "key", "value")],
                "qkv_b": [P + f"attention.self.{n}.bias" for n in ("query", "key", "value")],
                "o_w": [P + "attention.output.dense.weight"], "o_b": [P + "attention.output.dense.bias"],
                "g1": [P + "attention.output.LayerNorm.weight"], "b1n": [P + "attention.output.LayerNorm.bias"],
                "w1": [P + "intermediate.dense.weight"], "b1": [P + "intermediate.dense.bias"],
                "w2": [P + "output.dense.weight"], "b2": [P + "output.dense.bias"],
                "g2": [P + "output.LayerNorm.weight"], "b2n": [P + "output.LayerNorm.bias"]}

    def groups(self):
        w, out = self.enc._w, []
        for i in reversed(range(self.enc.layers)):
            k = self._lk(i)
            for name in ("g2", "b2n", "w2", "b2", "w1", "b1", "g1", "b1n", "o_w", "o_b", "qkv_w", "qkv_b"):     # gradient-ready order inside the layer
                out.append([(key, tuple(w[key].shape)) for key in k[name]])
        for key in ("embeddings.LayerNorm.weight", "embeddings.LayerNorm.bias", "embeddings.position_embeddings.weight",
                    "embeddings.token_type_embeddings.weight", "embeddings.word_embeddings.weight"):
            out.append([(key, tuple(w[key].shape))])
        return out

    def linears(self):
        d = {}
        for i in range(self.enc.layers):
            k = self._lk(i)
            d[f"{i}.qkv"], d[f"{i}.o"], d[f"{i}.w1"], d[f"{i}.w2"] = (k["qkv_w"], k["qkv_b"]), (k["o_w"], k["o_b"]), (k["w1"], k["b1"]), (k["w2"], k["b2"])
        return d

    def _save_bufs(self, B: int, Lq: int) -> dict:
        key = ("save", B, Lq)
        if key not in self._scratch:
            e, dev = self.enc, self.enc.device
            M, H, I = B * Lq, e.hidden, e.inter
            bf, f32 = dict(dtype=torch.bfloat16, device=dev), dict(dtype=torch.float32, device=dev)
            layers = [{"xb": torch.empty(M, H, **bf), "qkv": torch.empty(M, 3 * H, **bf), "ctx": torch.empty(M, H, **bf),
                       "lse": torch.empty(M, e.heads, **f32), "y1": torch.empty(M, H, **f32), "x1b": torch.empty(M, H, **bf),
                       "pre": torch.empty(M, I, **bf), "h": torch.empty(M, I, **bf), "y2": torch.empty(M, H, **f32)} for _ in range(e.layers)]
            self._scratch[key] = {"layers": layers, "s": torch.empty(M, H, **f32), "xf": torch.empty(M, H, **f32), "x1f": torch.empty(M, H, **f32),
                                  "xb_last": torch.empty(M, H, **bf), "hid": torch.empty(M, H, **f32), "feat": torch.empty(B, H, **f32),
                                  # backward
                                  "dx": torch.empty(M, H, **f32), "dyf": torch.empty(M, H, **f32), "dyb": torch.empty(M, H, **bf), "dx1": torch.empty(M, H, **f32),
                                  "dpre": torch.empty(M, I, **bf), "dctx": torch.empty(M, H, **bf), "dqkv": torch.empty(M, 3 * H, **bf),
                                  "aws": torch.empty(L.lib().ufnd_attention_bwd_workspace_floats(B, Lq, e.heads), **f32), "ds": torch.empty(M, H, **f32)}
        return self._scratch[key]

    @torch.no_grad()
    def forward_train(self, input_ids: torch.Tensor, attention_mask: torch.Tensor) -> torch.Tensor:
        e = self.enc
        e._require_hip()
        if not self._ops:
            self.refresh_operands()
        dev = e.device
        B, Lq = input_ids.shape
        ids = input_ids.to(dev, torch.int64).contiguous()
        mask = attention_mask.to(dev, torch.int32).contiguous()
        sv = self._save_bufs(B, Lq)
        M, H, w, s = B * Lq, e.hidden, e._w, L.stream_ptr(dev)
        lib = L.lib()
        # embeddings: the raw sums are kept (their LayerNorm's backward needs its input)
        L.check(lib.ufnd_bert_embed(ids.data_ptr(), w["embeddings.word_embeddings.weight"].data_ptr(), w["embeddings.position_embeddings.weight"].data_ptr(),
                                    w["embeddings.token_type_embeddings.weight"].data_ptr(), None, None, None, sv["s"].data_ptr(), B, Lq, H, e.vocab, e.eps, s),
                "ufnd_bert_embed")
        x_f, x_b = sv["xf"], sv["layers"][0]["xb"]
        e._ln(sv["s"], H, w["embeddings.LayerNorm.weight"], w["embeddings.LayerNorm.bias"], x_b, x_f, M, H, e.eps)
        for i, a in enumerate(sv["layers"]):
            k = self._lk(i)
            wqkv, wo, w1, w2 = self._ops[f"{i}.qkv"][0], self._ops[f"{i}.o"][0], self._ops[f"{i}.w1"][0], self._ops[f"{i}.w2"][0]
            self._gemm(a["xb"], wqkv, self.master(k["qkv_b"]), out_bf16=a["qkv"])
            L.check(lib.ufnd_attention_bf16_lse(a["qkv"].data_ptr(), mask.data_ptr(), a["ctx"].data_ptr(), a["lse"].data_ptr(), B, Lq, e.heads, s), "ufnd_attention_bf16_lse")
            self._gemm(a["ctx"], wo, self.master(k["o_b"]), out_f32=a["y1"], residual=x_f)
            e._ln(a["y1"], H, self.master(k["g1"]), self.master(k["b1n"]), a["x1b"], sv["x1f"], M, H, e.eps)
            self._gemm(a["x1b"], w1, self.master(k["b1"]), out_bf16=a["pre"])
            _act(a["pre"], a["h"], ACT_GELU)
            self._gemm(a["h"], w2, self.master(k["b2"]), out_f32=a["y2"], residual=sv["x1f"])
            nxt_b = sv["layers"][i + 1]["xb"] if i + 1 < e.layers else sv["xb_last"]
            e._ln(a["y2"], H, self.master(k["g2"]), self.master(k["b2n"]), nxt_b, x_f, M, H, e.eps)
        sv["hid"].copy_(x_f)
        L.check(lib.ufnd_masked_meanpool_l2(sv["hid"].data_ptr(), mask.data_ptr(), sv["feat"].data_ptr(), B, Lq, H, s), "ufnd_masked_meanpool_l2")
        self.saved = {"B": B, "L": Lq, "ids": ids, "mask": mask, "sv": sv}
        return sv["feat"]

    @torch.no_grad()
    def backward(self, dfeat: torch.Tensor) -> None:
        """Gradients of every encoder parameter (into the arena's gradient buffer) from d loss / d features (B, H)."""
        e, st = self.enc, self.saved
        if st is None:
            raise RuntimeError("backward() without forward_train()")
        B, Lq, ids, mask, sv = st["B"], st["L"], st["ids"], st["mask"], st["sv"]
        M, H, I = B * Lq, e.hidden, e.inter
        sc = self._bwd_scratch(M, (H, 3 * H, I))
        s = L.stream_ptr(e.device)
        dfeat = L.f32c(dfeat)
        L.check(L.lib().ufnd_masked_meanpool_l2_bwd(sv["hid"].data_ptr(), mask.data_ptr(), dfeat.data_ptr(), sv["dx"].data_ptr(), B, Lq, H, s),
                "ufnd_masked_meanpool_l2_bwd")
        dx = sv["dx"]
        for i in reversed(range(e.layers)):
            a, k = sv["layers"][i], self._lk(i)
            ops = {n: self._ops[f"{i}.{n}"] for n in ("qkv", "o", "w1", "w2")}
            # output.LayerNorm, output.dense, GELU, intermediate.dense
            self._ln_bwd(sc, a["y2"], H, self.master(k["g2"]), dx, sv["dyf"], sv["dyb"], H, self.grad(k["g2"]), self.grad(k["b2n"]), M)
            self._wgrad(sc, sv["dyb"], a["h"], self.grad(k["w2"]), self.grad(k["b2"]))
            self._dgrad(sv["dyb"], ops["w2"][1], out_bf16=sv["dpre"], aux=a["pre"], act=ACT_GELU_BWD)
            self._wgrad(sc, sv["dpre"], a["x1b"], self.grad(k["w1"]), self.grad(k["b1"]))
            self._dgrad(sv["dpre"], ops["w1"][1], out_f32=sv["dx1"], residual=sv["dyf"])
            # attention.output.LayerNorm, attention.output.dense, attention, q/k/v
            self._ln_bwd(sc, a["y1"], H, self.master(k["g1"]), sv["dx1"], sv["dyf"], sv["dyb"], H, self.grad(k["g1"]), self.grad(k["b1n"]), M)
            self._wgrad(sc, sv["dyb"], a["ctx"], self.grad(k["o_w"]), self.grad(k["o_b"]))
            self._dgrad(sv["dyb"], ops["o"][1], out_bf16=sv["dctx"])
            self._attn_bwd(a["qkv"], a["ctx"], sv["dctx"], a["lse"], mask, sv["dqkv"], sv["aws"], B, Lq)
            self._wgrad(sc, sv["dqkv"], a["xb"], self.grad(k["qkv_w"]), self.grad(k["qkv_b"]))
            self._dgrad(sv["dqkv"], ops["qkv"][1], out_f32=dx, residual=sv["dyf"])
        # embeddings: LayerNorm backward to the raw sums, then the three tables
        ek = "embeddings."
        self._ln_bwd(sc, sv["s"], H, self.master([ek + "LayerNorm.weight"]), dx, sv["ds"], None, H, self.grad([ek + "LayerNorm.weight"]),
                     self.grad([ek + "LayerNorm.bias"]), M)
        L.check(L.lib().ufnd_bert_embed_bwd(ids.data_ptr(), sv["ds"].data_ptr(), self.grad([ek + "word_embeddings.weight"]).data_ptr(),
                                            self.grad([ek + "position_embeddings.weight"]).data_ptr(), self.grad([ek + "token_type_embeddings.weight"]).data_ptr(),
                                            B, Lq, H, e.vocab, e.max_position, self.master([ek + "token_type_embeddings.weight"]).shape[0], s), "ufnd_bert_embed_bwd")
        self._flush_ln()                   # (the embedding LayerNorm's dgamma / dbeta: no Linear follows it)
        self.join_wgrad()


# =============================================================================================
class VisualBackprop(_Backprop):
    """ClipVisualEncoder with a backward: CLIP ViT (pre-LN) -> pooled CLS -> projection -> frame pooling."""

    V = "vision_model."

    def _lk_build(self, i: int) -> dict:
        P = self.V + f"encoder.layers.{i}."
        return {"qkv_w": [P + f"self_attn.{n}.weight" for n in ("q_proj", "k_proj", "v_proj")],
                "qkv_b": [P + f"self_attn.{n}.bias" for n in ("q_proj", "k_proj", "v_proj")],
                "o_w": [P + "self_attn.out_proj.weight"], "o_b": [P + "self_attn.out_proj.bias"],
                "g1": [P + "layer_norm1.weight"], "b1n": [P + "layer_norm1.bias"],
                "w1": [P + "mlp.fc1.weight"], "b1": [P + "mlp.fc1.bias"], "w2": [P + "mlp.fc2.weight"], "b2": [P + "mlp.fc2.bias"],
                "g2": [P + "layer_norm2.weight"], "b2n": [P + "layer_norm2.bias"]}

    def groups(self):
        w, V, out = self.enc._w, self.V, []
        for key in ("visual_projection.weight", V + "post_layernorm.weight", V + "post_layernorm.bias"):
            out.append([(key, tuple(w[key].shape))])
        for i in reversed(range(self.enc.layers)):
            k = self._lk(i)
            for name in ("w2", "b2", "w1", "b1", "g2", "b2n", "o_w", "o_b", "qkv_w", "qkv_b", "g1", "b1n"):
                out.append([(key, tuple(w[key].shape)) for key in k[name]])
        for key in (V + "pre_layrnorm.weight", V + "pre_layrnorm.bias", V + "embeddings.position_embedding.weight", V + "embeddings.class_embedding",
                    V + "embeddings.patch_embedding.weight"):
            out.append([(key, tuple(w[key].shape))])
        return out

    def linears(self):
        d = {"proj": (["visual_projection.weight"], None), "patch": ([self.V + "embeddings.patch_embedding.weight"], None)}
        for i in range(self.enc.layers):
            k = self._lk(i)
            d[f"{i}.qkv"], d[f"{i}.o"], d[f"{i}.w1"], d[f"{i}.w2"] = (k["qkv_w"], k["qkv_b"]), (k["o_w"], k["o_b"]), (k["w1"], k["b1"]), (k["w2"], k["b2"])
        return d

    def _save_bufs(self, B: int, Fr: int) -> dict:
        key = ("save", B, Fr)
        if key not in self._scratch:
            e, dev = self.enc, self.enc.device
            N, T, H, I = B * Fr, e.n_patches + 1, e.hidden, e.inter
            M, NP = N * T, N * e.n_patches
            bf, f32 = dict(dtype=torch.bfloat16, device=dev), dict(dtype=torch.float32, device=dev)
            layers = [{"xin": torch.empty(M, H, **f32), "h1b": torch.empty(M, H, **bf), "qkv": torch.empty(M, 3 * H, **bf), "ctx": torch.empty(M, H, **bf),
                       "lse": torch.empty(M, e.heads, **f32), "xmid": torch.empty(M, H, **f32), "h2b": torch.empty(M, H, **bf),
                       "pre": torch.empty(M, I, **bf), "m": torch.empty(M, I, **bf)} for _ in range(e.layers)]
            Np = _pad64(N)
            self._scratch[key] = {"layers": layers, "patches": torch.empty(NP, 3 * e.patch ** 2, **bf), "pe": torch.empty(NP, H, **f32),
                                  "s": torch.empty(M, H, **f32), "xout": torch.empty(M, H, **f32), "pooled_b": torch.zeros(Np, H, **bf),
                                  "pooled_f": torch.empty(N, H, **f32), "e": torch.empty(N, e.proj, **f32), "feat": torch.empty(B, e.proj, **f32),
                                  # backward
                                  "de": torch.empty(N, e.proj, **f32), "de_b": torch.zeros(Np, e.proj, **bf), "dpool": torch.empty(Np, H, **f32),
                                  "dx": torch.empty(M, H, **f32), "dxb": torch.empty(M, H, **bf), "dh": torch.empty(M, H, **f32),
                                  "dmid": torch.empty(M, H, **f32), "dmidb": torch.empty(M, H, **bf), "dpre": torch.empty(M, I, **bf),
                                  "dctx": torch.empty(M, H, **bf), "dqkv": torch.empty(M, 3 * H, **bf), "ds": torch.empty(M, H, **f32),
                                  "dpe": torch.empty(NP, H, **bf), "aws": torch.empty(L.lib().ufnd_attention_bwd_workspace_floats(N, T, e.heads), **f32)}
        return self._scratch[key]

    @torch.no_grad()
    def forward_train(self, frames: torch.Tensor) -> torch.Tensor:
        e = self.enc
        e._require_hip()
        if not self._ops:
            self.refresh_operands()
        if frames.dim() == 4:
            frames = frames[:, None]
        dev = e.device
        B, Fr = frames.shape[:2]
        if tuple(frames.shape[2:]) != (3, e.image, e.image):
            raise RuntimeError(f"frames: expected (B,F,3,{e.image},{e.image}), got {tuple(frames.shape)}")
        fr = L.f32c(frames.to(dev)).view(B * Fr, 3, e.image, e.image)
        sv = self._save_bufs(B, Fr)
        N, T, H, w, V = B * Fr, e.n_patches + 1, e.hidden, e._w, self.V
        M, s, lib = N * T, L.stream_ptr(dev), L.lib()
        L.check(lib.ufnd_vit_patchify(fr.data_ptr(), sv["patches"].data_ptr(), N, e.image, e.patch, s), "ufnd_vit_patchify")
        self._gemm(sv["patches"], self._ops["patch"][0], None, out_f32=sv["pe"])
        L.check(lib.ufnd_vit_assemble(sv["pe"].data_ptr(), w[V + "embeddings.class_embedding"].data_ptr(), w[V + "embeddings.position_embedding.weight"].data_ptr(),
                                      None, None, sv["s"].data_ptr(), None, None, N, e.n_patches, H, e.eps, s), "ufnd_vit_assemble")
        x = sv["layers"][0]["xin"]
        e._ln(sv["s"], H, w[V + "pre_layrnorm.weight"], w[V + "pre_layrnorm.bias"], None, x, M, H, e.eps)
        for i, a in enumerate(sv["layers"]):
            k = self._lk(i)
            wqkv, wo, w1, w2 = self._ops[f"{i}.qkv"][0], self._ops[f"{i}.o"][0], self._ops[f"{i}.w1"][0], self._ops[f"{i}.w2"][0]
            e._ln(a["xin"], H, self.master(k["g1"]), self.master(k["b1n"]), a["h1b"], None, M, H, e.eps)
            self._gemm(a["h1b"], wqkv, self.master(k["qkv_b"]), out_bf16=a["qkv"])
            L.check(lib.ufnd_attention_bf16_lse(a["qkv"].data_ptr(), None, a["ctx"].data_ptr(), a["lse"].data_ptr(), N, T, e.heads, s), "ufnd_attention_bf16_lse")
            self._gemm(a["ctx"], wo, self.master(k["o_b"]), out_f32=a["xmid"], residual=a["xin"])
            e._ln(a["xmid"], H, self.master(k["g2"]), self.master(k["b2n"]), a["h2b"], None, M, H, e.eps)
            self._gemm(a["h2b"], w1, self.master(k["b1"]), out_bf16=a["pre"])
            _act(a["pre"], a["m"], ACT_QUICK_GELU)
            nxt = sv["layers"][i + 1]["xin"] if i + 1 < e.layers else sv["xout"]
            self._gemm(a["m"], w2, self.master(k["b2"]), out_f32=nxt, residual=a["xmid"])
        e._ln(sv["xout"], T * H, w[V + "post_layernorm.weight"], w[V + "post_layernorm.bias"], sv["pooled_b"], sv["pooled_f"], N, H, e.eps)
        self._gemm(sv["pooled_b"][:N], self._ops["proj"][0], None, out_f32=sv["e"])
        L.check(lib.ufnd_l2norm_frames(sv["e"].data_ptr(), sv["feat"].data_ptr(), B, Fr, e.proj, s), "ufnd_l2norm_frames")
        self.saved = {"B": B, "F": Fr, "sv": sv}
        return sv["feat"]

    @torch.no_grad()
    def backward(self, dfeat: torch.Tensor) -> None:
        e, st = self.enc, self.saved
        if st is None:
            raise RuntimeError("backward() without forward_train()")
        B, Fr, sv, V = st["B"], st["F"], st["sv"], self.V
        N, T, H, I = B * Fr, e.n_patches + 1, e.hidden, e.inter
        M, NP = N * T, N * e.n_patches
        sc = self._bwd_scratch(M, (H, 3 * H, I))
        scp = self._bwd_scratch(NP, (H, 3 * e.patch ** 2))
        sch = self._bwd_scratch(N, (e.proj, H))
        s, lib = L.stream_ptr(e.device), L.lib()
        dfeat = L.f32c(dfeat)
        # frame pooling, projection (bias-free), post-LayerNorm on the CLS rows
        L.check(lib.ufnd_l2norm_frames_bwd(sv["e"].data_ptr(), dfeat.data_ptr(), sv["de"].data_ptr(), B, Fr, e.proj, s), "ufnd_l2norm_frames_bwd")
        sv["de_b"][:N].copy_(sv["de"])
        self._wgrad(sch, sv["de_b"][:N], sv["pooled_b"][:N], self.grad(["visual_projection.weight"]), None)
        self._dgrad(sv["de_b"][:N], self._ops["proj"][1], out_f32=sv["dpool"][:N])
        dx = sv["dx"]
        dx.zero_()                       # only the CLS rows of the last layer's output carry a gradient
        sv["dxb"].zero_()
        self._ln_bwd(sch, sv["xout"], T * H, self.master([V + "post_layernorm.weight"]), sv["dpool"][:N], dx, sv["dxb"], T * H,
                     self.grad([V + "post_layernorm.weight"]), self.grad([V + "post_layernorm.bias"]), N)
        for i in reversed(range(e.layers)):
            a, k = sv["layers"][i], self._lk(i)
            ops = {n: self._ops[f"{i}.{n}"] for n in ("qkv", "o", "w1", "w2")}
            # MLP branch: x_out = x_mid + fc2(quick_gelu(fc1(LN2(x_mid))))
            self._wgrad(sc, sv["dxb"], a["m"], self.grad(k["w2"]), self.grad(k["b2"]))
            self._dgrad(sv["dxb"], ops["w2"][1], out_bf16=sv["dpre"], aux=a["pre"], act=ACT_QUICK_GELU_BWD)
            self._wgrad(sc, sv["dpre"], a["h2b"], self.grad(k["w1"]), self.grad(k["b1"]))
            self._dgrad(sv["dpre"], ops["w1"][1], out_f32=sv["dh"])
            self._ln_bwd(sc, a["xmid"], H, self.master(k["g2"]), sv["dh"], sv["dmid"], sv["dmidb"], H, self.grad(k["g2"]), self.grad(k["b2n"]), M, add=dx)
            # attention branch: x_mid = x_in + out_proj(attn(qkv(LN1(x_in))))
            self._wgrad(sc, sv["dmidb"], a["ctx"], self.grad(k["o_w"]), self.grad(k["o_b"]))
            self._dgrad(sv["dmidb"], ops["o"][1], out_bf16=sv["dctx"])
            self._attn_bwd(a["qkv"], a["ctx"], sv["dctx"], a["lse"], None, sv["dqkv"], sv["aws"], N, T)
            self._wgrad(sc, sv["dqkv"], a["h1b"], self.grad(k["qkv_w"]), self.grad(k["qkv_b"]))
            self._dgrad(sv["dqkv"], ops["qkv"][1], out_f32=sv["dh"])
            self._ln_bwd(sc, a["xin"], H, self.master(k["g1"]), sv["dh"], dx, sv["dxb"], H, self.grad(k["g1"]), self.grad(k["b1n"]), M, add=sv["dmid"])
        # pre-LayerNorm, token assembly, patch embedding
        self._ln_bwd(sc, sv["s"], H, self.master([V + "pre_layrnorm.weight"]), dx, sv["ds"], None, H, self.grad([V + "pre_layrnorm.weight"]),
                     self.grad([V + "pre_layrnorm.bias"]), M)
        L.check(lib.ufnd_vit_assemble_bwd(sv["ds"].data_ptr(), self.grad([V + "embeddings.class_embedding"]).data_ptr(),
                                          self.grad([V + "embeddings.position_embedding.weight"]).data_ptr(), sv["dpe"].data_ptr(), N, e.n_patches, H, s),
                "ufnd_vit_assemble_bwd")
        self._wgrad(scp, sv["dpe"], sv["patches"], self.grad([V + "embeddings.patch_embedding.weight"]), None)
        self._flush_ln()
        self.join_wgrad()
