"""TemporalSyncNet -- MI355X-native mirror of the vector-level alignment the cache builder uses
(src/core_blocks/temporal_blocks.py:47-140): `align(text_vec, visual_vec) -> np.ndarray[out_dim]`
produces the `temporal (N,256)` input of the fusion step (fakesv_dataset.py:176).  Same
constructor arguments and `state_dict` keys (`proj.{0,3}.{weight,bias}`); the weights are never
trained in the reference (random init), so this is a fixed random projection.  `align_batch` is the batched
device-to-device form used inside the step.  Like the reference's, `align` honours `self.training`: its
`torch.inference_mode` decorator switches autograd off, not dropout, and the cache builder never calls `.eval()`, so
the projection's Dropout(0.1) is live there (mask from this module's own counter-based stream); call `.eval()` for
the deterministic projection the golden fixtures pin.

`use_tcn=True` adds the optional sequence path (:16-43 `_TinyTCN`, :141-157 `forward(text_seq, vis_seq)`): dilated
Conv1d -> BatchNorm1d -> GELU -> dropout blocks with residuals, mean+max pooling over time and a Linear head, run by
`ufnd_tcn_forward` (frames stay rows, every conv is an unfold + fp32 MFMA GEMM).  Same sub-module names and
`state_dict` keys (`tcn.convs.i.*`, `tcn.norms.i.*`, `head.*`) and, under the same torch seed, the same initial
weights.  Forward only, like everything else in this module.  `delay_score` / `estimate_av_lag` (:162-226) are the
reference's host-side helpers, restated in numpy."""
from __future__ import annotations

from typing import Dict, Union

import numpy as np
import torch
import torch.nn as nn

from . import _lib as L
from .state import StepStateBuffer


class _TinyTCN(nn.Module):
    """Parameter holder with the reference's layout (temporal_blocks.py:16-30); the compute is ufnd_tcn_forward."""

    def __init__(self, in_ch: int, hid: int = 128, layers: int = 2, k: int = 3, dropout: float = 0.1):
        super().__init__()
        self.convs, self.norms = nn.ModuleList(), nn.ModuleList()
        self.drop = nn.Dropout(dropout)
        ch = in_ch
        for i in range(layers):
            self.convs.append(nn.Conv1d(ch, hid, kernel_size=k, padding="same", dilation=2 ** i))
            self.norms.append(nn.BatchNorm1d(hid))
            ch = hid
        self.in_ch, self.hid, self.k = int(in_ch), int(hid), int(k)

    def forward(self, x):
        raise L.UltrafndHipError("_TinyTCN has no stand-alone forward here: call TemporalSyncNet.forward(text_seq, vis_seq)")


class TemporalSyncNet(nn.Module):
    def __init__(self, in_dim: int = 768, out_dim: int = 256, use_tcn: bool = False, tcn_hid: int = 128,
                 tcn_layers: int = 2, tcn_kernel: int = 3, dropout: float = 0.1):
        super().__init__()
        self.in_dim, self.out_dim = int(in_dim), int(out_dim)
        self.proj = nn.Sequential(nn.Linear(4 * self.in_dim + 1, 2 * self.out_dim), nn.GELU(), nn.Dropout(dropout),
                                  nn.Linear(2 * self.out_dim, self.out_dim))
        self.use_tcn = bool(use_tcn)
        if self.use_tcn:                                  # construction (and RNG) order of the reference, :87-94
            if tcn_hid % 32 or self.out_dim % 32:
                raise L.UltrafndHipError(f"sequence path: tcn_hid={tcn_hid} and out_dim={out_dim} must be multiples of 32")
            self.tcn = _TinyTCN(in_ch=self.in_dim, hid=tcn_hid, layers=tcn_layers, k=tcn_kernel, dropout=dropout)
            self.head = nn.Linear(tcn_hid * 2, self.out_dim)
        else:
            self.tcn = None
            self.head = None
        self._seq_packed = None
        self._seq_seed = (torch.initial_seed() ^ 0x7463_6E5F_6472_6F70) & 0x7FFF_FFFF_FFFF_FFFF     # no draw from the global RNG
        self._seq_calls = 0
        for p in self.parameters():
            p.requires_grad_(False)
        self._packed = None
        self._ws: Dict[int, torch.Tensor] = {}
        self._align_state = None

    def _apply(self, fn, recurse=True):
        out = super()._apply(fn, recurse)
        self._packed = None
        self._seq_packed = None
        self._ws.clear()
        self._align_state = None
        return out

    def load_state_dict(self, *a, **kw):
        self._packed = None
        self._seq_packed = None
        return super().load_state_dict(*a, **kw)

    def _pack(self):
        if self._packed is None:
            w0 = self.proj[0].weight.detach()
            ld = L.lib().ufnd_temporal_weight_ld(self.in_dim)
            w0p = torch.zeros(w0.shape[0], ld, dtype=torch.float32, device=w0.device)
            w0p[:, :w0.shape[1]].copy_(w0)
            self._packed = (w0p, self.proj[0].bias.detach().float().contiguous(), self.proj[3].weight.detach().float().contiguous(),
                            self.proj[3].bias.detach().float().contiguous())
        return self._packed

    @torch.no_grad()
    def align_batch(self, text: torch.Tensor, visual: torch.Tensor, out: torch.Tensor = None, training: bool = None) -> torch.Tensor:
        """(B,in_dim), (B,Dv) device tensors -> (B,out_dim) device tensor (written into `out` when given).
        `training` overrides `self.training` for this call (the trainer evaluates its validation / test splits with the
        deterministic projection whatever mode the module was left in)."""
        dev = self.proj[0].weight.device
        if dev.type != "cuda":
            raise L.UltrafndHipError("TemporalSyncNet runs on a HIP device only: call .to('cuda') (no CPU fallback)")
        t, v = L.f32c(text.to(dev)), L.f32c(visual.to(dev))
        B = t.shape[0]
        if t.shape[1] != self.in_dim or v.shape[0] != B:
            raise RuntimeError(f"align: expected text (B,{self.in_dim}) and visual (B,Dv), got {tuple(t.shape)}, {tuple(v.shape)}")
        w0, b0, w3, b3 = self._pack()
        if B not in self._ws:
            n = L.lib().ufnd_temporal_workspace_floats(B, self.in_dim, 2 * self.out_dim)
            self._ws[B] = torch.empty(n, dtype=torch.float32, device=dev)
        if out is None:
            out = torch.empty(B, self.out_dim, dtype=torch.float32, device=dev)
        elif tuple(out.shape) != (B, self.out_dim) or out.dtype != torch.float32 or not out.is_contiguous() or out.device != dev:
            raise RuntimeError(f"align_batch: out must be a contiguous fp32 ({B},{self.out_dim}) tensor on {dev}")
        p = float(self.proj[2].p) if (self.training if training is None else training) else 0.0
        st = None
        if p > 0.0:                  # device-resident counter: every call draws a fresh mask (stream-ordered, no host copy)
            if self._align_state is None or self._align_state.device != dev:
                self._align_state = StepStateBuffer(dev, seed=self._seq_seed ^ 0x616C_6967_6E)
            st = self._align_state
        L.check(L.lib().ufnd_temporal_align(t.data_ptr(), v.data_ptr(), w0.data_ptr(), b0.data_ptr(), w3.data_ptr(), b3.data_ptr(),
                                            self._ws[B].data_ptr(), out.data_ptr(), B, self.in_dim, v.shape[1], 2 * self.out_dim,
                                            self.out_dim, p, st.ptr if st is not None else None, L.stream_ptr(dev)), "ufnd_temporal_align")
        if st is not None:
            st.advance()
        return out

    def align(self, text_vec: Union[np.ndarray, torch.Tensor], visual_vec: Union[np.ndarray, torch.Tensor]) -> np.ndarray:
        """Reference signature: single vectors in, np.float32[out_dim] out."""
        def as2d(x):
            if isinstance(x, np.ndarray):
                x = torch.from_numpy(x)
            elif not isinstance(x, torch.Tensor):
                raise TypeError("text_vec / visual_vec must be np.ndarray or torch.Tensor")
            return x.unsqueeze(0) if x.dim() == 1 else x
        return self.align_batch(as2d(text_vec), as2d(visual_vec)).cpu().numpy()[0].astype(np.float32)

    # ------------------------------------------------------------------ sequence path (:141-157)
    def _pack_seq(self):
        """Conv1d weights (hid, ch, k) -> tap-major (hid, ld) rows; BatchNorm tensors are used in place (running
        statistics are written by the train-mode forward)."""
        if self._seq_packed is None:
            ws = []
            for conv in self.tcn.convs:
                w = conv.weight.detach().float()
                hid, ch, k = w.shape
                ld = L.lib().ufnd_tcn_weight_ld(ch, k)
                wp = torch.zeros(hid, ld, dtype=torch.float32, device=w.device)
                wp[:, :ch * k].copy_(w.permute(0, 2, 1).reshape(hid, k * ch))
                ws.append(wp)
            self._seq_packed = ws
        return self._seq_packed

    @torch.no_grad()
    def forward(self, text_seq: torch.Tensor, vis_seq: torch.Tensor) -> torch.Tensor:
        """text_seq (B,T,Dt), vis_seq (B,T,Dv) with Dt + Dv == in_dim -> (B,out_dim) device tensor.  `self.training`
        selects batch statistics + dropout (the nn.Module default, as in the reference) or running statistics."""
        assert self.use_tcn, "Enable use_tcn=True to use the sequence path."
        dev = self.proj[0].weight.device
        if dev.type != "cuda":
            raise L.UltrafndHipError("TemporalSyncNet runs on a HIP device only: call .to('cuda') (no CPU fallback)")
        t, v = L.f32c(text_seq.to(dev)), L.f32c(vis_seq.to(dev))
        if t.dim() != 3 or v.dim() != 3 or t.shape[:2] != v.shape[:2]:
            raise RuntimeError(f"forward: expected text_seq (B,T,Dt) and vis_seq (B,T,Dv), got {tuple(t.shape)}, {tuple(v.shape)}")
        B, T, Dt = t.shape
        Dv = v.shape[2]
        if Dt + Dv != self.tcn.in_ch:                     # nn.Conv1d's own complaint in the reference
            raise RuntimeError(f"forward: expected input to have {self.tcn.in_ch} channels, but got {Dt + Dv} channels instead")
        if B * T == 0:
            raise RuntimeError("forward: empty batch / sequence")
        train = bool(self.training)
        if train and B * T == 1:
            raise ValueError("Expected more than 1 value per channel when training")
        hid, k, n = self.tcn.hid, self.tcn.k, len(self.tcn.convs)
        packed = self._pack_seq()
        layers = (L.TcnLayer * n)()
        for i, (conv, bn) in enumerate(zip(self.tcn.convs, self.tcn.norms)):
            layers[i].w, layers[i].b = packed[i].data_ptr(), conv.bias.data_ptr()
            layers[i].gamma, layers[i].beta = bn.weight.data_ptr(), bn.bias.data_ptr()
            layers[i].running_mean, layers[i].running_var = bn.running_mean.data_ptr(), bn.running_var.data_ptr()
        p = float(self.tcn.drop.p) if train else 0.0
        st = None
        if p > 0:
            self._seq_calls += 1
            st = StepStateBuffer(dev, seed=self._seq_seed)
            st.set_u64("step", self._seq_calls)
        ws = torch.empty(L.lib().ufnd_tcn_workspace_floats(B, T, self.tcn.in_ch, hid, k), dtype=torch.float32, device=dev)
        out = torch.empty(B, self.out_dim, dtype=torch.float32, device=dev)
        bn0 = self.tcn.norms[0]
        L.check(L.lib().ufnd_tcn_forward(t.data_ptr(), Dt, v.data_ptr(), Dv, B, T, layers, n, k, hid, self.head.weight.data_ptr(),
                                         self.head.bias.data_ptr(), self.out_dim, int(train), p,
                                         float(bn0.momentum if bn0.momentum is not None else 0.1), float(bn0.eps),
                                         st.ptr if st is not None else None, ws.data_ptr(), out.data_ptr(), L.stream_ptr(dev)),
                "ufnd_tcn_forward")
        if train:
            for bn in self.tcn.norms:
                bn.num_batches_tracked += 1
        return out

    # ------------------------------------------------------------------ host-side delay estimators (:162-226)
    @staticmethod
    def delay_score(audio_len: int, video_len: int) -> float:
        """|a - v| / max(1, a, v) on the non-negative lengths: 0 matched .. 1 mismatched (:162-171)."""
        a, v = float(max(0, audio_len)), float(max(0, video_len))
        return float(abs(a - v) / max(1.0, a, v))

    @staticmethod
    def estimate_av_lag(audio_envelope, mouth_open, sr: float = 16000.0, fps: float = 25.0, max_lag_s: float = 0.5) -> float:
        """Lag (seconds) of the peak of the circular cross-correlation of the two standardised 1-D envelopes inside
        +-max_lag_s (:173-226): truncated to the shorter input, zero-padded FFT of the next power of two >= 2L, lags
        ordered -(L-1)..L-1, searched in a window around the middle of that array.  < 4 samples -> 0.0."""
        def flat(x):
            if isinstance(x, torch.Tensor):
                x = x.detach().cpu().float().numpy()
            return np.asarray(x).astype(np.float32).ravel()
        a, m = flat(audio_envelope), flat(mouth_open)
        n = min(a.size, m.size)
        if n < 4:
            return 0.0
        a, m = a[:n], m[:n]
        a = (a - a.mean()) / (a.std() + 1e-9)
        m = (m - m.mean()) / (m.std() + 1e-9)
        nfft = 1 << max(0, int(2 * n - 1).bit_length())
        xc = np.fft.irfft(np.fft.rfft(a, nfft) * np.conj(np.fft.rfft(m, nfft)), nfft)
        lags = np.concatenate([xc[nfft - (n - 1):], xc[:n]])       # 2n-1 values, lag -(n-1) .. n-1
        mid = lags.size // 2
        half = int(max_lag_s * sr)
        lo, hi = max(0, mid - half), min(lags.size, mid + half + 1)
        return float((lo + int(np.argmax(lags[lo:hi])) - mid) / sr)
