"""TemporalSyncNet -- MI355X-native mirror of the vector-level alignment the cache builder uses
(src/core_blocks/temporal_blocks.py:47-140): `align(text_vec, visual_vec) -> np.ndarray[out_dim]`
produces the `temporal (N,256)` input of the fusion step (fakesv_dataset.py:176).  Same
constructor arguments and `state_dict` keys (`proj.{0,3}.{weight,bias}`); the weights are never
trained in the reference (random init, inference_mode), so this is a fixed random projection.
`align_batch` is the batched device-to-device form used inside the step.  The optional TCN
sequence path (`use_tcn=True`, disabled by default in the reference's YAML) is not implemented."""
from __future__ import annotations

from typing import Dict, Union

import numpy as np
import torch
import torch.nn as nn

from . import _lib as L


class TemporalSyncNet(nn.Module):
    def __init__(self, in_dim: int = 768, out_dim: int = 256, use_tcn: bool = False, tcn_hid: int = 128,
                 tcn_layers: int = 2, tcn_kernel: int = 3, dropout: float = 0.1):
        super().__init__()
        if use_tcn:
            raise NotImplementedError("the sequence (TCN) path is outside the hot path; use_tcn=False is the reference's default")
        self.in_dim, self.out_dim = int(in_dim), int(out_dim)
        self.proj = nn.Sequential(nn.Linear(4 * self.in_dim + 1, 2 * self.out_dim), nn.GELU(), nn.Dropout(dropout),
                                  nn.Linear(2 * self.out_dim, self.out_dim))
        for p in self.parameters():
            p.requires_grad_(False)
        self._packed = None
        self._ws: Dict[int, torch.Tensor] = {}

    def _apply(self, fn, recurse=True):
        out = super()._apply(fn, recurse)
        self._packed = None
        self._ws.clear()
        return out

    def load_state_dict(self, *a, **kw):
        self._packed = None
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
    def align_batch(self, text: torch.Tensor, visual: torch.Tensor) -> torch.Tensor:
        """(B,in_dim), (B,Dv) device tensors -> (B,out_dim) device tensor."""
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
        out = torch.empty(B, self.out_dim, dtype=torch.float32, device=dev)
        L.check(L.lib().ufnd_temporal_align(t.data_ptr(), v.data_ptr(), w0.data_ptr(), b0.data_ptr(), w3.data_ptr(), b3.data_ptr(),
                                            self._ws[B].data_ptr(), out.data_ptr(), B, self.in_dim, v.shape[1], 2 * self.out_dim,
                                            self.out_dim, L.stream_ptr(dev)), "ufnd_temporal_align")
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
