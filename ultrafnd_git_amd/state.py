"""Device-resident `ufnd_step_state` (include/ultrafnd_hip.h): hyper-parameters, step counter,
dropout key and the step's scalar results live in HBM so that a captured hipGraph replays with
current values.  The host only touches it between epochs (lr) or to read results."""
from __future__ import annotations

from typing import Dict

import torch

from . import _lib as L

_OFF: Dict[str, int] = {n: getattr(L.StepState, n).offset for n, _ in L.StepState._fields_}


class StepStateBuffer:
    def __init__(self, device: torch.device, *, seed: int = 0, lr: float = 2e-4, weight_decay: float = 1e-4,
                 betas=(0.9, 0.999), eps: float = 1e-8, max_norm: float = 5.0, grad_scale: float = 1.0):
        st = L.StepState()
        st.step, st.seed = 0, int(seed) & 0xFFFFFFFFFFFFFFFF
        st.lr, st.weight_decay, st.beta1, st.beta2, st.eps = lr, weight_decay, betas[0], betas[1], eps
        st.max_norm, st.grad_scale = max_norm, grad_scale
        self.device = torch.device(device)
        self.buf = torch.frombuffer(bytearray(bytes(st)), dtype=torch.uint8).clone().to(self.device)

    @property
    def ptr(self) -> int:
        return self.buf.data_ptr()

    def set_float(self, name: str, value: float) -> None:
        src = torch.tensor([value], dtype=torch.float32).view(torch.uint8)
        self.buf[_OFF[name]:_OFF[name] + 4].copy_(src)

    def set_u64(self, name: str, value: int) -> None:
        src = torch.tensor([value], dtype=torch.int64).view(torch.uint8)
        self.buf[_OFF[name]:_OFF[name] + 8].copy_(src)

    def float_view(self, name: str) -> torch.Tensor:
        """0-d device tensor aliasing a float field (no sync)."""
        return self.buf[_OFF[name]:_OFF[name] + 4].view(torch.float32)[0]

    def read(self) -> L.StepState:
        """Synchronising device->host copy of the whole struct."""
        raw = bytes(self.buf.cpu().numpy().tobytes())
        return L.StepState.from_buffer_copy(raw)

    def advance(self) -> None:
        L.check(L.lib().ufnd_step_advance(self.ptr, L.stream_ptr(self.device)), "ufnd_step_advance")
