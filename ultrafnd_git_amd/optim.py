"""clip_grad_norm_ + AdamW + StepLR over the flat arena (forensic_trainer.py:173-177,292-298,341).

Two streaming kernels replace the reference's ~104 per-tensor norms and its foreach AdamW:
`ufnd_grad_norm` (4 B/param) and `ufnd_adamw_step` (28 B/param), both over the arena's
contiguous with-grad range.  Hyper-parameters live in the device step state so a captured
graph sees lr changes.
"""
from __future__ import annotations

from typing import List

import torch

from . import _lib as L
from .arena import FlatArena
from .state import StepStateBuffer


class FusedAdamW:
    def __init__(self, arena: FlatArena, lr: float = 2e-4, weight_decay: float = 1e-4, betas=(0.9, 0.999),
                 eps: float = 1e-8, max_norm: float = 5.0, seed: int = 0, grad_scale: float = 1.0):
        if arena.device.type != "cuda":
            raise L.UltrafndHipError("FusedAdamW needs the parameter arena on a HIP device (no CPU fallback)")
        self.arena = arena
        self.state = StepStateBuffer(arena.device, seed=seed, lr=lr, weight_decay=weight_decay, betas=betas, eps=eps,
                                     max_norm=max_norm if max_norm else 0.0, grad_scale=grad_scale)
        self.param_groups: List[dict] = [{"lr": lr, "initial_lr": lr, "weight_decay": weight_decay, "betas": betas,
                                          "eps": eps}]
        self._partials = torch.empty(1024, dtype=torch.float32, device=arena.device)
        self.fused = True          # ufnd_clip_adamw_step (two launches) instead of norm + finalize + AdamW + advance (four)
        arena.ensure_grad()
        arena.ensure_moments()

    def zero_grad(self, set_to_none: bool = True) -> None:
        """Kept for API parity (forensic_trainer.py:290).  Gradients are overwritten by every
        backward, so there is nothing to clear."""

    def set_lr(self, lr: float) -> None:
        self.param_groups[0]["lr"] = lr
        self.state.set_float("lr", lr)

    def clip_and_step(self) -> None:
        """nn.utils.clip_grad_norm_(params, max_norm) followed by optim.step()."""
        a, dev = self.arena, self.arena.device
        s = L.stream_ptr(dev)
        lib = L.lib()
        if self.fused:       # two launches: sum of squares (+ step counter), then norm / clip / AdamW
            L.check(lib.ufnd_clip_adamw_step(a.data.data_ptr(), a.grad.data_ptr(), a.exp_avg.data_ptr(), a.exp_avg_sq.data_ptr(), a.n_grad,
                                             self._partials.data_ptr(), self.state.ptr, s), "ufnd_clip_adamw_step")
            return
        L.check(lib.ufnd_grad_norm(a.grad.data_ptr(), a.n_grad, self._partials.data_ptr(), self.state.ptr, s),
                "ufnd_grad_norm")
        L.check(lib.ufnd_adamw_step(a.data.data_ptr(), a.grad.data_ptr(), a.exp_avg.data_ptr(),
                                    a.exp_avg_sq.data_ptr(), a.n_grad, self.state.ptr, s), "ufnd_adamw_step")
        L.check(lib.ufnd_step_advance(self.state.ptr, s), "ufnd_step_advance")

    step = clip_and_step

    def state_dict(self) -> dict:
        st = self.state.read()
        return {"step": int(st.step), "lr": float(st.lr), "exp_avg": self.arena.exp_avg.clone(),
                "exp_avg_sq": self.arena.exp_avg_sq.clone()}


class StepLR:
    """torch.optim.lr_scheduler.StepLR(optim, step_size, gamma) for FusedAdamW (forensic_trainer.py:177)."""

    def __init__(self, optimizer: FusedAdamW, step_size: int, gamma: float = 0.1):
        self.optimizer, self.step_size, self.gamma = optimizer, int(step_size), float(gamma)
        self.base_lr = optimizer.param_groups[0]["initial_lr"]
        self.last_epoch = 0

    def step(self) -> None:
        self.last_epoch += 1
        self.optimizer.set_lr(self.base_lr * self.gamma ** (self.last_epoch // self.step_size))

    def get_last_lr(self):
        return [self.optimizer.param_groups[0]["lr"]]


class CosineAnnealingLR:
    """torch.optim.lr_scheduler.CosineAnnealingLR(optim, T_max, eta_min) for FusedAdamW (closed form), the
    integrated variant's schedule (forensic_trainer_integrated.py:151-155: T_max = epochs, eta_min = lr * min_lr_scale)."""

    def __init__(self, optim: FusedAdamW, T_max: int, eta_min: float = 0.0):
        import math
        self._math = math
        self.optim, self.T_max, self.eta_min = optim, max(1, int(T_max)), float(eta_min)
        self.base_lr = optim.param_groups[0]["initial_lr"]
        self.last_epoch = 0

    def get_last_lr(self):
        return [self.optim.param_groups[0]["lr"]]

    def step(self) -> None:
        self.last_epoch += 1
        lr = self.eta_min + (self.base_lr - self.eta_min) * (1 + self._math.cos(self._math.pi * self.last_epoch / self.T_max)) / 2
        self.optim.set_lr(lr)
