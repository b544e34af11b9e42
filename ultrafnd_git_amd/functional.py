"""autograd glue between torch and the C ABI: one Function per reference module, so
`loss.backward()` (forensic_trainer.py:291) drives the hand-written HIP backward.

Gradients are written by the kernels straight into the flat gradient arena; each parameter's
`.grad` is bound to its arena view (the Functions return None for parameters, so autograd's
accumulate step is skipped).  Consequence: a backward OVERWRITES the gradients, which equals
the reference's `zero_grad(set_to_none=True)` + `backward()` sequence; accumulating several
backward passes into one `.grad` is not supported on this path.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib as L


def _bind_grads(module) -> None:
    arena = module._arena
    for name, p in module.named_parameters():
        if arena.has_grad(module.akey(name)):
            p.grad = module.gview(name)


class FusionFunction(torch.autograd.Function):
    """CrossModalTransformer.forward / its backward (cross_modal_transformer.py:134-210)."""

    @staticmethod
    def forward(ctx, module, train, needs_grad, text, audio, visual, temporal, gnn, *params):
        dev = L.require_hip(text, audio, visual, temporal, gnn, module._arena.data)
        B, H = text.shape[0], module.hidden
        ws = module.workspace(B, needs_grad)
        state = module.rng()
        if train:
            state.advance()
        fused = torch.empty(B, H, dtype=torch.float32, device=dev)
        logits = torch.empty(B, 2, dtype=torch.float32, device=dev)
        forensic = torch.empty(3, B, dtype=torch.float32, device=dev)
        d = module.dims()
        L.check(L.lib().ufnd_fusion_forward(C.byref(d), C.byref(module.param_table()), text.data_ptr(),
                                            audio.data_ptr(), visual.data_ptr(), temporal.data_ptr(), gnn.data_ptr() if gnn.numel() else None,
                                            B, int(bool(train)), ws.data_ptr(), fused.data_ptr(), H, logits.data_ptr(),
                                            forensic.data_ptr(), state.ptr, L.stream_ptr(dev)), "ufnd_fusion_forward")
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(forensic)
        if needs_grad:
            ctx.module, ctx.train, ctx.B = module, bool(train), B
            ctx.save_for_backward(text, audio, visual, temporal, gnn)
            module._gen[B] = module._gen.get(B, 0) + 1
            ctx.gen = module._gen[B]
        return fused, logits, forensic

    @staticmethod
    def backward(ctx, d_fused, d_logits, _d_forensic):
        n_in = 8
        module = ctx.module
        nparams = sum(1 for p in module.parameters() if p.requires_grad)
        none = (None,) * (n_in + nparams)
        if d_fused is None and d_logits is None:
            return none
        if module._gen.get(ctx.B) != ctx.gen:
            raise RuntimeError("CrossModalTransformer: the saved activations of this forward were overwritten by a "
                               "later grad-enabled forward of the same batch size; call backward() first "
                               "(or run the other forward under torch.no_grad())")
        text, audio, visual, temporal, gnn = ctx.saved_tensors
        dev = text.device
        module._arena.ensure_grad()
        gt = module.grad_table()
        d = module.dims()
        H = module.hidden
        df = L.f32c(d_fused) if d_fused is not None else None
        dl = L.f32c(d_logits) if d_logits is not None else None
        L.check(L.lib().ufnd_fusion_backward(C.byref(d), C.byref(module.param_table()), C.byref(gt), text.data_ptr(),
                                             audio.data_ptr(), visual.data_ptr(), temporal.data_ptr(), gnn.data_ptr() if gnn.numel() else None,
                                             ctx.B, int(ctx.train), module.workspace(ctx.B, True).data_ptr(), L.ptr(df),
                                             H, L.ptr(dl), module.rng().ptr, L.stream_ptr(dev), None, 1), "ufnd_fusion_backward")
        _bind_grads(module)
        if dl is not None:
            module.classifier.weight.grad = module._cls_grad[:2 * H].view(2, H).clone()
            module.classifier.bias.grad = module._cls_grad[2 * H:2 * H + 2].clone()
        return none


class ClassifierFunction(torch.autograd.Function):
    """DeepTruthClassifier.forward / its backward (deep_truth_classifier.py:148-171)."""

    @staticmethod
    def forward(ctx, module, train, needs_grad, fused, aux, *params):
        dev = L.require_hip(fused, aux, module._arena.data)
        B, H = fused.shape[0], module.hidden
        if fused.dim() != 2 or fused.shape[1] != H:
            raise RuntimeError(f"fused: expected (B,{H}), got {tuple(fused.shape)}")
        if aux is not None and tuple(aux.shape) != (B, module.aux_dim):
            raise RuntimeError(f"aux: expected ({B},{module.aux_dim}), got {tuple(aux.shape)}")
        if fused.stride(1) != 1 or fused.stride(0) % 4 != 0 or fused.data_ptr() % 16 != 0:
            fused = fused.contiguous()
        ws = module.workspace(B, needs_grad)
        state = module.rng()
        if train:
            state.advance()
        logits = torch.empty(B, 2, dtype=torch.float32, device=dev)
        probs = torch.empty(B, 2, dtype=torch.float32, device=dev)
        d = module.dims()
        L.check(L.lib().ufnd_classifier_forward(C.byref(d), C.byref(module.param_table()), fused.data_ptr(),
                                                fused.stride(0), L.ptr(aux), B, int(bool(train)), ws.data_ptr(),
                                                logits.data_ptr(), probs.data_ptr(), state.ptr, L.stream_ptr(dev)),
                "ufnd_classifier_forward")
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(probs)
        if needs_grad:
            ctx.module, ctx.train, ctx.B = module, bool(train), B
            module._gen = getattr(module, "_gen", {})
            module._gen[B] = module._gen.get(B, 0) + 1
            ctx.gen = module._gen[B]
        return logits, probs

    @staticmethod
    def backward(ctx, d_logits, _d_probs):
        module = ctx.module
        nparams = sum(1 for p in module.parameters() if p.requires_grad)
        if d_logits is None:
            return (None,) * (5 + nparams)
        if module._gen.get(ctx.B) != ctx.gen:
            raise RuntimeError("DeepTruthClassifier: saved activations were overwritten by a later grad-enabled "
                               "forward of the same batch size; call backward() first")
        dev = d_logits.device
        module._arena.ensure_grad()
        d = module.dims()
        H = module.hidden
        d_fused = torch.empty(ctx.B, H, dtype=torch.float32, device=dev)
        dl = L.f32c(d_logits)
        L.check(L.lib().ufnd_classifier_backward(C.byref(d), C.byref(module.param_table()), C.byref(module.grad_table()),
                                                 ctx.B, int(ctx.train), module.workspace(ctx.B, True).data_ptr(),
                                                 dl.data_ptr(), d_fused.data_ptr(), H, module.rng().ptr,
                                                 L.stream_ptr(dev), None, 1), "ufnd_classifier_backward")
        _bind_grads(module)
        return (None, None, None, d_fused, None) + (None,) * nparams


class _CrossEntropy(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels):
        dev = L.require_hip(logits, labels)
        B = logits.shape[0]
        if logits.dim() != 2 or logits.shape[1] != 2:
            raise RuntimeError("cross_entropy: the HIP path implements the reference's 2-class head")
        logits = L.f32c(logits)
        labels = labels.to(torch.int64).contiguous()
        from .state import StepStateBuffer
        st = StepStateBuffer(dev)
        dlog = torch.empty_like(logits)
        L.check(L.lib().ufnd_softmax_ce(logits.data_ptr(), labels.data_ptr(), B, None, dlog.data_ptr(), st.ptr,
                                        L.stream_ptr(dev)), "ufnd_softmax_ce")
        ctx.save_for_backward(dlog)
        return st.float_view("loss").clone()

    @staticmethod
    def backward(ctx, g):
        (dlog,) = ctx.saved_tensors
        return dlog * g, None


def cross_entropy(logits: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
    """F.cross_entropy(logits, labels) (mean) on the HIP kernel (forensic_trainer.py:287)."""
    return _CrossEntropy.apply(logits, labels)
