"""ctypes binding of libultrafnd_hip.so (the C ABI declared in include/ultrafnd_hip.h).

There is NO CPU fallback: if the library is missing, or a call is made with tensors that do
not live on a HIP device, this raises.  torch is used only for device memory and streams.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path
from typing import Optional

import torch

_PKG = Path(__file__).resolve().parent
LIB_PATH = _PKG / "libultrafnd_hip.so"      # (tools that A/B two builds set this before the first lib() call)


class UltrafndHipError(RuntimeError):
    pass


# ------------------------------------------------------------------ structs (mirror the header)
class StepState(C.Structure):
    _fields_ = [("step", C.c_uint64), ("seed", C.c_uint64), ("lr", C.c_float), ("weight_decay", C.c_float),
                ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float), ("max_norm", C.c_float),
                ("grad_scale", C.c_float), ("loss", C.c_float), ("grad_norm", C.c_float), ("clip_coef", C.c_float),
                ("bc1", C.c_float), ("bc2_sqrt", C.c_float), ("reserved", C.c_float * 3)]


class Dims(C.Structure):
    _fields_ = [("hidden", C.c_int), ("text_dim", C.c_int), ("audio_dim", C.c_int), ("visual_dim", C.c_int),
                ("temporal_dim", C.c_int), ("gnn_dim", C.c_int), ("aux_dim", C.c_int), ("trees", C.c_int),
                ("depth", C.c_int), ("classes", C.c_int), ("fusion_dropout", C.c_float), ("clf_dropout", C.c_float),
                ("node_dropout", C.c_float)]


_FP = C.c_void_p


class FusionParams(C.Structure):
    _fields_ = [(n, _FP) for n in ("text_w", "text_b", "audio_w", "audio_b", "visual_w", "visual_b", "temporal_w",
                                   "temporal_b", "gnn_w", "gnn_b", "qkv_w", "qkv_b")] + \
               [("ev0_w", _FP * 3), ("ev0_b", _FP * 3), ("ev2_w", _FP * 3), ("ev2_b", _FP * 3)] + \
               [(n, _FP) for n in ("fuse0_w", "fuse0_b", "fuse3_w", "fuse3_b", "cls_w", "cls_b")]


class ClfParams(C.Structure):
    _fields_ = [(n, _FP) for n in ("pre0_w", "pre0_b", "pre3_w", "pre3_b", "gates", "thresh", "leaf", "tau",
                                   "bypass_w", "bypass_b", "temperature")]


class GcnParams(C.Structure):
    _fields_ = [(n, _FP) for n in ("w1", "b1", "w2", "b2")]


class GatherItem(C.Structure):
    """ufnd_gather_item: one cached tensor -> one static batch buffer."""
    _fields_ = [("src", _FP), ("dst", _FP), ("row_bytes", C.c_int), ("src_rows", C.c_int64)]


class TcnLayer(C.Structure):
    """ufnd_tcn_layer: one Conv1d + BatchNorm1d block of the sequence path."""
    _fields_ = [(n, _FP) for n in ("w", "b", "gamma", "beta", "running_mean", "running_var")]


class HeadIO(C.Structure):
    """ufnd_head_io: the buffers of the fused head step (ufnd_head_forward_loss / ufnd_head_backward)."""
    _fields_ = [(n, C.c_void_p) for n in ("text", "audio", "visual", "temporal", "gnn", "aux", "labels", "fusion_workspace", "clf_workspace",
                                           "logits", "probs", "forensic", "d_logits")]


class PartialsJob(C.Structure):
    """ufnd_partials_job: a LayerNorm's deferred dgamma / dbeta finish."""
    _fields_ = [("part", C.c_void_p), ("nblk", C.c_int), ("H", C.c_int), ("out0", C.c_void_p), ("out1", C.c_void_p)]


class RefreshItem(C.Structure):
    """ufnd_refresh_item: one Linear of a grouped bf16 operand refresh."""
    _fields_ = [("master", C.c_void_p), ("w", C.c_void_p), ("wt", C.c_void_p), ("rows", C.c_int), ("cols", C.c_int), ("ld_master", C.c_int),
                ("ld_w", C.c_int), ("ld_wt", C.c_int), ("tile0", C.c_int)]


PARTIALS_DEFER = 2
WGRAD_ALL, WGRAD_TRANSPOSE, WGRAD_PRODUCT = 0, 1, 2


class GemmLn(C.Structure):
    """ufnd_gemm_ln: the LayerNorm extras of ufnd_gemm_bf16_ln."""
    _fields_ = [("a_stats", _FP), ("colsum", _FP), ("r_stats", _FP), ("r_gamma", _FP), ("r_beta", _FP), ("out_stats", _FP),
                ("a_parts", C.c_int), ("r_parts", C.c_int), ("a_eps", C.c_float), ("r_eps", C.c_float), ("width", C.c_int),
                ("tile_cfg", C.c_int), ("residual_bf16", _FP), ("ldrb", C.c_int), ("guard", _FP)]

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        if "tile_cfg" not in kw:
            self.tile_cfg = -1


STEP_STATE_BYTES = C.sizeof(StepState)
BWD_ALL, BWD_FUSE_MLP, BWD_REST = 0, 1, 2
BWD_NO_LINEAR_GRADS = 16          # OR-ed into the phase / flags: the factor form of the gradient exchange (dp.FactorExchange)
ABI_VERSION = 5
FOLD_GUARD_SLOTS = 1024      # UFND_FOLD_GUARD_SLOTS

_lib: Optional[C.CDLL] = None


def _declare(lib: C.CDLL) -> None:
    P, I, S = C.c_void_p, C.c_int, C.c_size_t
    lib.ufnd_last_error.restype = C.c_char_p
    lib.ufnd_abi_version.restype = I
    lib.ufnd_fusion_workspace_floats.restype = S
    lib.ufnd_fusion_workspace_floats.argtypes = [C.POINTER(Dims), I]
    lib.ufnd_clf_workspace_floats.restype = S
    lib.ufnd_clf_workspace_floats.argtypes = [C.POINTER(Dims), I]
    lib.ufnd_clf_input_panel.restype = P
    lib.ufnd_clf_input_panel.argtypes = [C.POINTER(Dims), P, I, C.POINTER(I)]
    lib.ufnd_fusion_forward.argtypes = [C.POINTER(Dims), C.POINTER(FusionParams), P, P, P, P, P, I, I, P, P, I, P, P, P, P]
    lib.ufnd_fusion_backward.argtypes = [C.POINTER(Dims), C.POINTER(FusionParams), C.POINTER(FusionParams),
                                         P, P, P, P, P, I, I, P, P, I, P, P, P, P, I]
    lib.ufnd_fusion_backward_phase.argtypes = [C.POINTER(Dims), C.POINTER(FusionParams), C.POINTER(FusionParams),
                                               P, P, P, P, P, I, I, P, P, I, P, P, P, P, I, I]
    lib.ufnd_fusion_backward_phase.restype = I
    lib.ufnd_classifier_forward.argtypes = [C.POINTER(Dims), C.POINTER(ClfParams), P, I, P, I, I, P, P, P, P, P]
    lib.ufnd_classifier_backward.argtypes = [C.POINTER(Dims), C.POINTER(ClfParams), C.POINTER(ClfParams), I, I, P, P, P,
                                             I, P, P, P, I]
    lib.ufnd_classifier_backward_ex.argtypes = [C.POINTER(Dims), C.POINTER(ClfParams), C.POINTER(ClfParams), I, I, P, P, P,
                                                I, P, P, P, I, I]
    lib.ufnd_classifier_backward_ex.restype = I
    lib.ufnd_head_factor_floats.argtypes = [C.POINTER(Dims), I]
    lib.ufnd_head_factor_floats.restype = S
    lib.ufnd_head_pack_factors.argtypes = [C.POINTER(Dims), P, P, P, P, P, I, P, P, P, P]
    lib.ufnd_head_pack_factors.restype = I
    lib.ufnd_head_linear_grads_from_factors.argtypes = [C.POINTER(Dims), C.POINTER(FusionParams), C.POINTER(ClfParams), P, S, I, I, P]
    lib.ufnd_head_linear_grads_from_factors.restype = I
    lib.ufnd_head_forward_loss.argtypes = [C.POINTER(Dims), C.POINTER(FusionParams), C.POINTER(ClfParams), C.POINTER(HeadIO), I, I, P, P]
    lib.ufnd_head_forward_loss.restype = I
    lib.ufnd_head_backward.argtypes = [C.POINTER(Dims), C.POINTER(FusionParams), C.POINTER(FusionParams), C.POINTER(ClfParams), C.POINTER(ClfParams),
                                       C.POINTER(HeadIO), I, I, P, P, P, I, I]
    lib.ufnd_head_backward.restype = I
    lib.ufnd_softmax_ce.argtypes = [P, P, I, P, P, P, P]
    lib.ufnd_softmax_ce_weighted.argtypes = [P, P, I, C.c_float, C.c_float, C.c_float, P, P, P, P]
    lib.ufnd_softmax_ce_weighted.restype = I
    lib.ufnd_grad_norm.argtypes = [P, S, P, P, P]
    lib.ufnd_adamw_step.argtypes = [P, P, P, P, S, P, P]
    lib.ufnd_step_advance.argtypes = [P, P]
    lib.ufnd_clip_adamw_step.argtypes = [P, P, P, P, S, P, P, P]
    lib.ufnd_clip_adamw_step.restype = I
    for name in ("ufnd_fusion_forward", "ufnd_fusion_backward", "ufnd_classifier_forward", "ufnd_classifier_backward",
                 "ufnd_softmax_ce", "ufnd_grad_norm", "ufnd_adamw_step", "ufnd_step_advance"):
        getattr(lib, name).restype = I
    _declare_encoders(lib)


def _declare_encoders(lib: C.CDLL) -> None:
    """Tier-B entry points (include/ultrafnd_hip.h, second half)."""
    P, I, S, F = C.c_void_p, C.c_int, C.c_size_t, C.c_float
    sigs = {
        "ufnd_cast_bf16": [P, P, S, P],
        "ufnd_gemm_bf16": [P] * 6 + [I] * 9 + [P],
        "ufnd_gemm_bf16_ex": [P] * 6 + [I] * 10 + [P],
        "ufnd_layernorm": [P, I, P, P, P, P, I, I, F, P],
        "ufnd_attention_bf16": [P, P, P, I, I, I, P],
        "ufnd_qkv_attention_bf16": [P, P, P, P, P, I, I, I, I, I, C.POINTER(GemmLn), P],
        "ufnd_bert_embed": [P] * 8 + [I, I, I, I, F, P],
        "ufnd_masked_meanpool_l2": [P, P, P, I, I, I, P],
        "ufnd_bert_embed_packed": [P] * 9 + [I, I, I, I, F, P],
        "ufnd_attention_bf16_varlen": [P, P, P, I, I, I, P],
        "ufnd_meanpool_l2_packed": [P, P, P, P, I, I, P],
        "ufnd_vit_patchify": [P, P, I, I, I, P],
        "ufnd_vit_assemble": [P] * 8 + [I, I, I, F, P],
        "ufnd_gemm_bf16_ln": [P] * 6 + [I] * 9 + [C.POINTER(GemmLn), P],
        "ufnd_gemm_bf16_stat_parts": [I, I, I],
        "ufnd_ln_fold_guard": [P, I, I, I, F, P, P],
        "ufnd_ln_fold_guard_multi": [P, I, I, I, S, I, F, P, P],
        "ufnd_gemm_bf16_tile_count": [],
        "ufnd_gemm_bf16_tile_info": [I, C.POINTER(I), C.POINTER(I), C.POINTER(I)],
        "ufnd_stream_create_cu_mask": [C.POINTER(C.c_uint32), I, C.POINTER(P)],
        "ufnd_stream_destroy": [P],
        "ufnd_device_cu_count": [],
        "ufnd_l2norm_frames": [P, P, I, I, I, P],
        "ufnd_field_mean_l2": [P, P, P, I, I, I, P],
        "ufnd_temporal_align": [P] * 8 + [I] * 5 + [F, P, P],
    }
    sigs.update({
        "ufnd_gemm_bf16_dgrad": [P] * 6 + [I] * 10 + [P],
        "ufnd_gemm_bf16_wgrad": [P, P, P, I, I, I, I, I, I, P, I, P],
        "ufnd_transpose_bf16": [P, I, I, I, I, P, I, I, P, P, I, P],
        "ufnd_linear_wgrad": [P, I, P, I, I, I, I, P, P, P, P, I, P, P, C.POINTER(PartialsJob), I, P],
        "ufnd_refresh_operands": [P, I, I, P],
        "ufnd_layernorm_bwd_blocks": [I],
        "ufnd_row_partials_finish": [C.POINTER(PartialsJob), I, P],
        "ufnd_attention_bf16_lse": [P, P, P, P, I, I, I, P],
        "ufnd_attention_bf16_bwd": [P, P, P, P, P, P, P, I, I, I, P],
        "ufnd_layernorm_bwd": [P, I, P, P, I, P, I, P, P, I, P, P, P, I, I, I, F, P],
        "ufnd_masked_meanpool_l2_bwd": [P, P, P, P, I, I, I, P],
        "ufnd_l2norm_frames_bwd": [P, P, P, I, I, I, P],
        "ufnd_bert_embed_bwd": [P, P, P, P, P, I, I, I, I, I, I, P],
        "ufnd_vit_assemble_bwd": [P, P, P, P, I, I, I, P],
        "ufnd_act_bf16": [P, P, S, I, P],
    })
    for name, argtypes in sigs.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = I
    for name, argtypes in (("ufnd_gemm_bf16_wgrad_workspace_floats", [I, I, I]), ("ufnd_transpose_colsum_workspace_floats", [I, I]),
                           ("ufnd_attention_bwd_workspace_floats", [I, I, I]), ("ufnd_layernorm_bwd_workspace_floats", [I, I])):
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = S
    D = C.c_double
    lib.ufnd_ocr_adjacency.argtypes = [P, P, I, D, P, I, P]
    lib.ufnd_ocr_adjacency.restype = I
    lib.ufnd_gcn_workspace_floats.argtypes = [I, I, I, I, I]
    lib.ufnd_gcn_workspace_floats.restype = S
    lib.ufnd_gcn_forward.argtypes = [P, P, I, C.POINTER(GcnParams), P, P, I, I, I, I, F, P, P]
    lib.ufnd_gcn_forward.restype = I
    lib.ufnd_gcn_pretrain_step.argtypes = [P, P, I, C.POINTER(GcnParams), P, P, P, P, P, P, I, I, I, I, F, F, F, I, P, P, P]
    lib.ufnd_gcn_pretrain_step.restype = I
    lib.ufnd_ocr_adjacency_weighted.argtypes = [P, P, I, D, P, I, P]
    lib.ufnd_ocr_adjacency_weighted.restype = I
    lib.ufnd_node_features.argtypes = [P, I, P, I, P, I, P, I, I, I, I, I, I, P, P]
    lib.ufnd_node_features.restype = I
    lib.ufnd_gnn_workspace_floats.argtypes = [I, I, I, I]
    lib.ufnd_gnn_workspace_floats.restype = S
    lib.ufnd_gnn_forward.argtypes = [P, P, I, C.POINTER(GcnParams), P, P, I, I, I, I, F, P, P]
    lib.ufnd_gnn_forward.restype = I
    lib.ufnd_gnn_backward.argtypes = [P, C.POINTER(GcnParams), P, P, P, P, P, P, I, I, I, I, F, P, P]
    lib.ufnd_gnn_backward.restype = I
    lib.ufnd_fusion_gnn_input_grad.argtypes = [C.POINTER(Dims), C.POINTER(FusionParams), P, I, P, P, P]
    lib.ufnd_fusion_gnn_input_grad.restype = I
    lib.ufnd_fusion_feature_grads.argtypes = [C.POINTER(Dims), C.POINTER(FusionParams), P, I, P, P, P, P]
    lib.ufnd_fusion_feature_grads.restype = I
    lib.ufnd_gather_rows.argtypes = [P, I, C.POINTER(GatherItem), I, P]
    lib.ufnd_gather_rows.restype = I
    lib.ufnd_tcn_weight_ld.argtypes = [I, I]
    lib.ufnd_tcn_weight_ld.restype = I
    lib.ufnd_tcn_workspace_floats.argtypes = [I, I, I, I, I]
    lib.ufnd_tcn_workspace_floats.restype = S
    lib.ufnd_tcn_forward.argtypes = [P, I, P, I, I, I, C.POINTER(TcnLayer), I, I, I, P, P, I, I, F, F, F, P, P, P, P]
    lib.ufnd_tcn_forward.restype = I
    lib.ufnd_temporal_weight_ld.argtypes = [I]
    lib.ufnd_temporal_weight_ld.restype = I
    lib.ufnd_temporal_workspace_floats.argtypes = [I, I, I]
    lib.ufnd_temporal_workspace_floats.restype = S


def lib() -> C.CDLL:
    """The loaded library; raises (never falls back) when it is not built."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise UltrafndHipError(
                f"{LIB_PATH} is missing: run `python -m ultrafnd_git_amd.build` (hipcc, gfx950). "
                "This package has no CPU fallback.")
        l = C.CDLL(str(LIB_PATH))
        _declare(l)
        if l.ufnd_abi_version() != ABI_VERSION:
            raise UltrafndHipError("libultrafnd_hip.so ABI version mismatch; rebuild")
        _lib = l
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise UltrafndHipError(f"{what} failed (code {rc}): {lib().ufnd_last_error().decode()}")


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def require_hip(*tensors: Optional[torch.Tensor]) -> torch.device:
    dev = None
    for t in tensors:
        if t is None:
            continue
        if t.device.type != "cuda":
            raise UltrafndHipError(
                f"tensor on {t.device}: the fusion hot path runs on a HIP device only (no CPU fallback); "
                "move the module and its inputs to 'cuda'")
        dev = dev or t.device
        if t.device != dev:
            raise UltrafndHipError("all tensors of one call must live on the same device")
    return dev


def stream_ptr(device: torch.device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def f32c(t: torch.Tensor) -> torch.Tensor:
    """fp32 + contiguous (no copy when already so)."""
    if t.dtype != torch.float32:
        t = t.float()
    return t if t.is_contiguous() else t.contiguous()
