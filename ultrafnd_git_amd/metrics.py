"""Epoch metrics of the training loop (src/training/metrics/forensic_metrics.py:62-181): threshold
accuracy / precision / recall / F1, ROC-AUC (0.5 when a class is absent), CMCS, DFDR.  Host-side
float64 numpy, once per epoch; `fit()` early-stops on `auc`, so values must equal the reference's
(which calls sklearn) -- pinned by tests/golden/metrics_kat.json."""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
from scipy.stats import rankdata


def _to_prob_1(y_score) -> np.ndarray:
    y = np.asarray(y_score)
    if y.ndim == 1:
        return y
    if y.ndim == 2 and y.shape[1] == 2:
        if np.allclose(y.sum(axis=1), 1.0, atol=1e-3):
            return y[:, 1]
        z = y - y.max(axis=1, keepdims=True)
        ez = np.exp(z)
        return (ez / np.clip(ez.sum(axis=1, keepdims=True), 1e-12, None))[:, 1]
    return np.max(y, axis=1)


def _safe_auc(y_true: np.ndarray, y_prob: np.ndarray) -> float:
    y_true, y_prob = np.asarray(y_true), np.asarray(y_prob, dtype=float)
    if y_true.size == 0 or np.unique(y_true).size < 2 or not np.all(np.isfinite(y_prob)):
        return 0.5
    pos = y_true == 1
    n_pos, n_neg = int(pos.sum()), int((~pos).sum())
    r = rankdata(y_prob, method="average")
    return float((r[pos].sum() - n_pos * (n_pos + 1) / 2.0) / (n_pos * n_neg))


def compute_classification_metrics(y_true, y_score, threshold: float = 0.5, include_cm: bool = False) -> Dict[str, float]:
    y_true = np.asarray(y_true).astype(int)
    y_prob = _to_prob_1(y_score).astype(float)
    y_pred = (y_prob >= threshold).astype(int)
    tp = float(np.sum((y_pred == 1) & (y_true == 1)))
    tn = float(np.sum((y_pred == 0) & (y_true == 0)))
    fp = float(np.sum((y_pred == 1) & (y_true == 0)))
    fn = float(np.sum((y_pred == 0) & (y_true == 1)))
    n = float(y_true.size)
    prec = tp / (tp + fp) if tp + fp > 0 else 0.0
    rec = tp / (tp + fn) if tp + fn > 0 else 0.0
    out = {"accuracy": (tp + tn) / n if n else 0.0, "auc": _safe_auc(y_true, y_prob), "precision": prec,
           "recall": rec, "f1": 2 * prec * rec / (prec + rec) if prec + rec > 0 else 0.0}
    if include_cm and n:
        out.update({"cm_tn": tn, "cm_fp": fp, "cm_fn": fn, "cm_tp": tp})
    return out


def compute_cmcs(semantic_conflict, temporal_delay) -> float:
    mix = np.clip(0.5 * (np.asarray(semantic_conflict, float) + np.asarray(temporal_delay, float)), 0.0, 1.0)
    return float(1.0 - mix.mean()) if mix.size else 0.0


def compute_dfdr(y_true, y_score, threshold: float = 0.5) -> float:
    y_true = np.asarray(y_true).astype(int)
    y_pred = (_to_prob_1(y_score).astype(float) >= threshold).astype(int)
    pos = y_true == 1
    denom = float(pos.sum())
    return float(np.sum(y_pred[pos] == 1)) / denom if denom >= 1.0 else 0.0


def aggregate_epoch_metrics(y_true, y_score, forensic: Optional[Dict[str, np.ndarray]] = None, threshold: float = 0.5,
                            include_cm: bool = False) -> Dict[str, float]:
    cls = compute_classification_metrics(y_true, y_score, threshold=threshold, include_cm=include_cm)
    if forensic:
        sc, td = forensic.get("semantic_conflict"), forensic.get("temporal_delay")
        if sc is not None and td is not None:
            cls["cmcs"] = compute_cmcs(sc, td)
        ei = forensic.get("emotion_intensity")
        if ei is not None:
            ei = np.asarray(ei, float)
            cls["emotion_intensity_mean"] = float(ei.mean()) if ei.size else 0.0
        cls["dfdr"] = compute_dfdr(y_true, y_score, threshold=threshold)
    return cls


def pretty_print(split: str, m: Dict[str, float]) -> None:
    ordered = ["accuracy", "auc", "precision", "recall", "f1", "cmcs", "dfdr"]
    extras = [k for k in m if k not in ordered and not k.startswith("cm_")]
    line = " | ".join(f"{k}:{m[k]:.4f}" for k in ordered if k in m)
    if extras:
        line += " | " + " ".join(f"{k}:{m[k]:.4f}" for k in extras)
    print(f"[{split}] {line}")
