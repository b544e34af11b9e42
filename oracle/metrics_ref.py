"""TEST INFRASTRUCTURE (see oracle/__init__.py) -- epoch metrics oracle.

numpy-only restatement of src/training/metrics/forensic_metrics.py:62-171 (the
reference calls sklearn; the formulas below are the definitions sklearn
implements for the binary case and are pinned against the reference's outputs
by tests/golden/make_golden.py -> tests/golden/metrics_kat.json).
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np


def _to_prob_1(y_score) -> np.ndarray:
    """forensic_metrics.py:35-56."""
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


def roc_auc(y_true: np.ndarray, y_prob: np.ndarray) -> float:
    """Mann-Whitney U with average ranks for ties == sklearn.roc_auc_score (binary).
    0.5 when a class is absent (forensic_metrics.py:19-32)."""
    y_true = np.asarray(y_true)
    y_prob = np.asarray(y_prob, dtype=float)
    if y_true.size == 0 or np.unique(y_true).size < 2:
        return 0.5
    order = np.argsort(y_prob, kind="mergesort")
    s = y_prob[order]
    ranks = np.empty(len(s), dtype=float)
    i = 0
    while i < len(s):
        j = i
        while j + 1 < len(s) and s[j + 1] == s[i]:
            j += 1
        ranks[order[i:j + 1]] = 0.5 * (i + j) + 1.0
        i = j + 1
    pos = y_true == 1
    n_pos, n_neg = int(pos.sum()), int((~pos).sum())
    return float((ranks[pos].sum() - n_pos * (n_pos + 1) / 2.0) / (n_pos * n_neg))


def compute_classification_metrics(y_true, y_score, threshold: float = 0.5,
                                   include_cm: bool = False) -> Dict[str, float]:
    """forensic_metrics.py:62-99."""
    y_true = np.asarray(y_true).astype(int)
    y_prob = _to_prob_1(y_score).astype(float)
    y_pred = (y_prob >= threshold).astype(int)
    tp = float(((y_pred == 1) & (y_true == 1)).sum())
    tn = float(((y_pred == 0) & (y_true == 0)).sum())
    fp = float(((y_pred == 1) & (y_true == 0)).sum())
    fn = float(((y_pred == 0) & (y_true == 1)).sum())
    n = float(y_true.size)
    prec = tp / (tp + fp) if (tp + fp) > 0 else 0.0
    rec = tp / (tp + fn) if (tp + fn) > 0 else 0.0
    f1 = 2 * prec * rec / (prec + rec) if (prec + rec) > 0 else 0.0
    out = {"accuracy": (tp + tn) / n if n else 0.0, "auc": roc_auc(y_true, y_prob),
           "precision": prec if n else 0.0, "recall": rec if n else 0.0, "f1": f1 if n else 0.0}
    if include_cm and n:
        out.update({"cm_tn": tn, "cm_fp": fp, "cm_fn": fn, "cm_tp": tp})
    return out


def compute_cmcs(semantic_conflict, temporal_delay) -> float:
    """forensic_metrics.py:105-119."""
    mix = np.clip(0.5 * (np.asarray(semantic_conflict, float) + np.asarray(temporal_delay, float)), 0.0, 1.0)
    return float(1.0 - mix.mean()) if mix.size else 0.0


def compute_dfdr(y_true, y_score, threshold: float = 0.5) -> float:
    """forensic_metrics.py:122-141."""
    y_true = np.asarray(y_true).astype(int)
    y_pred = (_to_prob_1(y_score).astype(float) >= threshold).astype(int)
    pos = y_true == 1
    return float((y_pred[pos] == 1).sum()) / float(pos.sum()) if pos.sum() >= 1 else 0.0


def aggregate_epoch_metrics(y_true, y_score, forensic: Optional[Dict[str, np.ndarray]] = None,
                            threshold: float = 0.5, include_cm: bool = False) -> Dict[str, float]:
    """forensic_metrics.py:144-171."""
    cls = compute_classification_metrics(y_true, y_score, threshold, include_cm)
    if forensic:
        sc, td = forensic.get("semantic_conflict"), forensic.get("temporal_delay")
        if sc is not None and td is not None:
            cls["cmcs"] = compute_cmcs(sc, td)
        ei = forensic.get("emotion_intensity")
        if ei is not None:
            ei = np.asarray(ei, float)
            cls["emotion_intensity_mean"] = float(ei.mean()) if ei.size else 0.0
        cls["dfdr"] = compute_dfdr(y_true, y_score, threshold)
    return cls
