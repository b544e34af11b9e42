"""CPU: metrics oracle vs the reference's known answers (tests/golden/metrics_kat.json)."""
import json

import numpy as np

from oracle import metrics_ref as OM
from tests.helpers import GOLDEN


def test_metric_kats():
    kat = json.loads((GOLDEN / "metrics_kat.json").read_text())
    for c in kat["aggregate"]:
        f = c["forensic"] and {k: np.array(v) for k, v in c["forensic"].items()}
        got = OM.aggregate_epoch_metrics(np.array(c["y"]), np.array(c["p"]), forensic=f, include_cm=c["include_cm"])
        assert set(got) == set(c["expected"]), c["name"]
        for k, v in c["expected"].items():
            assert abs(got[k] - v) <= 1e-12, (c["name"], k)
    for c in kat["two_column"]:
        got = OM.compute_classification_metrics(np.array(c["y"]), np.array(c["score"]))
        for k, v in c["expected"].items():
            assert abs(got[k] - v) <= 1e-12, (c["name"], k)


def test_survey_known_answers():
    """SURVEY.md 8c literal values."""
    m = OM.aggregate_epoch_metrics(np.array([0, 1, 1, 0, 1, 0]), np.array([.2, .7, .4, .6, .9, .1]),
                                   forensic={"semantic_conflict": np.array([.1, .5, .9, .3, .2, .4]),
                                             "temporal_delay": np.array([.2, .4, .6, .8, 1.0, 0.0]),
                                             "emotion_intensity": np.array([.1, .2, .3, .4, .5, .6])}, include_cm=True)
    assert abs(m["accuracy"] - 2 / 3) < 1e-12 and abs(m["auc"] - 8 / 9) < 1e-12
    assert (m["cm_tn"], m["cm_fp"], m["cm_fn"], m["cm_tp"]) == (2, 1, 1, 2)
    assert abs(m["cmcs"] - 0.55) < 1e-12 and abs(m["emotion_intensity_mean"] - 0.35) < 1e-12
    assert OM.aggregate_epoch_metrics(np.array([]), np.array([]))["auc"] == 0.5
