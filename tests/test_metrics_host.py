"""CPU: the product's epoch metrics reproduce the reference's known answers."""
import json

import numpy as np

from tests.helpers import GOLDEN
from ultrafnd_git_amd import metrics as M


def test_metric_kats_product():
    kat = json.loads((GOLDEN / "metrics_kat.json").read_text())
    for c in kat["aggregate"]:
        f = c["forensic"] and {k: np.array(v) for k, v in c["forensic"].items()}
        got = M.aggregate_epoch_metrics(np.array(c["y"]), np.array(c["p"]), forensic=f, include_cm=c["include_cm"])
        assert set(got) == set(c["expected"]), c["name"]
        for k, v in c["expected"].items():
            assert abs(got[k] - v) <= 1e-12, (c["name"], k)
    for c in kat["two_column"]:
        got = M.compute_classification_metrics(np.array(c["y"]), np.array(c["score"]))
        for k, v in c["expected"].items():
            assert abs(got[k] - v) <= 1e-12, (c["name"], k)
    assert M.aggregate_epoch_metrics(np.array([]), np.array([]))["auc"] == 0.5
