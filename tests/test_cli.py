"""run_train_eval.py: the reference's flags (run_train_eval.py:28-47) parse with the same names and defaults (CPU), and the
script trains / evaluates end to end on a small synthetic cache and prints the reference's result keys (:102-109) -- what
scripts/smoke_test_v2.py::test_trainer_initialization asserts (result keys present) -- on the GPU."""
import re
import subprocess
import sys
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parents[1]


def test_cli_flags_mirror_the_reference(monkeypatch):
    sys.path.insert(0, str(REPO))
    import run_train_eval as R
    monkeypatch.setattr(sys, "argv", ["run_train_eval.py"])
    a = R.parse_args()
    # names and defaults of the reference's parser
    assert (a.out_dir, a.epochs, a.batch_size, a.lr, a.weight_decay, a.gnn_dim, a.gnn_overlap_thresh, a.seed) == \
           ("outputs_v2", 12, 16, 2e-4, 1e-4, 128, 0.12, 42)
    assert a.cpu is False and a.no_gnn is False and a.eval_only is False and a.ocr_phrase_pkl == ""
    monkeypatch.setattr(sys, "argv", ["run_train_eval.py", "--cpu"])
    with pytest.raises(SystemExit):
        R.main()                       # no CPU path: refused loudly, nothing silently falls back


@pytest.mark.gpu
def test_cli_trains_and_reports_the_reference_result_keys(tmp_path):
    out = tmp_path / "out"
    cmd = [sys.executable, str(REPO / "run_train_eval.py"), "--synthetic", "96", "--epochs", "2", "--batch_size", "16", "--out_dir", str(out)]
    r = subprocess.run(cmd, cwd=str(REPO), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    for key in ("Test Loss:", "Test Acc :", "Test AUC :", "Test Precision:", "Test Recall:", "Test F1:", "Test Cmcs:", "Test Dfdr:"):
        assert key in r.stdout, (key, r.stdout[-1500:])
    assert (out / "best.pt").exists()
    loss = float(re.search(r"Test Loss: ([0-9.]+)", r.stdout).group(1))
    assert 0.0 < loss < 5.0
    # --eval_only re-loads best.pt and reports the same test loss
    r2 = subprocess.run(cmd + ["--eval_only"], cwd=str(REPO), capture_output=True, text=True, timeout=600)
    assert r2.returncode == 0, r2.stderr[-2000:]
    assert abs(float(re.search(r"Test Loss: ([0-9.]+)", r2.stdout).group(1)) - loss) <= 1e-4
