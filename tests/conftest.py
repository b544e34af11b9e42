import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parents[1]
if str(REPO) not in sys.path:
    sys.path.insert(0, str(REPO))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

GOLDEN = REPO / "tests" / "golden"

_launcher = None


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # GPU runs: start the job launcher NOW, before any test module has had a chance to initialise the GPU in this
    # process (tests/launcher.py: process trees for the data-parallel tests are its children, not ours)
    global _launcher
    expr = config.getoption("-m") or ""
    if "gpu" in expr and "not gpu" not in expr and _launcher is None:
        _launcher = subprocess.Popen([sys.executable, str(REPO / "tests" / "launcher.py")], stdin=subprocess.PIPE,
                                     stdout=subprocess.PIPE, text=True, cwd=str(REPO))


def pytest_unconfigure(config):
    global _launcher
    if _launcher is not None:
        try:
            _launcher.stdin.write(json.dumps({"quit": True}) + "\n")
            _launcher.stdin.flush()
            _launcher.wait(timeout=10)
        except Exception:
            _launcher.kill()
        _launcher = None


@pytest.fixture(scope="session")
def launch_job():
    """launch_job(cmd, env=None, timeout=600) -> {"rc", "out", "err"}: runs `cmd` as a child of the launcher."""
    if _launcher is None:
        pytest.skip("job launcher not started (run with -m gpu)")

    def run(cmd, env=None, timeout=600):
        _launcher.stdin.write(json.dumps({"cmd": list(cmd), "env": env or {}, "timeout": timeout, "cwd": str(REPO)}) + "\n")
        _launcher.stdin.flush()
        line = _launcher.stdout.readline()
        if not line:
            raise RuntimeError("job launcher died")
        return json.loads(line)
    return run


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
