"""bench.py --gpus N started plainly launches its own N ranks (a child torchrun, before anything touches the GPU) or
fails loudly -- it never reports a number for fewer ranks than it was asked for.  The
step being scaled is the reference's train-loop body, src/training/forensic_trainer.py:285-298."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parents[1]


def _bench(*args, env=None):
    e = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, str(REPO / "bench.py"), *args], cwd=str(REPO), env=e, capture_output=True, text=True, timeout=300)


def test_dry_launch_prints_the_torchrun_command_and_spawns_nothing():
    r = _bench("--gpus", "2", "--steps", "7", "--warmup", "2", "--dry-launch")
    assert r.returncode == 0, r.stderr[-2000:]
    cmd = json.loads(r.stdout.strip().splitlines()[-1])["dry_launch"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=2" in cmd
    assert "--standalone" in cmd and cmd[cmd.index("--local-addr") + 1] == "127.0.0.1"      # torchrun binds its own port
    i = cmd.index(str(REPO / "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "2", "--steps", "7", "--warmup", "2"]          # same arguments, minus --dry-launch


def test_launcher_process_never_imports_torch():
    """The launcher branch must be GPU-free by construction: it runs (dry) with torch made unimportable."""
    r = _bench("--gpus", "2", "--dry-launch", env={"PYTHONPATH": str(REPO / "tests" / "no_torch")})
    assert r.returncode == 0, r.stderr[-2000:]
    assert "dry_launch" in r.stdout
    r = _bench("--gpus", "1", "--self-launch", "--dry-launch", env={"PYTHONPATH": str(REPO / "tests" / "no_torch")})
    assert r.returncode == 0 and "--self-launch" in json.loads(r.stdout.strip().splitlines()[-1])["dry_launch"]


def test_sysfs_gpu_counter_on_a_fake_topology(tmp_path):
    """count_gpus_sysfs: KFD topology nodes with simd_count > 0 are GPUs (CPU nodes report 0)."""
    sys.path.insert(0, str(REPO))
    import importlib
    bench = importlib.import_module("bench")
    for i, simd in enumerate([0, 0, 1024, 1024, 1024]):
        d = tmp_path / str(i)
        d.mkdir()
        (d / "properties").write_text(f"cpu_cores_count {64 if simd == 0 else 0}\nsimd_count {simd}\nmem_banks_count 1\n")
    (tmp_path / "5").mkdir()                     # a node without a properties file is skipped
    assert bench.count_gpus_sysfs(str(tmp_path)) == 3
    assert bench.count_gpus_sysfs(str(tmp_path / "missing")) >= 0


def test_world_size_that_disagrees_with_gpus_is_an_error():
    r = _bench("--gpus", "4", env={"RANK": "0", "WORLD_SIZE": "2", "LOCAL_RANK": "0"})
    assert r.returncode == 3 and "WORLD_SIZE=2" in r.stderr
    r = _bench("--gpus", "1", env={"RANK": "0", "WORLD_SIZE": "2", "LOCAL_RANK": "0"})
    assert r.returncode == 3


def test_too_few_devices_is_a_loud_failure_on_cpu():
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("this host has two devices")
    r = _bench("--gpus", "2", "--steps", "3")
    assert r.returncode == 3 and "refusing to measure fewer ranks" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]         # no JSON line, no number


@pytest.mark.gpu
def test_gpus_2_on_the_one_gpu_box_exits_non_zero(launch_job):
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("this box has two devices")
    r = launch_job([sys.executable, "bench.py", "--gpus", "2", "--steps", "4", "--warmup", "1"], timeout=300)
    assert r["rc"] == 3, r
    assert "refusing to measure fewer ranks" in r["err"] and '"value"' not in r["out"]


@pytest.mark.gpu
def test_self_launch_spawn_path_runs_rccl_at_world_1(launch_job):
    """The branch the driver's N > 1 run takes -- a GPU-free launcher that starts torchrun, ranks that initialise RCCL -- proven on
    the one-GPU box at N = 1: one JSON line, RCCL saw one rank, and the gradient exchange was measured."""
    r = launch_job([sys.executable, "bench.py", "--gpus", "1", "--self-launch", "--steps", "4", "--warmup", "1", "--no-cpu-baseline",
                    "--no-lookahead-compare", "--repeats", "1"], timeout=600)
    assert r["rc"] == 0, r
    lines = [ln for ln in r["out"].splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r["out"][-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 1 and j["exchange"]["ranks_seen"] == 1 and j["exchange"]["backend"] == "nccl"
    assert j["exchange"]["allreduce_ms"] is not None and j["exchange"]["allreduce_ms"] > 0
