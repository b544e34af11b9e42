"""GPU: the data-parallel leg of the step on the one MI355X this box has.

Each test runs a FRESH process tree under `python -m torch.distributed.run` (started by tests/launcher.py, a process
that never touched the GPU): tests/dp_child.py does the work and prints one JSON line."""
import json
import socket
import sys

import pytest

pytestmark = pytest.mark.gpu


def _port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(launch_job, nproc, mode, out_dir, env):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(_port()), "tests/dp_child.py", "--mode", mode, "--out", str(out_dir)]
    r = launch_job(cmd, env={"HSA_ENABLE_IPC_MODE_LEGACY": "0", **env}, timeout=900)
    assert r["rc"] == 0, (r["rc"], r["out"][-3000:], r["err"][-6000:])
    line = [ln for ln in r["out"].splitlines() if ln.startswith("{")][-1]
    return json.loads(line)


def test_forced_rccl_exchange_on_one_rank_changes_no_bit(launch_job, tmp_path):
    """torchrun world 1, RCCL, ForensicTrainer(force_exchange=True): init_process_group (high-priority group stream), the two-bucket
    asynchronous all-reduce started inside backward (two captured head graphs with the collectives between them),
    grad_scale in the norm / AdamW, graph capture with the process group alive -- three plain steps (graph and eager)
    and three pipelined steps leave the arena bit-identical to the same steps without any exchange."""
    res = _run(launch_job, 1, "force1", tmp_path, {})
    assert res["backend"] == "nccl" and res["steps"] == 3
    assert res["plain_steps_bit_identical"] and res["pipelined_bit_identical"] and res["lookahead_bit_identical"], res
    # trainable encoders: four buckets (head x 2, text encoder, visual encoder) over the joint arena, two steps
    assert res["train_encoders_bit_identical"] and res["train_encoders_buckets"] == 4 and res["train_encoders_arena"] > 20_000_000, res


def test_two_ranks_on_one_gpu_equal_the_full_batch_step(launch_job, tmp_path):
    """Two ranks sharing the GPU (gloo; device tensors through host memory by tests/host_staged.py -- not a product path): sharded batch + summed bucketed
    gradients + 1/world folded into grad_scale == the single-process full-batch trainer, three steps; then fit() /
    test() over sharded loaders: gathered metrics agree on both ranks, evaluation shards partition the split
    (no wrapped duplicates), rank 0 checkpoints and every rank tests with rank 0's parameters."""
    res = _run(launch_job, 2, "world2", tmp_path, {})
    assert res["ranks_agree"] and res["fit_ranks_agree"] and res["ckpt_exists"], res
    assert res["param_max_abs_err"] <= 2e-5 * max(1.0, res["param_scale"]), res
    assert abs(res["grad_norm"] - res["grad_norm_ref"]) <= 1e-4 * max(1.0, res["grad_norm_ref"]), res
    assert res["val_rows_total"] == res["val_rows_split"], res
    assert 0.5 < res["best_val_auc"] <= 1.0


def test_exchange_variants_run_on_the_device(launch_job, tmp_path):
    """torchrun world 1, RCCL, forced exchange: bf16 gradient payload (25.5 MB instead of 51) and reduce-scatter + all-gather,
    inside the captured two-bucket step.  A one-rank sum: the fp32 forms change no bit; the bf16 form is one rounding of the
    gradient (the arena's parameters stay within bf16 rounding of the plain run after two steps).  World-2 numerics of the
    same variants: tests/test_dp_gloo.py."""
    res = _run(launch_job, 1, "variants", tmp_path, {})
    assert res["backend"] == "nccl"
    assert res["fp32_forms_bit_identical"] and res["bf16_forms_agree"] and res["bf16_grad_is_bf16_valued"], res
    assert res["ar_bf16_wire_MB"] * 2 == pytest.approx(res["ar_fp32_wire_MB"], rel=1e-3)
    assert 0.0 < res["bf16_param_max_rel"] < 1e-3, res
    # factor exchange at world 1 (RCCL all-gather of one pack, the Linear gradients formed from it): no bit changes; 2.5 MB + the small ranges
    assert res["factors_bit_identical"], res
    assert res["factors_wire_MB"] * 15 < res["ar_fp32_wire_MB"], res


def test_factor_exchange_two_ranks_on_one_gpu(launch_job, tmp_path):
    """TrainConfig.grad_exchange="factors" with two ranks sharing the GPU (gloo, host-staged -- not a product path): every rank all-gathers
    its dY / X panels and forms the summed Linear gradients over both ranks' rows in one launch (ufnd_head_linear_grads_from_factors); the
    other 21 k gradient floats are all-reduced.  Against the all-reduce form of the same three steps: equal within fp32 summation order
    (one chain over 2 x 8 rows instead of two chains of 8 added); against the single-process full-batch step: as the all-reduce form is;
    both ranks end bit-identical; the captured and the eager step agree bit for bit."""
    res = _run(launch_job, 2, "factors2", tmp_path, {})
    assert res["ranks_agree"] and res["graph_equals_eager"], res
    assert res["grad_rel_vs_all_reduce"] <= 2e-6 and res["param_rel_vs_all_reduce"] <= 2e-6, res
    assert res["grad_rel_vs_full_batch"] <= 2e-5, res
    assert res["param_max_abs_err"] <= 2e-5 * max(1.0, res["param_scale"]), res
    assert abs(res["grad_norm"] - res["grad_norm_ref"]) <= 1e-4 * max(1.0, res["grad_norm_ref"]), res
    small = sum(hi - lo for lo, hi in res["small_ranges"])
    assert 15_000 < small < 40_000 and len(res["small_ranges"]) <= 4, res          # gates / thresholds / leaves / bypass + evidence_proj
    assert res["wire_bytes"]["factors"] * 15 < res["wire_bytes"]["all_reduce"], res
