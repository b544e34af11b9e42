#!/usr/bin/env python3
"""bench.py -- train-step samples/s of the text+vision fusion hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Started plainly with --gpus N > 1 (no RANK in the environment) the script launches its own N ranks: before anything touches
the GPU it checks that the node has N devices (exit code 3 with a message otherwise -- it never measures fewer ranks than
asked for) and starts `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...` with the
same arguments as a CHILD process, passes its output through and exits with its code.  --dry-launch prints that command
and starts nothing.  A WORLD_SIZE that disagrees with --gpus is an error (exit code 3), not a warning.

One "step" = one pass of the hot path over one synthetic FakeSV batch (BASELINE.json configs[1]:
full model, batch 32 per GPU, seq_len 128, one 224x224 frame), inputs resident in HBM:
  BERT-base text encoder fwd (bf16 MFMA) + ViT-B/32 visual encoder fwd (bf16 MFMA) -> pooled
  features (+ temporal = TemporalSyncNet.align(text, visual)) -> CrossModalTransformer +
  DeepTruthClassifier fwd (train-mode dropout) -> CE ->
  backward -> [RCCL all-reduce] -> clip_grad_norm_(5) -> AdamW.   (encoders frozen, as in the
  reference; nothing is cached or skipped inside the timed region.)
Encoder lookahead (--lookahead G, default 4): the frozen encoders run over G consecutive batches per pass -- every batch is
encoded exactly once, inside the timed region, with features bit-identical to one-batch passes -- while the head, the loss,
the exchange, the clip and AdamW step batch by batch (32 samples per optimizer step).  The line's `lookahead_1` field is the
same run's throughput with one batch per encoder pass.
Prints ONE JSON line (rank 0) with the contract's fields plus `roofline` (the bf16 GEMM kernel,
timed with HIP events on its stream) and `cpu_baseline` (the oracle's same step on host cores,
bounded sample).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
REPO = Path(__file__).resolve().parent
sys.path.insert(0, str(REPO))

torch = None      # imported by the RANKS only (main(), behind the launcher branch): the launcher process never loads torch, so
dist = None       # nothing in it can initialise the GPU -- by construction, not by what a particular ROCm build's device query does


def _import_torch():
    global torch, dist
    import torch as _torch
    import torch.distributed as _dist
    torch, dist = _torch, _dist

SEQ_LEN, IMAGE, FRAMES = 128, 224, 1
MFMA_BF16_PEAK_TFLOPS = 2500.0     # dense bf16, /opt/skills/guides/MI355X_MICROARCH.md


def gemm_flops_per_step(B: int) -> tuple[float, int]:
    """Algorithmic FLOPs and launch count of ufnd_gemm_bf16 in one step."""
    H, I3 = 768, 3072
    per_tok = 2.0 * (H * 3 * H + H * H + 2 * H * I3)           # QKV + out + FFN1 + FFN2
    text = 12 * (B * SEQ_LEN) * per_tok
    n = B * FRAMES
    vis = 2.0 * (n * 49) * 3072 * H + 12 * (n * 50) * per_tok + 2.0 * n * H * 512
    return text + vis, 12 * 4 + 1 + 12 * 4 + 1


def gemm_bytes_per_step(B: int, residual_dtype: str = "bf16") -> float:
    """Algorithmic HBM bytes of the same launches: bf16 operands in, bf16 rows out; the two residual GEMMs of a layer
    also read the residual stream and write 8 B of row statistics per 32 output columns.  bf16 stream (default): the
    residual is the bf16 row the previous GEMM wrote (2 B per element) and no fp32 copy exists; fp32 stream: 4 B read +
    4 B written per element on top.  The text encoder's fused projection + attention launch (L = 128) reads X and the
    stacked weight and writes ctx only."""
    H, I3 = 768, 3072
    res = (2.0 if residual_dtype == "bf16" else 8.0) + 0.25

    def g(M, N, K, residual):
        return 2.0 * (M * K + N * K + M * N) + (res * M * N if residual else 0.0)

    def layer(M, fused_attention):
        qkv = 2.0 * (M * H + 3 * H * H + M * H) if fused_attention else g(M, 3 * H, H, False)
        return qkv + g(M, H, H, True) + g(M, I3, H, False) + g(M, H, I3, True)
    n = B * FRAMES
    return 12 * layer(B * SEQ_LEN, SEQ_LEN == 128) + 12 * layer(n * 50, False) + (2.0 * (n * 49 * 3072 + H * 3072) + 4.0 * n * 49 * H) + 2.0 * (n * H + 512 * H) + 4.0 * n * 512


def pmc_traffic():
    """HBM bytes per gemm_bf16_kernel launch from the committed rocprofv3 PMC passes of this same command (newest
    profiles/r*_pmc_traffic.md: reads = 2 x FETCH_SIZE, writes = WRITE_SIZE, launch-weighted over the GEMM rows).
    PMC counters cannot be collected from inside the process; None when no summary is present."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r*_pmc_traffic.md")))
    if not files:
        return None, None
    tot = n = 0.0
    for line in open(files[-1]):
        m = re.match(r"\| gemm_(?:bf16|pp)_kernel<[^|]*\| (\d+) \| (\d+) \| ([0-9.]+) \| ([0-9.]+) \|", line)
        if m:
            tot += int(m.group(2)) * (float(m.group(3)) + float(m.group(4))) * 1e6
            n += int(m.group(2))
    return (tot / n, "profiles/" + os.path.basename(files[-1])) if n else (None, None)


def make_batches(B: int, n: int, seed: int, dev: torch.device):
    """Synthetic FakeSV-shaped raw batches, already in HBM."""
    g = torch.Generator().manual_seed(seed)
    out = []
    for _ in range(n):
        ids = torch.randint(0, 30522, (B, SEQ_LEN), generator=g)
        ids[:, 0] = 101
        lens = torch.randint(16, SEQ_LEN + 1, (B,), generator=g)
        a = torch.randn(B, 128, generator=g)
        out.append({"input_ids": ids.to(dev), "attention_mask": (torch.arange(SEQ_LEN)[None] < lens[:, None]).to(torch.int32).to(dev),
                    "frames": torch.randn(B, FRAMES, 3, IMAGE, IMAGE, generator=g).to(dev),
                    "audio_features": (a / a.norm(dim=1, keepdim=True)).to(dev),
                    "temporal_features": torch.randn(B, 256, generator=g).to(dev),
                    "gnn_feat": torch.randn(B, 128, generator=g).to(dev), "aux": torch.rand(B, 2, generator=g).to(dev),
                    "label": torch.randint(0, 2, (B,), generator=g).to(dev)})
    return out


def cpu_baseline(B: int, slice_b: int = 8, warm: int = 3, timed: int = 20, budget_s: float = 60.0):
    """The oracle's restatement of the same step on the host cores (kind 'port'): `warm` warm-up + `timed` timed steps
    (median) on a `slice_b`-sample slice of the workload (a full B=32 step takes 3.5-5.5 s on these hosts; the slice
    keeps the whole leg near 25 s).  The encoders are >99.9 % of the step's FLOPs and linear in the sample count."""
    import statistics
    from oracle import encoders_ref as E
    from oracle import tier_a as O
    b = min(B, slice_b)
    wt = E.seeded_weights(E.bert_shapes(), 1)
    wv = E.seeded_weights(E.vit_shapes(), 2)
    fus, clf = O.seeded_params(3)
    opt = O.AdamWState()
    ids, mask = E.synthetic_tokens(4, b, SEQ_LEN)
    frames = E.synthetic_frames(5, b, FRAMES)
    batch = O.seeded_batch(6, b)

    def step():
        with torch.no_grad():
            batch["text_features"] = E.text_features(wt, ids, mask)
            batch["visual_features"] = E.visual_features(wv, frames)
        O.train_step(fus, clf, batch, opt, grad_clip=5.0, train=True, dropout=0.1)
    for _ in range(warm):
        step()
    times, t_all = [], time.perf_counter()
    while len(times) < timed and (len(times) < 5 or time.perf_counter() - t_all < budget_s):
        t0 = time.perf_counter()
        step()
        times.append(time.perf_counter() - t0)
    med = statistics.median(times)
    short = "" if len(times) >= timed else f" (time budget {budget_s:.0f} s reached: {len(times)} of {timed} timed steps)"
    return {"value": round(b / med, 3), "unit": "samples/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"median of {len(times)} timed steps after {warm} warm-up on a {b}-sample slice of the same workload "
                      f"(L={SEQ_LEN}, {FRAMES}x{IMAGE}^2 frame; per-sample work identical to B={B}){short}: {med * 1e3:.0f} ms/step "
                      f"(min {min(times) * 1e3:.0f}, max {max(times) * 1e3:.0f}); host has {os.cpu_count()} cpus"}

def exchange_probe(tr, dev, n: int = 20):
    """What RCCL saw: the process group's world size and the time of one gradient exchange of the step -- the two-bucket
    all-reduce of the flat fp32 gradient arena, exactly as the step issues it -- alone on the GPU (median of `n`, HIP events
    on the step's stream).  After the timed region; the gradient buffer is zeroed first (a sum of zeros costs the same and
    cannot overflow).  None at one rank without a process group."""
    if not dist.is_initialized():
        return {"ranks_seen": 1, "allreduce_ms": None, "bytes": int(tr.arena.n_grad) * 4, "backend": None}
    import statistics
    g = tr.reducer.grad
    g.zero_()
    was = tr.reducer.force
    tr.reducer.force = True                    # (world 1 under torchrun: run the collectives anyway)
    factors = getattr(tr.reducer, "factors", False)
    if factors:                                # the factor form: all-gather of one pack + the small all-reduces + the grouped dW launch over all ranks' rows
        B = tr.cfg.batch_size
        hb = tr.head.bufs(B, True, 0)
        hb["fws"].zero_(); hb["cws"].zero_()
        pack = tr.head.factor_pack(hb, B)
        form = lambda p, s_, r: tr.head.linear_grads_from_factors(hb, B, p, s_, r)
    times = []
    for _ in range(n + 3):
        torch.cuda.synchronize(dev)
        dist.barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        if factors:
            tr.reducer.start_factors(pack, form)
        tr.reducer.start()
        tr.reducer.finish()
        e1.record()
        torch.cuda.synchronize(dev)
        times.append(e0.elapsed_time(e1))
    tr.reducer.force = was
    t = torch.tensor([statistics.median(times[3:])], device=dev, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    what = (f"median of {n} factor exchanges (all-gather of the rank's {tr.reducer.wire_bytes()} B of factor panels and small gradient ranges, then the "
            "summed Linear gradients formed over all ranks' rows in one launch), alone on the GPU, max over ranks" if factors else
            f"median of {n} {len(tr.reducer.buckets)}-bucket sum-all-reduces of the fp32 gradient arena, alone on the GPU, max over ranks")
    return {"ranks_seen": dist.get_world_size(), "allreduce_ms": round(float(t.item()), 4), "bytes": int(tr.reducer.wire_bytes()),
            "buckets": len(tr.reducer.buckets), "backend": dist.get_backend(), "algorithm": "factors" if factors else tr.reducer.algorithm,
            "what": what}


def bench_train_encoders(args, dev, world, rank):
    """The same step with BOTH encoders trained (TrainConfig.train_encoders): encoder forwards that keep their activations,
    head forward / CE / backward, feature gradients, encoder backwards (data and weight gradients of every Linear, attention,
    LayerNorms, embeddings), one global-norm clip + AdamW over head + encoders (197 M parameters), operand re-cast.  Not
    the headline (the reference never trains its encoders); roofline = 3 x the forward's GEMM FLOPs / step time."""
    from ultrafnd_git_amd.encoders import BertTextEncoder, ClipVisualEncoder
    from ultrafnd_git_amd.temporal import TemporalSyncNet
    from ultrafnd_git_amd.trainer import ForensicTrainer, TrainConfig, synthetic_cache
    B = args.batch
    torch.manual_seed(42)
    tenc, venc = BertTextEncoder().to(dev), ClipVisualEncoder().to(dev)
    cfg = TrainConfig(data_root="", ocr_phrase_pkl=None, out_dir="/tmp/ufnd_bench", batch_size=B, device=str(dev), use_graph=not args.no_graph,
                      encode_inline=True, seed=42, train_encoders=True)
    tsync = TemporalSyncNet(in_dim=768, out_dim=256).to(dev)
    tr = ForensicTrainer(cfg, cache=synthetic_cache(64, seed=1), text_encoder=tenc, visual_encoder=venc, temporal_net=tsync)
    tr.text_bp.overlap_wgrad = tr.vis_bp.overlap_wgrad = not args.no_wgrad_overlap
    tr.fusion.train(); tr.clf.train()
    batches = make_batches(B, 4, 42 + 2 + 1000 * rank, dev)

    def fence():
        torch.cuda.synchronize(dev)
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize(dev)
    for i in range(args.warmup):
        tr.train_step(batches[i % 4])
    blocks, host = [], []
    for _ in range(max(1, args.repeats)):
        fence()
        t0 = time.perf_counter()
        for i in range(args.steps):
            tr.train_step(batches[i % 4])
        enq = time.perf_counter() - t0          # the host has enqueued every launch of the block (about 1,200 per step)
        fence()
        dt = time.perf_counter() - t0
        if dist.is_initialized():
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        blocks.append(dt)
        host.append(enq)
    dt = sorted(blocks)[len(blocks) // 2]
    flops, _ = gemm_flops_per_step(B)
    step_ms = dt / args.steps * 1e3
    ach = 3.0 * flops / (step_ms * 1e-3) / 1e12
    xch = exchange_probe(tr, dev) if dist.is_initialized() else {"ranks_seen": 1}
    if rank == 0:
        print(json.dumps({"metric": "train-step samples/sec with BOTH encoders trained (not the headline: the reference keeps them frozen)",
                          "value": round(world * B * args.steps / dt, 2), "unit": "samples/s", "n_gpus": world, "ranks_seen": xch["ranks_seen"],
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(step_ms, 4), "higher_is_better": True, "scaling": "weak",
                          "dtype": "bf16", "data": "synthetic", "final_loss": float(tr.optim.state.read().loss),
                          "config": {"workload": f"full Ultrafnd step with trainable encoders (BERT-base L={SEQ_LEN} + {FRAMES} x ViT-B/32 fwd AND bwd, "
                                                 "fusion + classifier fwd / bwd, one clip + AdamW over head + encoders)", "per_gpu_batch": B,
                                     "trainable_parameters": int(tr.arena.n_grad), "parallelism": f"dp{world}"},
                          "timing": {"ms_per_step_blocks": [round(x / args.steps * 1e3, 4) for x in blocks],
                                     "host_enqueue_ms_per_step": round(sorted(host)[len(host) // 2] / args.steps * 1e3, 4)}, "exchange": xch,
                          "roofline": {"bound": "mfma", "kernel": "gemm_bf16_kernel (forward, data-gradient and weight-gradient forms)",
                                       "achieved": round(ach, 2), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / MFMA_BF16_PEAK_TFLOPS, 4),
                                       "basis": "3 x the forward's algorithmic GEMM FLOPs (forward + dgrad + wgrad of every encoder Linear) / step time",
                                       "flops_per_step": 3.0 * flops, "traffic": None}}))
    if dist.is_initialized():
        dist.destroy_process_group()


def count_gpus_sysfs(root: str = "/sys/class/kfd/kfd/topology/nodes") -> int:
    """GPUs of this node from the KFD topology: nodes whose `properties` file has simd_count > 0 (CPU nodes have 0).  Plain file
    reads: no HIP / torch call, so the launcher stays GPU-free whatever the ROCm build's device query would do.  Falls back to
    counting /dev/dri/renderD* when the topology is absent (0 when neither exists)."""
    n = 0
    base = Path(root)
    if base.is_dir():
        for node in base.iterdir():
            try:
                props = dict(line.split()[:2] for line in (node / "properties").read_text().splitlines() if len(line.split()) >= 2)
            except OSError:
                continue
            if int(props.get("simd_count", "0")) > 0:
                n += 1
        return n
    dri = Path("/dev/dri")
    return len(list(dri.glob("renderD*"))) if dri.is_dir() else 0


def launch_command(n: int, argv: list) -> list:
    """The torchrun command line `python bench.py --gpus n ...` turns itself into (one rank per GPU, RCCL over xGMI).  A
    standalone rendezvous on 127.0.0.1: torchrun binds its own free port (no pick-then-bind race)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--standalone", "--local-addr", "127.0.0.1",
            str(Path(__file__).resolve())] + [a for a in argv if a != "--dry-launch"]


def self_launch(args) -> None:
    """--gpus N > 1 (or --self-launch at N = 1) without a RANK in the environment: become the launcher.  THIS process never
    imports torch and makes no HIP call (the device count comes from sysfs), so the ranks are ordinary children of a GPU-free
    parent; their stdout (rank 0's JSON line) and stderr pass through, and this process exits with torchrun's return code."""
    import subprocess
    assert "torch" not in sys.modules, "the launcher must stay torch-free"
    cmd = launch_command(args.gpus, sys.argv[1:])
    if args.dry_launch:
        print(json.dumps({"dry_launch": cmd}))
        raise SystemExit(0)
    have = count_gpus_sysfs()
    if have < args.gpus:
        print(f"[bench] --gpus {args.gpus} but this node has {have} HIP device(s) (KFD topology): refusing to measure fewer ranks than asked for",
              file=sys.stderr)
        raise SystemExit(3)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    raise SystemExit(subprocess.call(cmd, env=env, cwd=str(REPO)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32, help="per-GPU batch (BASELINE configs[1]: 32)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--repeats", type=int, default=5, help="timed blocks of --steps steps, back to back; the reported figures are the median block")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--seq-len", type=int, default=128, help="text length (BASELINE configs[3]: 512 with --batch 128)")
    ap.add_argument("--frames", type=int, default=1, help="frames per sample (BASELINE configs[4]: 8 with --batch 8 per GPU)")
    ap.add_argument("--no-fold-ln", action="store_true", help="one LayerNorm kernel per LayerNorm instead of folding them into the GEMM epilogues")
    ap.add_argument("--cu-split", type=str, default=None, help="compute units of the text,visual encoder streams, e.g. 192,64 (0 = ordinary streams)")
    ap.add_argument("--text-tiles", type=str, default=None, help="experiments: GEMM tile ids of the text encoder, e.g. qkv=22,out=16,ffn1=22,ffn2=16")
    ap.add_argument("--vis-tiles", type=str, default=None, help="experiments: GEMM tile ids of the visual encoder")
    ap.add_argument("--lookahead", type=int, default=4,
                    help="encoder lookahead G: the frozen encoders run over G consecutive batches per pass (1 = one batch per pass); "
                         "the head, the exchange and the optimizer always step batch by batch")
    ap.add_argument("--no-lookahead-compare", action="store_true", help="skip the lookahead-1 comparison leg")
    ap.add_argument("--no-fuse-attn", action="store_true", help="text encoder: Q/K/V projection and attention as two launches per layer")
    ap.add_argument("--residual-dtype", choices=("fp32", "bf16"), default="bf16",
                    help="encoders' residual stream: fp32 beside the bf16 operands, or the bf16 rounding itself (encoders.py)")
    ap.add_argument("--head-only", action="store_true",
                    help="secondary measurement: the reference's actual training mode (cached features, no encoders in the step)")
    ap.add_argument("--force-exchange", action="store_true",
                    help="under torchrun at world 1: run the bucketed gradient exchange anyway (the RCCL path's fixed cost on one GPU)")
    ap.add_argument("--grad-exchange", choices=("all_reduce", "rs_ag", "factors"), default="all_reduce",
                    help="data-parallel exchange of the head's gradients (dp.py): bucketed all-reduce (default), reduce-scatter + all-gather, or "
                         "all-gathered factor panels with the Linear gradients formed locally over all ranks' rows")
    ap.add_argument("--no-wgrad-overlap", action="store_true", help="--train-encoders: weight-gradient products on the backward's own stream instead of a second one")
    ap.add_argument("--train-encoders", action="store_true",
                    help="secondary measurement: fine-tune both encoders with the head (forward with saved activations + hand-written "
                         "backward; the reference keeps its encoders frozen)")
    ap.add_argument("--dry-launch", action="store_true", help="--gpus N > 1 started plainly: print the torchrun command it would start, start nothing")
    ap.add_argument("--self-launch", action="store_true",
                    help="take the launcher branch at any N (N = 1 on a one-GPU box: the torchrun spawn path, RCCL at world 1 with the "
                         "exchange forced, so that `exchange` reports a measured all-reduce)")
    args = ap.parse_args()

    if (args.gpus > 1 or args.self_launch) and "RANK" not in os.environ:
        self_launch(args)                       # (never returns; torch is not imported yet)
    _import_torch()
    if args.self_launch:      # (a rank started by the launcher branch: the flag travels with the arguments)
        args.force_exchange = True
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if rank == 0:
            print(f"[bench] --gpus {args.gpus} but WORLD_SIZE={world}: refusing to report a number for a different rank count",
                  file=sys.stderr)
        raise SystemExit(3)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (MI355X); there is no CPU path to time as the product")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    launched = "RANK" in os.environ and "MASTER_PORT" in os.environ      # started by torch.distributed.run
    if world > 1 or launched:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        from ultrafnd_git_amd.dp import init_process_group
        init_process_group(dev)

    from ultrafnd_git_amd.encoders import BertTextEncoder, ClipVisualEncoder
    from ultrafnd_git_amd.temporal import TemporalSyncNet
    from ultrafnd_git_amd.trainer import ForensicTrainer, TrainConfig, synthetic_cache

    B = args.batch
    torch.manual_seed(42)
    if args.head_only:
        cfg = TrainConfig(data_root="", ocr_phrase_pkl=None, out_dir="/tmp/ufnd_bench", batch_size=B, device=str(dev),
                          use_graph=not args.no_graph, seed=42, grad_exchange=args.grad_exchange)
        tr = ForensicTrainer(cfg, cache=synthetic_cache(max(64, 8 * B), seed=1), force_exchange=args.force_exchange)
        tr.fusion.train(); tr.clf.train()
        from ultrafnd_git_amd.trainer import IndexedBatch
        ds = tr.train_loader.dataset            # batches as the loader yields them: row indices of the HBM-resident split
        bl = [IndexedBatch(ds, (torch.arange(B, device=dev) + k * B) % len(ds)) for k in range(2)]
        def sync():
            torch.cuda.synchronize(dev)
            if dist.is_initialized():
                dist.barrier()
            torch.cuda.synchronize(dev)
        for i in range(args.warmup):
            tr.train_step(bl[i % 2])
        blocks = []
        for _ in range(max(1, args.repeats)):
            sync()
            t0 = time.perf_counter()
            for i in range(args.steps):
                tr.train_step(bl[i % 2])
            sync()
            blocks.append(time.perf_counter() - t0)
        dt = sorted(blocks)[len(blocks) // 2]
        P = tr.arena.n_grad
        gbps = 11 * 4 * P / (dt / args.steps) / 1e9
        xch = exchange_probe(tr, dev)
        if rank == 0:
            print(json.dumps({"metric": "head-only train-step samples/sec (cached features: the reference's own training mode)",
                              "value": round(world * B * args.steps / dt, 1), "unit": "samples/s", "ms_per_step": round(dt / args.steps * 1e3, 4),
                              "per_gpu_batch": B, "n_gpus": world, "ranks_seen": xch["ranks_seen"], "exchange": xch,
                              "steps": args.steps, "warmup": args.warmup,
                              "gradient_exchange": ("none (one rank)" if not tr.reducer.active else
                                                    "factor panels all-gathered, Linear gradients formed locally" if args.grad_exchange == "factors" else
                                                    "bucketed all-reduce overlapped with backward"),
                              "timing": {"what": f"median of {len(blocks)} blocks of {args.steps} steps",
                                         "ms_per_step_blocks": [round(x / args.steps * 1e3, 4) for x in blocks]},
                              "roofline": {"bound": "hbm", "kernel": "whole head step (21 launches; AdamW + grad-norm + the three fuse_mlp.0 GEMMs move 90 % of the bytes)",
                                           "achieved": round(gbps, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(gbps / 8000.0, 4),
                                           "algorithmic_MB_per_step": round(11 * 4 * P / 1e6, 1), "traffic": None}}))
        if dist.is_initialized():
            dist.destroy_process_group()
        return
    global SEQ_LEN, FRAMES
    SEQ_LEN, FRAMES = args.seq_len, args.frames       # (defaults = the headline configuration, BASELINE configs[1])
    if args.train_encoders:
        return bench_train_encoders(args, dev, world, rank)
    tenc = BertTextEncoder(fold_ln=not args.no_fold_ln, residual_dtype=args.residual_dtype).to(dev)    # BERT-base geometry, random init (no checkpoints offline)
    venc = ClipVisualEncoder(fold_ln=not args.no_fold_ln, residual_dtype=args.residual_dtype).to(dev)  # CLIP ViT-B/32 geometry, random init
    split = None
    if args.cu_split and args.cu_split != "0":
        split = tuple(int(x) for x in args.cu_split.split(","))
    cfg = TrainConfig(data_root="", ocr_phrase_pkl=None, out_dir="/tmp/ufnd_bench", batch_size=B, device=str(dev),
                      use_graph=not args.no_graph, encode_inline=True, seed=42, cu_split=split, grad_exchange=args.grad_exchange,
                      persistent_inputs=True)       # (the four input buffer sets below are rotated, never reallocated)
    tenc.fuse_qkv_attention = not args.no_fuse_attn
    for enc, spec in ((tenc, args.text_tiles), (venc, args.vis_tiles)):
        if spec:
            enc.tiles = {k: int(v) for k, v in (kv.split("=") for kv in spec.split(","))}
    tsync = TemporalSyncNet(in_dim=768, out_dim=256).to(dev)   # temporal = align(text, visual), as the cache builder does
    tr = ForensicTrainer(cfg, cache=synthetic_cache(64, seed=1), text_encoder=tenc, visual_encoder=venc, temporal_net=tsync,
                         force_exchange=args.force_exchange)
    tr.fusion.train()
    tr.clf.train()
    G = max(1, args.lookahead)
    batches = make_batches(G * B, 4, 42 + 2 + 1000 * rank, dev)       # four persistent lookahead groups of G batches each (G = 1: four batches)

    def run(n):
        """n optimizer steps, software-pipelined: the head / exchange / optimizer of the current batches overlap the encoders of
        the next ones.  G > 1: encoders once per group of G batches (a last, partial group still encodes all G: never less work)."""
        if G == 1:
            tr.prefetch_features(batches[0])
            for i in range(n):
                tr.train_step_pipelined(batches[i % 4], batches[(i + 1) % 4] if i + 1 < n else None)
            return
        ng = (n + G - 1) // G
        tr.prefetch_features(batches[0], group=True)
        for gi in range(ng):
            tr.train_group_pipelined(batches[gi % 4], batches[(gi + 1) % 4] if gi + 1 < ng else None, steps=min(G, n - gi * G))

    def fence():
        torch.cuda.synchronize(dev)
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize(dev)

    # graph-capture priming (setup, not a step): the loader rotates four persistent input buffers through the trainer's two
    # input slots, and the encoders keep one captured graph per (input buffers, slot) so that nothing is restaged -- capture all
    # eight before the warm-up, so that no capture can fall into a timed block
    for slot in (0, 1):
        for bt in batches:
            tr.prefetch_features(bt, slot, group=G > 1)
    torch.cuda.synchronize(dev)
    run(args.warmup)
    blocks = []
    for _ in range(max(1, args.repeats)):      # every block: EXACTLY --steps steps between two barrier + synchronize fences
        fence()
        t0 = time.perf_counter()
        run(args.steps)
        fence()
        dt = time.perf_counter() - t0
        if dist.is_initialized():
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        blocks.append(dt)
    dt = sorted(blocks)[len(blocks) // 2]      # the median block is the one reported
    final_loss = float(tr.optim.state.read().loss)
    # the same steps with ONE batch per encoder pass (lookahead 1), in the same run, for comparison: not the headline
    la1 = None
    if G > 1 and not args.no_lookahead_compare:
        singles = [{k: v[:B] for k, v in bt.items()} for bt in batches]         # the first batch of every group (leading rows: contiguous views)

        def run1(n):
            tr.prefetch_features(singles[0])
            for i in range(n):
                tr.train_step_pipelined(singles[i % 4], singles[(i + 1) % 4] if i + 1 < n else None)
        for slot in (0, 1):
            for bt in singles:
                tr.prefetch_features(bt, slot)
        torch.cuda.synchronize(dev)
        run1(args.warmup)
        b1 = []
        for _ in range(3):
            fence()
            t0 = time.perf_counter()
            run1(args.steps)
            fence()
            d1 = time.perf_counter() - t0
            if dist.is_initialized():
                t = torch.tensor([d1], device=dev, dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                d1 = float(t.item())
            b1.append(d1)
        d1 = sorted(b1)[1]
        la1 = {"value": round(world * B * args.steps / d1, 2), "ms_per_step": round(d1 / args.steps * 1e3, 4),
               "what": "the same optimizer steps with one batch per encoder pass (lookahead 1), median of 3 blocks, same process"}

    # ---- roofline of the dominant kernel (bf16 GEMM): HIP events around every launch, instrumented pass
    roof = None
    if rank == 0:
        gem_ms, launches = tr.measure_gemm_time(batches[0], steps=3)          # one encoder pass = G batches
        flops, n_launch = gemm_flops_per_step(B)
        assert launches == n_launch, (launches, n_launch)
        gem_ms /= G                                                            # GEMM launch time per optimizer step
        step_ms = dt / args.steps * 1e3
        achieved = flops / (step_ms * 1e-3) / 1e12             # whole step: launches of the two encoder streams overlap in time
        per_launch = flops / (gem_ms * 1e-3) / 1e12
        headline = (SEQ_LEN, FRAMES, B, G) == (128, 1, 32, 4) and not args.no_fold_ln
        traffic, traffic_src = pmc_traffic() if headline else (None, None)       # (the PMC passes were taken on the headline run)
        raw_us = tr.last_raw_interval_us
        roof = {"bound": "mfma", "kernel": "gemm_bf16_kernel + gemm_pp_kernel (the persistent form of the same GEMM, FFN1 launches)", "achieved": round(achieved, 2), "peak": MFMA_BF16_PEAK_TFLOPS,
                "unit": "TFLOP/s", "frac": round(achieved / MFMA_BF16_PEAK_TFLOPS, 4),
                "basis": "FLOPs of the step's %d bf16 GEMM launches / wall time of the step (ms_per_step): the text and the visual "
                         "launches overlap in time on two streams, so this -- not the sum of launch durations -- is what the driver's "
                         "clock can check; nothing else in the step is credited" % n_launch,
                "traffic": round(traffic) if traffic else None, "traffic_unit": "bytes/launch", "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": round(gemm_bytes_per_step(G * B, args.residual_dtype) / n_launch),
                "launches_per_encoder_pass": n_launch, "optimizer_steps_per_encoder_pass": G, "flops_per_step": flops,
                "flops_per_launch_avg": flops * G / n_launch,
                "per_launch": {"how": "HIP events on the launch stream around every launch, sequential instrumented pass after the timed "
                                      "region (text encoder, then visual encoder); avg_launch_us = raw event interval - marker price, "
                                      "marker price = lower quartile of the event-to-event gaps with no kernel in between",
                               "avg_event_interval_us": round(raw_us, 2), "event_marker_us": round(tr.last_marker_us, 2),
                               "avg_launch_us": round(gem_ms * G * 1e3 / n_launch, 2), "gemm_ms_per_step": round(gem_ms, 4),
                               "achieved": round(per_launch, 2), "frac": round(per_launch / MFMA_BF16_PEAK_TFLOPS, 4),
                               "by_shape_MxNxK": {k: {"launches_per_encoder_pass": v[0] // 3, "avg_us": round(v[1] / v[0] * 1e3, 2),
                                                      "tflops": round(2.0 * eval(k.replace("x", "*")) / (v[1] / v[0] * 1e-3) / 1e12, 1)}
                                                  for k, v in tr.last_gemm_by_shape.items()}}}
    xch = exchange_probe(tr, dev)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:      # (the CPU leg is timed at N = 1 only)
        cpu = cpu_baseline(B)
    if rank == 0:
        value = world * B * args.steps / dt
        print(json.dumps({
            "metric": "train-step samples/sec (FakeSV batch, seq128+224^2)", "value": round(value, 2), "unit": "samples/s",
            "n_gpus": world, "ranks_seen": xch["ranks_seen"], "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": ("BASELINE configs[1]" if (SEQ_LEN, FRAMES, B) == (128, 1, 32) else "variant of BASELINE configs[1]") +
                                   f": full Ultrafnd step (BERT-base L={SEQ_LEN} fwd + {FRAMES} x ViT-B/32 224^2 fwd, frozen; "
                                   "fusion+classifier fwd/bwd, clip, AdamW)" +
                                   (f"; the frozen encoders run over {G} consecutive batches per pass (lookahead: every batch is encoded exactly "
                                    "once, inside the timed region; features are bit-identical to one-batch passes), one optimizer step per batch" if G > 1 else ""),
                       "per_gpu_batch": B, "global_batch": world * B, "encoder_lookahead_batches": G,
                       "seq_len": SEQ_LEN, "frames": FRAMES, "image": IMAGE, "parallelism": f"dp{world}",
                       "encoder_dtype": "bf16 operands / fp32 accumulate", "residual_stream": args.residual_dtype, "head_dtype": "fp32", "hip_graph": not args.no_graph,
                       "weights": "random init of the named architectures"},
            "timing": {"what": f"median of {len(blocks)} back-to-back blocks of {args.steps} steps, each between barrier + synchronize fences",
                       "ms_per_step_blocks": [round(x / args.steps * 1e3, 4) for x in blocks],
                       "ms_per_step_min": round(min(blocks) / args.steps * 1e3, 4), "ms_per_step_max": round(max(blocks) / args.steps * 1e3, 4)},
            "lookahead_1": la1, "final_loss": final_loss, "exchange": xch, "roofline": roof, "cpu_baseline": cpu}))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
