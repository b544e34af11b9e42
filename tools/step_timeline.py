#!/usr/bin/env python3
"""Where a pipelined train step spends its time, WITHOUT a profiler (GPU box): timing events on the
three streams (head/optimizer, text encoder graph, visual encoder graph) of bench.py's loop, printed
relative to the start of each step (median over steps), plus host enqueue time per step.
usage: step_timeline.py [--steps 20] [--batch 32] [--no-fold-ln] [--serial]"""
import argparse
import statistics
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch

import bench
from ultrafnd_git_amd.encoders import BertTextEncoder, ClipVisualEncoder
from ultrafnd_git_amd.temporal import TemporalSyncNet
from ultrafnd_git_amd.trainer import ForensicTrainer, TrainConfig, synthetic_cache


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--no-fold-ln", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    import os
    if "RANK" in os.environ:          # under torchrun: the data-parallel path (force_exchange=True runs the collective at world 1)
        import torch.distributed as dist
        torch.cuda.set_device(dev)
        dist.init_process_group("nccl", device_id=dev)
    B = args.batch
    tenc = BertTextEncoder(fold_ln=not args.no_fold_ln).to(dev)
    venc = ClipVisualEncoder(fold_ln=not args.no_fold_ln).to(dev)
    cfg = TrainConfig(data_root="", ocr_phrase_pkl=None, out_dir="/tmp/ufnd_tl", batch_size=B, device=str(dev), use_graph=True,
                      encode_inline=True, seed=42)
    tsync = TemporalSyncNet(in_dim=768, out_dim=256).to(dev)
    tr = ForensicTrainer(cfg, cache=synthetic_cache(64, seed=1), text_encoder=tenc, visual_encoder=venc, temporal_net=tsync,
                         force_exchange="RANK" in os.environ)
    tr.fusion.train(); tr.clf.train()
    batches = bench.make_batches(B, 4, 45, dev)

    def run(n):
        tr.prefetch_features(batches[0])
        host = []
        for i in range(n):
            t0 = time.perf_counter()
            tr.train_step_pipelined(batches[i % 4], batches[(i + 1) % 4] if i + 1 < n else None)
            host.append(time.perf_counter() - t0)
        return host

    run(5)
    torch.cuda.synchronize()
    tr.pipe.timeline = []
    t0 = time.perf_counter()
    host = run(args.steps)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / args.steps
    tl = tr.pipe.timeline
    tr.pipe.timeline = None
    # split into steps at every "step0"
    steps, cur = [], None
    for tag, ev in tl:
        if tag == "step0":
            cur = {}
            steps.append(cur)
        if cur is not None:
            cur[tag] = ev
    rows = {}
    for i in range(2, len(steps) - 2):
        s0 = steps[i]["step0"]
        nxt = steps[i + 1]["step0"]
        for tag, ev in steps[i].items():
            rows.setdefault(tag, []).append(s0.elapsed_time(ev) * 1e3)
        rows.setdefault("next step0", []).append(s0.elapsed_time(nxt) * 1e3)
    print(f"wall {wall * 1e3:.3f} ms/step; host enqueue median {statistics.median(host) * 1e3:.3f} ms/step")
    print("event offsets from the step's first compute-stream event, us (median):")
    print("  (text0/vis0..text1/vis1 = the encoder graphs of the NEXT batch, launched behind this step's head)")
    for tag in ("step0", "head0", "head1", "reduce1", "text0", "text1", "vis0", "vis1", "opt1", "next step0"):
        if tag in rows:
            print(f"   {tag:12s} {statistics.median(rows[tag]):9.1f}")


if __name__ == "__main__":
    main()
