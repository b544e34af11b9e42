#!/usr/bin/env python3
"""Train / evaluate the fusion head on MI355X -- same flags and printed result keys as the
reference's entry point (run_train_eval.py:28-47,102-109 of Nuralamsiddik16/Ultrafnd_git).

  python run_train_eval.py --data_root <dir with feature_cache.npz> --out_dir outputs_v2 --epochs 12
  torchrun --standalone --local-addr 127.0.0.1 --nproc-per-node 8 run_train_eval.py ...   # data parallel

`--data_root` must hold `feature_cache.npz` (the cache dict of build_gnn_cache_from_raw_dataset,
plus gnn_Z and split_{train,val,test}); building it from the raw FakeSV corpus is out of scope
(SURVEY.md section 2 row 8).  `--synthetic N` trains on a FakeSV-shaped synthetic cache instead.
"""
import argparse
import os
from pathlib import Path

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch  # noqa: E402

from ultrafnd_git_amd.trainer import ForensicTrainer, TrainConfig, synthetic_cache  # noqa: E402


def parse_args():
    p = argparse.ArgumentParser(description="Ultrafnd fusion head on MI355X -- train/test")
    p.add_argument("--data_root", type=str, default="/Volumes/SR_disk/FakeSV")
    p.add_argument("--ocr_phrase_pkl", type=str, default="")
    p.add_argument("--out_dir", type=str, default="outputs_v2")
    p.add_argument("--epochs", type=int, default=12)
    p.add_argument("--batch_size", type=int, default=16)
    p.add_argument("--lr", type=float, default=2e-4)
    p.add_argument("--weight_decay", type=float, default=1e-4)
    p.add_argument("--gnn_dim", type=int, default=128)
    p.add_argument("--gnn_overlap_thresh", type=float, default=0.12)
    p.add_argument("--seed", type=int, default=42)
    p.add_argument("--cpu", action="store_true", help="accepted for flag parity; refused: there is no CPU path")
    p.add_argument("--no_gnn", action="store_true")
    p.add_argument("--eval_only", action="store_true")
    p.add_argument("--synthetic", type=int, default=0, help="train on N synthetic FakeSV-shaped samples")
    p.add_argument("--no_graph", action="store_true", help="launch kernels eagerly instead of replaying a hipGraph")
    return p.parse_args()


def main():
    args = parse_args()
    if args.cpu:
        raise SystemExit("--cpu: this package runs the fusion step on a HIP device only")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        from ultrafnd_git_amd.dp import init_process_group
        init_process_group(torch.device("cuda", local))
    rank = int(os.environ.get("RANK", "0"))
    out_dir = Path(args.out_dir).expanduser()
    out_dir.mkdir(parents=True, exist_ok=True)
    torch.manual_seed(args.seed)
    if rank == 0:
        print("==== Ultrafnd fusion head (MI355X) ====")
        print(f"Device:          cuda:{local} x {world}")
        print(f"Data root:       {args.data_root}")
        print(f"Output dir:      {out_dir}")
        print(f"Epochs:          {args.epochs}")
        print(f"Batch size:      {args.batch_size}")
        print(f"Use GNN:         {not args.no_gnn}")
        print("=======================================")
    cfg = TrainConfig(data_root=str(Path(args.data_root).expanduser()), ocr_phrase_pkl=args.ocr_phrase_pkl or None,
                      out_dir=str(out_dir), batch_size=args.batch_size, epochs=args.epochs, lr=args.lr,
                      weight_decay=args.weight_decay, gnn_dim=args.gnn_dim, gnn_overlap_thresh=args.gnn_overlap_thresh,
                      seed=args.seed, use_mps=False, use_gnn=(not args.no_gnn), save_best=True, device=f"cuda:{local}",
                      use_graph=not args.no_graph)
    cache = synthetic_cache(args.synthetic, seed=args.seed, gnn_dim=args.gnn_dim) if args.synthetic else None
    trainer = ForensicTrainer(cfg, cache=cache)
    if not args.eval_only:
        if rank == 0:
            print("\n>>> Training...")
        trainer.fit()
    if rank == 0:
        print("\n>>> Testing best checkpoint...")
    results = trainer.test()
    if rank == 0:
        print("\n==== Final Results ====")
        print(f"Test Loss: {results['test_loss']:.4f}")
        print(f"Test Acc : {results['test_acc']:.4f}")
        print(f"Test AUC : {results['test_auc']:.4f}")
        for k in ("test_precision", "test_recall", "test_f1", "test_cmcs", "test_dfdr"):
            if k in results:
                print(f"{k.replace('test_', 'Test ').title()}: {results[k]:.4f}")
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
