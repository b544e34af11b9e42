"""Host enqueue time of the head-only train step (cached features, hipGraph replay) against its GPU time: behind a GPU
spin the host queues 60 steps; 0.06 ms/step of host work against 0.37 ms on the GPU -- the step is not host-bound.
usage: head_host_time.py"""
import sys, time, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from ultrafnd_git_amd.trainer import ForensicTrainer, TrainConfig, synthetic_cache, IndexedBatch
dev = torch.device("cuda", 0)
B = 32
cfg = TrainConfig(data_root="", ocr_phrase_pkl=None, out_dir="/tmp/ufnd_bench", batch_size=B, device=str(dev), use_graph=True, seed=42)
tr = ForensicTrainer(cfg, cache=synthetic_cache(256, seed=1)); tr.fusion.train(); tr.clf.train()
ds = tr.train_loader.dataset
bl = [IndexedBatch(ds, (torch.arange(B, device=dev) + k * B) % len(ds)) for k in range(2)]
for i in range(10): tr.train_step(bl[i % 2])
torch.cuda.synchronize()
torch.cuda._sleep(int(40e-3 * 2.1e9))            # GPU busy for ~40 ms: the host runs ahead
t0 = time.perf_counter()
for i in range(60): tr.train_step(bl[i % 2])
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue {1e3 * (t1 - t0) / 60:.3f} ms/step; total incl. drain {(t2 - t0) * 1e3 / 60:.3f} ms/step")
