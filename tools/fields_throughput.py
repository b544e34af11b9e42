#!/usr/bin/env python3
"""Row 8f-1 (the batched text-feature builder that replaces BERTContextEncoder.encode_fields' per-string loop,
src/core_blocks/text_blocks.py:108-128): records/s of encode_fields on synthetic FakeSV-like records -- 12 parts per
record (title, OCR, <= 10 comments), token counts ~ U{4..48} padded to the reference's max_length 256 -- with the
padding computed (as HF does) and skipped (unpad=True, identical results).   usage: fields_throughput.py [records]"""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch

from ultrafnd_git_amd.encoders import BertTextEncoder

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
enc = BertTextEncoder().to("cuda")
g = torch.Generator().manual_seed(0)
Mx, Lq = 12, 256
ids = torch.randint(0, 30522, (N, Mx, Lq), generator=g).cuda()
lens = torch.randint(4, 49, (N, Mx), generator=g)
mask = (torch.arange(Lq)[None, None, :] < lens[..., None]).int().cuda()
valid = (torch.rand(N, Mx, generator=g) < 0.8).int().cuda()
outs = {}
for unpad in (False, True):
    enc.encode_fields(ids[:32], mask[:32], valid[:32], unpad=unpad)          # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    outs[unpad] = enc.encode_fields(ids, mask, valid, unpad=unpad).clone()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    parts = int(valid.sum())
    print(f"unpad={unpad}: {N} records / {parts} parts in {dt * 1e3:.1f} ms -> {N / dt:.0f} records/s, {parts / dt:.0f} strings/s "
          f"(kept tokens {int((mask * valid[..., None]).sum())} of {parts * Lq} padded)")
print("identical:", torch.equal(outs[False], outs[True]))
