#!/usr/bin/env python3
"""Per-launch time of ufnd_attention_bf16 for short and long sequences (GPU box): HIP events around 20 back-to-back launches.
usage: attn_bench.py"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch

from ultrafnd_git_amd import _lib as L


def main():
    dev = torch.device("cuda")
    for B, Lq, heads in ((32, 50, 12), (128, 50, 12), (32, 128, 12), (128, 128, 12), (128, 512, 12)):
        qkv = torch.randn(B * Lq, 3 * heads * 64, device=dev).bfloat16()
        ctx = torch.empty(B * Lq, heads * 64, dtype=torch.bfloat16, device=dev)
        s = L.stream_ptr(dev)
        for _ in range(5):
            L.check(L.lib().ufnd_attention_bf16(qkv.data_ptr(), None, ctx.data_ptr(), B, Lq, heads, s), "attention")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(20):
            L.check(L.lib().ufnd_attention_bf16(qkv.data_ptr(), None, ctx.data_ptr(), B, Lq, heads, s), "attention")
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        mb = (qkv.numel() + ctx.numel()) * 2 / 1e6
        # the same work through the packed-sequence entry (always the 4-wave form)
        cu = (torch.arange(B + 1, device=dev, dtype=torch.int32) * Lq).contiguous()
        for _ in range(5):
            L.check(L.lib().ufnd_attention_bf16_varlen(qkv.data_ptr(), cu.data_ptr(), ctx.data_ptr(), B, Lq, heads, s), "varlen")
        torch.cuda.synchronize()
        e0.record()
        for _ in range(20):
            L.check(L.lib().ufnd_attention_bf16_varlen(qkv.data_ptr(), cu.data_ptr(), ctx.data_ptr(), B, Lq, heads, s), "varlen")
        e1.record()
        torch.cuda.synchronize()
        us4 = e0.elapsed_time(e1) / 20 * 1e3
        print(f"B={B:4d} L={Lq:4d} heads={heads}: {us:7.2f} us per launch, {mb:6.1f} MB -> {mb / us:5.2f} TB/s   (4-wave form via the varlen entry: {us4:7.2f} us)")


if __name__ == "__main__":
    main()
