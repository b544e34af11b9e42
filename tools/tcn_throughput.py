#!/usr/bin/env python3
"""Row 8f-3, the sequence path of TemporalSyncNet (src/core_blocks/temporal_blocks.py:141-157): clips/s of
forward(text_seq, vis_seq) at a FakeSV-like shape (B clips x T frames, 384 + 384 channels, default TCN: 2 blocks of 128
channels, kernel 3), eval mode, timed with HIP events over a hipGraph-free loop; next to it the same arithmetic through
torch's own conv1d/batch_norm on the same device (a reference point, not a product path).  usage: tcn_throughput.py [B] [T]"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import torch.nn.functional as F

from ultrafnd_git_amd.temporal import TemporalSyncNet

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
T = int(sys.argv[2]) if len(sys.argv) > 2 else 64
torch.manual_seed(0)
net = TemporalSyncNet(in_dim=768, out_dim=256, use_tcn=True).to("cuda").eval()
ts, vs = torch.randn(B, T, 384, device="cuda"), torch.randn(B, T, 384, device="cuda")


def torch_path():
    h = torch.cat([ts, vs], dim=-1).transpose(1, 2)
    for i, (c, n) in enumerate(zip(net.tcn.convs, net.tcn.norms)):
        z = F.gelu(F.batch_norm(F.conv1d(h, c.weight, c.bias, padding="same", dilation=2 ** i), n.running_mean, n.running_var, n.weight, n.bias,
                                False, 0.1, n.eps))
        h = h + z if z.shape == h.shape else z
    return F.linear(torch.cat([h.mean(-1), h.max(-1).values], -1), net.head.weight, net.head.bias)


def timed(fn, n=50):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


with torch.no_grad():
    a, b = net(ts, vs), torch_path()
    print(f"max-abs difference to torch's conv1d/batch_norm on the device: {(a - b).abs().max().item():.2e}")
    ms, ms_t = timed(lambda: net(ts, vs)), timed(torch_path)
M = B * T
flops = 2.0 * M * 128 * (3 * 768 + 3 * 128) + 2.0 * B * 256 * 256
byts = 4.0 * (M * 768 + M * (3 * 768 + 3 * 128) * 2 + M * 128 * 6 + 128 * 3 * (768 + 128))
print(f"B={B} T={T}: {ms * 1e3:.1f} us per call -> {B / ms * 1e3:.0f} clips/s; {flops / ms / 1e9:.2f} TFLOP/s, "
      f"{byts / ms / 1e6:.1f} GB/s of {byts / 1e6:.1f} MB algorithmic (unfolded rows written and read once); torch ops: {ms_t * 1e3:.1f} us")
