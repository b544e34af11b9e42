#!/usr/bin/env python3
"""In-kernel phase times of the fused projection + attention kernel (diagnostics library, GPU box): median over
workgroups and launches of prologue (entry -> first K-step landed), K loop, projection epilogue (-> LDS images),
attention (-> end), in us of s_memrealtime (100 MHz) and the shader clock held in the K loop.   usage: attn_stamps.py [--cold] [--ln]"""
import ctypes as C
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
sys.path.insert(0, str(Path(__file__).resolve().parent))
import torch

from _diaglib import check as dcheck, diag
from ultrafnd_git_amd import _lib as L

dev = "cuda"
B, heads, H = 32, 12, 768
g = torch.Generator().manual_seed(0)
x = torch.randn(B * 128, H, generator=g).to(dev)
W = (torch.randn(3 * H, H, generator=g) / H ** 0.5).to(dev).bfloat16()
bias = torch.randn(3 * H, generator=g).to(dev)
mask = torch.ones(B, 128, dtype=torch.int32, device=dev)
ctx = torch.empty(B * 128, H, dtype=torch.bfloat16, device=dev)
ln = None
if "--ln" in sys.argv:
    xs = x.view(B * 128, 24, 32)
    st = torch.stack([xs.sum(-1), (xs * xs).sum(-1)], -1).contiguous()
    cs = W.float().sum(1).contiguous()
    ln = L.GemmLn()
    ln.a_stats, ln.colsum, ln.a_parts, ln.a_eps, ln.r_eps, ln.width = st.data_ptr(), cs.data_ptr(), 24, 1e-12, 1e-12, H
nblk = B * heads // 2
cold = "--cold" in sys.argv
junk = torch.empty(192 << 20, dtype=torch.uint8, device=dev) if cold else None
xs_rot = [x.bfloat16().clone() for _ in range(8)]
rows = []
for it in range(16):
    stt = torch.zeros(nblk, 10, dtype=torch.int64, device=dev)
    if cold:
        junk.add_(1)
    dcheck(diag().ufnd_diag_qkv_attention_stamps(xs_rot[it % 8].data_ptr(), W.data_ptr(), bias.data_ptr(), mask.data_ptr(), ctx.data_ptr(), B, heads,
                                                 C.byref(ln) if ln is not None else None, stt.data_ptr(), L.stream_ptr(torch.device(dev))), "stamps")
    torch.cuda.synchronize()
    if it >= 4:
        rows.append(stt.clone())
s = torch.stack(rows).double()          # (launch, block, 10)
rt = s[..., 1::2] * 0.01                 # realtime, us: entry, first landed, loop done, end, epilogue done
mt = s[..., 0::2]
pro = (rt[..., 1] - rt[..., 0]).median().item()
loop = (rt[..., 2] - rt[..., 1]).median().item()
epi = (rt[..., 4] - rt[..., 2]).median().item()
att = (rt[..., 3] - rt[..., 4]).median().item()
tot = (rt[..., 3] - rt[..., 0]).median().item()
span = (rt[..., 3].max(dim=1).values - rt[..., 0].min(dim=1).values).median().item()
ghz = ((mt[..., 2] - mt[..., 1]) / ((rt[..., 2] - rt[..., 1]) * 1e3)).median().item()
print(f"fused projection + attention, B=32 (192 workgroups){' cold' if cold else ''}{' LN-folded' if ln is not None else ''}: per workgroup "
      f"prologue {pro:.2f} us, K loop {loop:.2f} us ({loop / 12 * ghz * 1e3:.0f} cycles per K-step at {ghz:.2f} GHz), projection epilogue {epi:.2f} us, "
      f"attention + store {att:.2f} us, total {tot:.2f} us; launch span (first entry -> last end) {span:.2f} us")
