#!/usr/bin/env python3
"""Experiment (GPU box): one text-encoder pass over 128 samples against TWO concurrent passes over 64 samples each on two
streams (two encoder instances with the same weights, so each has its own workspace).  The question: do two half-size
launch chains, whose compute and traffic phases are out of step, pack better than one chain of full-size launches?
usage: text_split_probe.py [samples=128]"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch

from ultrafnd_git_amd.encoders import BertTextEncoder


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    dev = torch.device("cuda")
    torch.manual_seed(0)
    e0 = BertTextEncoder().to(dev)
    e1 = BertTextEncoder().to(dev)
    e1.load_state_dict(e0.state_dict())
    e2 = BertTextEncoder().to(dev)
    e2.load_state_dict(e0.state_dict())
    g = torch.Generator().manual_seed(1)
    ids = torch.randint(0, 30522, (n, 128), generator=g).to(dev)
    lens = torch.randint(16, 129, (n,), generator=g)
    mask = (torch.arange(128)[None] < lens[:, None]).to(torch.int32).to(dev)
    h = n // 2
    ia, ma, ib, mb = ids[:h].contiguous(), mask[:h].contiguous(), ids[h:].contiguous(), mask[h:].contiguous()
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()

    def whole():
        return e0(ids, mask)

    def split():
        main_s = torch.cuda.current_stream()        # (inside a capture: the capture stream)
        sa.wait_stream(main_s)
        sb.wait_stream(main_s)
        with torch.cuda.stream(sa):
            fa = e1(ia, ma)
        with torch.cuda.stream(sb):
            fb = e2(ib, mb)
        main_s.wait_stream(sa)
        main_s.wait_stream(sb)
        return fa, fb

    ref = whole().clone()
    fa, fb = split()
    torch.cuda.synchronize()
    same = torch.equal(ref[:h], fa) and torch.equal(ref[h:], fb)
    # graphs (the trainer replays captured passes)
    gw, gs = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
    with torch.cuda.graph(gw):
        whole()
    with torch.cuda.graph(gs):
        split()
    res = {}
    for name, gr in (("one pass", gw), ("two half passes", gs), ("one pass again", gw), ("two half passes again", gs)):
        for _ in range(3):
            gr.replay()
        e_0, e_1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e_0.record()
        for _ in range(10):
            gr.replay()
        e_1.record()
        torch.cuda.synchronize()
        res[name] = e_0.elapsed_time(e_1) / 10
    print(f"{n} samples x 128 tokens, 12 layers; features bit-identical: {same}")
    for k, v in res.items():
        print(f"   {k:24s} {v:7.3f} ms")


if __name__ == "__main__":
    main()
