#!/usr/bin/env python3
"""Where do workgroups of a CU-masked stream run, and what does a mask do to a GEMM launch?  (GPU box.)

  1. placement: one 160-KiB-LDS workgroup per CU, all resident at once, report {XCC_ID, HW_ID} per workgroup
     for an unmasked stream and for the two candidate mask layouts ("striped": bit b -> XCD b % 8; "block":
     bit b -> XCD b // 32) -- eager and replayed from a captured hipGraph;
  2. timing: the BERT QKV GEMM (192 tiles) unmasked / on 192 CUs, the ViT QKV GEMM on 64 CUs, alone and together.
"""
import json
import sys
from collections import Counter
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
sys.path.insert(0, str(Path(__file__).resolve().parent))
import torch

from _diaglib import check as dcheck, diag
from ultrafnd_git_amd import _lib as L
from ultrafnd_git_amd.streams import MaskedStream, partition_bits

dev = torch.device("cuda:0")


def where(stream, blocks=256, spin=1 << 22, graph=False):
    out = torch.zeros(blocks, 2, dtype=torch.int32, device=dev)
    with torch.cuda.stream(stream):
        if graph:
            dcheck(diag().ufnd_diag_where(out.data_ptr(), blocks, 1 << 12, stream.cuda_stream), "where")   # warm-up
            stream.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=stream, capture_error_mode="thread_local"):
                dcheck(diag().ufnd_diag_where(out.data_ptr(), blocks, spin, torch.cuda.current_stream().cuda_stream), "where")
            out.zero_()
            g.replay()
        else:
            dcheck(diag().ufnd_diag_where(out.data_ptr(), blocks, spin, stream.cuda_stream), "where")
    stream.synchronize()
    o = out.cpu().numpy().astype("uint32")
    hw, xcc = o[:, 0], o[:, 1] & 0xF
    cu, sh, se = (hw >> 8) & 0xF, (hw >> 12) & 0x1, (hw >> 13) & 0x7
    keys = Counter((int(x), int(s), int(h), int(c)) for x, s, h, c in zip(xcc, se, sh, cu))
    per_xcc = Counter(k[0] for k in keys)
    where.last_keys = set(keys)
    return {"distinct_cus": len(keys), "per_xcc": dict(sorted(per_xcc.items())), "max_blocks_on_one_cu": max(keys.values())}


def time_gemm(calls, reps=30):
    """calls: [(stream, M, N, K, tile)] launched together each rep; per-call ms (events on its stream)."""
    bufs = []
    for (st, M, N, K, t) in calls:
        A = torch.randn(M, K, device=dev).bfloat16()
        W = torch.randn(N, K, device=dev).bfloat16()
        O = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        bufs.append((A, W, O))
    torch.cuda.synchronize()
    tot = [0.0] * len(calls)
    for r in range(reps + 5):
        evs = []
        for (st, M, N, K, t), (A, W, O) in zip(calls, bufs):
            with torch.cuda.stream(st):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(4):
                    L.check(L.lib().ufnd_gemm_bf16_ex(A.data_ptr(), W.data_ptr(), None, None, O.data_ptr(), None, M, N, K, K, K, 0, N, 0, 0, t,
                                                      st.cuda_stream), "gemm")
                e1.record()
                evs.append((e0, e1))
        torch.cuda.synchronize()
        if r >= 5:
            for i, (e0, e1) in enumerate(evs):
                tot[i] += e0.elapsed_time(e1) / 4
    return [round(t / reps * 1e3, 2) for t in tot]      # us per launch


def main():
    res = {"cu_count": L.lib().ufnd_device_cu_count()}
    if "--placement-only" in sys.argv:
        for layout in ("striped", "block"):
            bt, bv = partition_bits((192, 64), layout)
            st, sv = MaskedStream(dev, bt), MaskedStream(dev, bv)
            a = where(sv, 128); kv = where.last_keys
            b = where(st, 256)
            res[layout] = {"vis": a["distinct_cus"], "text": b["distinct_cus"], "shared": len(kv & where.last_keys)}
        print(json.dumps(res))
        return
    plain = torch.cuda.Stream(device=dev)
    res["unmasked"] = where(plain, 256)
    for layout in ("striped", "block"):
        bt, bv = partition_bits((192, 64), layout)
        st, sv = MaskedStream(dev, bt), MaskedStream(dev, bv)
        res[layout] = {"vis64_eager": where(sv, 128)}
        kv = where.last_keys
        res[layout]["vis64_graph"] = where(sv, 128, graph=True)
        res[layout]["text192_eager"] = where(st, 256)
        res[layout]["cus_shared_by_text_and_vis"] = len(kv & where.last_keys)
        # timing
        q_text, q_vis = (4096, 2304, 768, 22), (1600, 2304, 768, 15)
        res[layout]["us"] = {
            "text_qkv_unmasked_alone": time_gemm([(plain, *q_text)])[0],
            "text_qkv_192_alone": time_gemm([(st, *q_text)])[0],
            "vis_qkv_t15_unmasked_alone": time_gemm([(plain, *q_vis)])[0],
            "vis_qkv_t15_64_alone": time_gemm([(sv, *q_vis)])[0],
            "both_masked(text,vis)": time_gemm([(st, *q_text), (sv, *q_vis)]),
            "both_unmasked(text,vis)": time_gemm([(plain, *q_text), (torch.cuda.Stream(device=dev), 1600, 2304, 768, 16)]),
        }
        del st, sv
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
