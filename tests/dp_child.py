"""Child of the data-parallel GPU tests (tests/test_gpu_dp.py), run under `python -m torch.distributed.run`.

  --mode force1   world 1, backend nccl (= RCCL), ForensicTrainer(force_exchange=True): the bucketed, overlapped gradient
                  exchange is live on one GPU.  Plain and pipelined steps through ForensicTrainer must leave the parameter arena
                  bit-identical to the same steps without any exchange (a one-rank sum is the identity).
  --mode variants world 1, RCCL, forced exchange: the bf16-payload and the reduce-scatter + all-gather forms of the exchange run
                  on the device (a one-rank sum: fp32 payloads leave the gradient unchanged, the bf16 payload rounds it once).
  --mode factors2 two ranks sharing cuda:0 over gloo (host-staged like world2): the FACTOR form of the exchange (TrainConfig.grad_exchange =
                  "factors": all-gathered dY / X panels, the summed Linear gradients formed locally) against the all-reduce form and the
                  single-process full-batch step.
  --mode world2   two ranks sharing cuda:0 over gloo (device tensors staged through host memory by tests/host_staged.py's
                  Collectives subclass -- the product has no such path, its exchange is RCCL): sharded batches + summed gradients + 1/world == the single-process
                  full-batch step; then fit() / test() with sharded loaders, gathered metrics, the rank-0 checkpoint.
Prints one JSON line on rank 0."""
import argparse
import json
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
sys.path.insert(0, str(Path(__file__).resolve().parent))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np
import torch
import torch.distributed as dist

DEV = torch.device("cuda", 0)


def make_trainer(out_dir, B, use_graph=True, group=None, n=96, force_exchange=False, **kw):
    from ultrafnd_git_amd.trainer import ForensicTrainer, TrainConfig, synthetic_cache
    cfg = TrainConfig(data_root="", ocr_phrase_pkl=None, out_dir=out_dir, batch_size=B, device="cuda:0", use_graph=use_graph, **kw)
    return ForensicTrainer(cfg, cache=synthetic_cache(n, seed=3), group=group, force_exchange=force_exchange)


def dict_batches(B, n, seed):
    g = torch.Generator().manual_seed(seed)
    out = []
    for _ in range(n):
        out.append({"text_features": torch.randn(B, 768, generator=g).to(DEV), "audio_features": torch.randn(B, 128, generator=g).to(DEV),
                    "visual_features": torch.randn(B, 512, generator=g).to(DEV), "temporal_features": torch.randn(B, 256, generator=g).to(DEV),
                    "gnn_feat": torch.randn(B, 128, generator=g).to(DEV), "aux": torch.rand(B, 2, generator=g).to(DEV),
                    "label": torch.randint(0, 2, (B,), generator=g).to(DEV)})
    return out


def force1(out_dir):
    from ultrafnd_git_amd.dp import init_process_group
    init_process_group(DEV)
    res = {}
    B = 16
    arenas = {}
    for tag, active in (("exchange", True), ("plain", False)):
        for graph in (True, False):
            torch.manual_seed(5)
            tr = make_trainer(out_dir, B, use_graph=graph, force_exchange=active)
            assert tr.reducer.force == active and tr.reducer.active == active and len(tr.reducer.buckets) == 2
            tr.fusion.train(); tr.clf.train()
            for b in dict_batches(B, 3, 11):
                out = tr.train_step(b)
            st = tr.optim.state.read()
            arenas[(tag, graph)] = (tr.arena.data.clone(), out["logits"].clone(), float(st.grad_norm), int(st.step))
    ref = arenas[("plain", False)]
    res["plain_steps_bit_identical"] = all(torch.equal(v[0], ref[0]) and torch.equal(v[1], ref[1]) and v[2] == ref[2] for v in arenas.values())
    res["steps"] = ref[3]
    # pipelined steps (encoders inside the step, small encoders): exchange active vs none
    from oracle import encoders_ref as E
    from ultrafnd_git_amd.encoders import BertTextEncoder, ClipVisualEncoder
    from ultrafnd_git_amd.trainer import ForensicTrainer, TrainConfig, synthetic_cache
    wt = E.seeded_weights(E.bert_shapes(layers=2, vocab=500), 11)
    wv = E.seeded_weights(E.vit_shapes(layers=2), 12)
    pipe = {}
    for tag, active in (("exchange", True), ("plain", False)):
        torch.manual_seed(7)
        tenc, venc = BertTextEncoder(layers=2, vocab_size=500), ClipVisualEncoder(layers=2)
        tenc.load_state_dict(wt); venc.load_state_dict(wv)
        tenc, venc = tenc.to(DEV), venc.to(DEV)
        cfg = TrainConfig(data_root="", ocr_phrase_pkl=None, out_dir=out_dir, batch_size=8, device="cuda:0", use_graph=True, encode_inline=True)
        tr = ForensicTrainer(cfg, cache=synthetic_cache(32, seed=1), text_encoder=tenc, visual_encoder=venc, force_exchange=active)
        tr.fusion.train(); tr.clf.train()
        raws = []
        for k in range(3):
            ids, mask = E.synthetic_tokens(20 + k, 8, 32, vocab=500, min_len=8)
            d = dict_batches(8, 1, 30 + k)[0]
            d.update({"input_ids": ids.to(DEV), "attention_mask": mask.to(torch.int32).to(DEV), "frames": E.synthetic_frames(40 + k, 8, 1).to(DEV)})
            raws.append(d)
        tr.prefetch_features(raws[0])
        for k in range(3):
            out = tr.train_step_pipelined(raws[k], raws[k + 1] if k + 1 < 3 else None)
        torch.cuda.synchronize()
        pipe[tag] = (tr.arena.data.clone(), out["logits"].clone())
    res["pipelined_bit_identical"] = bool(torch.equal(pipe["exchange"][0], pipe["plain"][0]) and torch.equal(pipe["exchange"][1], pipe["plain"][1]))
    # encoder-lookahead groups (two batches per encoder pass) with the exchange live: the next group's encoders are enqueued
    # before the first collective of the current one
    grp = {}
    for tag, active in (("exchange", True), ("plain", False)):
        torch.manual_seed(7)
        tenc, venc = BertTextEncoder(layers=2, vocab_size=500), ClipVisualEncoder(layers=2)
        tenc.load_state_dict(wt); venc.load_state_dict(wv)
        tenc, venc = tenc.to(DEV), venc.to(DEV)
        cfg = TrainConfig(data_root="", ocr_phrase_pkl=None, out_dir=out_dir, batch_size=8, device="cuda:0", use_graph=True, encode_inline=True)
        tr = ForensicTrainer(cfg, cache=synthetic_cache(32, seed=1), text_encoder=tenc, visual_encoder=venc, force_exchange=active)
        tr.fusion.train(); tr.clf.train()
        groups = []
        for k in range(2):
            ids, mask = E.synthetic_tokens(50 + k, 16, 32, vocab=500, min_len=8)
            d = dict_batches(16, 1, 60 + k)[0]
            d.update({"input_ids": ids.to(DEV), "attention_mask": mask.to(torch.int32).to(DEV), "frames": E.synthetic_frames(70 + k, 16, 1).to(DEV)})
            groups.append(d)
        tr.prefetch_features(groups[0], group=True)
        tr.train_group_pipelined(groups[0], groups[1])
        out = tr.train_group_pipelined(groups[1], None)
        torch.cuda.synchronize()
        grp[tag] = (tr.arena.data.clone(), out["logits"].clone(), int(tr.optim.state.read().step))
    res["lookahead_bit_identical"] = bool(torch.equal(grp["exchange"][0], grp["plain"][0]) and torch.equal(grp["exchange"][1], grp["plain"][1])
                                          and grp["exchange"][2] == grp["plain"][2] == 4)
    # trainable encoders (TrainConfig.train_encoders): four buckets -- the head's two, then the text and the visual encoder's arena
    # ranges, started as each backward (text on the step's stream, visual on its second stream) completes
    enc = {}
    for tag, active in (("exchange", True), ("plain", False)):
        torch.manual_seed(7)
        tenc, venc = BertTextEncoder(layers=2, vocab_size=500), ClipVisualEncoder(layers=2)
        tenc.load_state_dict(wt); venc.load_state_dict(wv)
        tenc, venc = tenc.to(DEV), venc.to(DEV)
        cfg = TrainConfig(data_root="", ocr_phrase_pkl=None, out_dir=out_dir, batch_size=8, device="cuda:0", use_graph=True, encode_inline=True,
                          train_encoders=True)
        tr = ForensicTrainer(cfg, cache=synthetic_cache(32, seed=1), text_encoder=tenc, visual_encoder=venc, force_exchange=active)
        tr.fusion.train(); tr.clf.train()
        nb = len(tr.reducer.buckets)
        for k in range(2):
            ids, mask = E.synthetic_tokens(80 + k, 8, 32, vocab=500, min_len=8)
            d = dict_batches(8, 1, 90 + k)[0]
            d.update({"input_ids": ids.to(DEV), "attention_mask": mask.to(torch.int32).to(DEV), "frames": E.synthetic_frames(95 + k, 8, 1).to(DEV)})
            out = tr.train_step(d)
        torch.cuda.synchronize()
        enc[tag] = (tr.arena.data.clone(), out["logits"].clone(), nb, int(tr.arena.data.numel()))
    res["train_encoders_bit_identical"] = bool(torch.equal(enc["exchange"][0], enc["plain"][0]) and torch.equal(enc["exchange"][1], enc["plain"][1]))
    res["train_encoders_buckets"] = enc["exchange"][2]
    res["train_encoders_arena"] = enc["exchange"][3]
    res["backend"] = dist.get_backend()
    print(json.dumps(res))
    dist.destroy_process_group()


def world2(out_dir):
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    B = 8                                            # global batch; every rank takes B / world rows
    batches = dict_batches(B, 3, 17)
    ref = None
    if rank == 0:                                    # single-process full-batch reference, BEFORE the group exists
        torch.manual_seed(5)
        tr0 = make_trainer(os.path.join(out_dir, "ref"), B, use_graph=False)
        tr0.fusion.dropout = tr0.clf.dropout = tr0.clf.node_dropout = 0.0
        tr0.head.step_bufs.clear()
        tr0.fusion.train(); tr0.clf.train()
        for b in batches:
            o = tr0.train_step(b)
        ref = (tr0.arena.data.clone(), float(tr0.optim.state.read().grad_norm))
        del tr0
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from host_staged import HostStagedCollectives
    comm = HostStagedCollectives()
    torch.manual_seed(5)
    tr = make_trainer(out_dir, B // world, use_graph=True, group=comm)
    assert tr.world == world and tr.reducer.active and abs(tr.reducer.grad_scale - 0.5) < 1e-12
    tr.fusion.dropout = tr.clf.dropout = tr.clf.node_dropout = 0.0
    tr.head.step_bufs.clear()
    tr.fusion.train(); tr.clf.train()
    for b in batches:
        shard = {k: v[rank::world].contiguous() for k, v in b.items()}
        tr.train_step(shard)
    torch.cuda.synchronize()
    res = {}
    gn = float(tr.optim.state.read().grad_norm)
    if rank == 0:
        err = (tr.arena.data - ref[0]).abs().max().item()
        scale = ref[0].abs().max().item()
        res.update({"param_max_abs_err": err, "param_scale": scale, "grad_norm": gn, "grad_norm_ref": ref[1],
                    "via_host": comm.staged_calls > 0})
    # every rank holds the same parameters after the exchange
    mine = tr.arena.data.cpu()
    other = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(other, mine)
    res["ranks_agree"] = bool(all(torch.equal(o, other[0]) for o in other))
    # fit / test with sharded loaders, gathered metrics, rank-0 checkpoint + broadcast
    from ultrafnd_git_amd.trainer import ForensicTrainer, TrainConfig, synthetic_cache
    cache = synthetic_cache(150, seed=9)
    y = cache["labels"].astype(np.float32)
    t = cache["text"].copy()
    t[:, :64] += (2 * y[:, None] - 1) * 0.15
    cache["text"] = t / np.linalg.norm(t, axis=1, keepdims=True)
    cfg = TrainConfig(data_root="", ocr_phrase_pkl=None, out_dir=os.path.join(out_dir, "fit"), batch_size=8, epochs=3, device="cuda:0",
                      lr=1e-3, early_stop_patience=8)
    tr2 = ForensicTrainer(cfg, cache=cache, group=comm)
    best = tr2.fit()
    out = tr2.test()
    vals = torch.tensor([best, out["test_auc"], out["test_loss"]], dtype=torch.float64)
    allv = [torch.empty_like(vals) for _ in range(world)]
    dist.all_gather(allv, vals)
    res["fit_ranks_agree"] = bool(all(torch.equal(a, allv[0]) for a in allv))
    res["best_val_auc"], res["test_auc"] = float(best), float(out["test_auc"])
    res["ckpt_exists"] = os.path.exists(tr2.ckpt_path)
    n_val = len(tr2.val_loader.dataset)
    per_rank = torch.tensor([sum(dict.__getitem__(b, "index").numel() for b in tr2.val_loader)], dtype=torch.int64)
    cnt = [torch.empty_like(per_rank) for _ in range(world)]
    dist.all_gather(cnt, per_rank)
    res["val_rows_total"], res["val_rows_split"] = int(sum(int(c) for c in cnt)), n_val
    if rank == 0:
        print(json.dumps(res))
    dist.destroy_process_group()


def factors2(out_dir):
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    B = 16                                           # global batch; every rank takes B / world rows
    batches = dict_batches(B, 3, 23)
    ref = None
    if rank == 0:                                    # single-process full-batch reference, BEFORE the group exists
        torch.manual_seed(5)
        tr0 = make_trainer(os.path.join(out_dir, "ref"), B, use_graph=False)
        tr0.fusion.dropout = tr0.clf.dropout = tr0.clf.node_dropout = 0.0
        tr0.head.step_bufs.clear()
        tr0.fusion.train(); tr0.clf.train()
        for b in batches:
            tr0.train_step(b)
        ref = (tr0.arena.data.clone(), tr0.arena.grad.clone(), float(tr0.optim.state.read().grad_norm))
        del tr0
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from host_staged import HostStagedCollectives
    got = {}
    for tag, kw, graph in (("all_reduce", {}, True), ("factors", dict(grad_exchange="factors"), True), ("factors_eager", dict(grad_exchange="factors"), False)):
        comm = HostStagedCollectives()
        torch.manual_seed(5)
        tr = make_trainer(os.path.join(out_dir, tag), B // world, use_graph=graph, group=comm, **kw)
        assert tr.world == world and tr.reducer.active
        assert (type(tr.reducer).__name__ == "FactorExchange") == tag.startswith("factors")
        tr.fusion.dropout = tr.clf.dropout = tr.clf.node_dropout = 0.0
        tr.head.step_bufs.clear()
        tr.fusion.train(); tr.clf.train()
        for b in batches:
            tr.train_step({k: v[rank::world].contiguous() for k, v in b.items()})
        torch.cuda.synchronize()
        got[tag] = (tr.arena.data.clone(), tr.arena.grad.clone(), float(tr.optim.state.read().grad_norm), tr.reducer.wire_bytes(),
                    [list(r) for r in getattr(tr.reducer, "small_ranges", [])], tr.arena.n_grad)
    res = {}
    rel = lambda a, b: float(((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item())
    # the factor form sums a dW element over world x B rows in one chain, the all-reduce adds per-rank chains: fp32 summation order only
    res["grad_rel_vs_all_reduce"] = rel(got["factors"][1], got["all_reduce"][1])
    res["param_rel_vs_all_reduce"] = rel(got["factors"][0], got["all_reduce"][0])
    res["graph_equals_eager"] = bool(torch.equal(got["factors"][0], got["factors_eager"][0]) and torch.equal(got["factors"][1], got["factors_eager"][1]))
    res["wire_bytes"] = {k: v[3] for k, v in got.items()}
    res["small_ranges"], res["n_grad"] = got["factors"][4], got["factors"][5]
    mine = got["factors"][0].cpu()
    other = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(other, mine)
    res["ranks_agree"] = bool(all(torch.equal(o, other[0]) for o in other))
    if rank == 0:
        # the summed gradient of the two half-batch ranks is 2 x the full-batch mean gradient (grad_scale = 1 / world applies later)
        res["grad_rel_vs_full_batch"] = rel(got["factors"][1] * 0.5, ref[1])
        res["param_max_abs_err"] = (got["factors"][0] - ref[0]).abs().max().item()
        res["param_scale"] = ref[0].abs().max().item()
        res["grad_norm"], res["grad_norm_ref"] = got["factors"][2], ref[2]
        print(json.dumps(res))
    dist.destroy_process_group()


def variants(out_dir):
    """The exchange variants on the device (RCCL, one rank): every form must run inside the step (graphs, two buckets, overlap),
    fp32 payloads must not change a bit of a one-rank sum, the bf16 payload must equal one bf16 rounding of the gradient."""
    from ultrafnd_git_amd.dp import init_process_group
    init_process_group(DEV)
    res, B, arenas = {}, 16, {}
    for tag, kw in (("plain", None), ("ar_fp32", dict(grad_payload="fp32", grad_exchange="all_reduce")),
                    ("rsag_fp32", dict(grad_payload="fp32", grad_exchange="rs_ag")), ("ar_bf16", dict(grad_payload="bf16", grad_exchange="all_reduce")),
                    ("rsag_bf16", dict(grad_payload="bf16", grad_exchange="rs_ag")), ("factors", dict(grad_exchange="factors"))):
        torch.manual_seed(5)
        tr = make_trainer(out_dir, B, use_graph=True, force_exchange=kw is not None, **(kw or {}))
        tr.fusion.train(); tr.clf.train()
        for b in dict_batches(B, 2, 11):
            out = tr.train_step(b)
        torch.cuda.synchronize()
        arenas[tag] = (tr.arena.data.clone(), tr.arena.grad.clone())
        res[tag + "_wire_MB"] = round(tr.reducer.wire_bytes() / 1e6, 2)
    ref = arenas["plain"]
    res["fp32_forms_bit_identical"] = bool(torch.equal(arenas["ar_fp32"][0], ref[0]) and torch.equal(arenas["rsag_fp32"][0], ref[0]))
    # the factor form with one rank forms every Linear gradient from its own pack: the same rows in the same order -- no bit changes
    res["factors_bit_identical"] = bool(torch.equal(arenas["factors"][0], ref[0]) and torch.equal(arenas["factors"][1], ref[1]))
    res["bf16_forms_agree"] = bool(torch.equal(arenas["ar_bf16"][0], arenas["rsag_bf16"][0]))
    g = arenas["ar_bf16"][1]
    res["bf16_grad_is_bf16_valued"] = bool(torch.equal(g, g.to(torch.bfloat16).float()))
    res["bf16_param_max_rel"] = float(((arenas["ar_bf16"][0] - ref[0]).abs().max() / ref[0].abs().max()).item())
    res["backend"] = dist.get_backend()
    print(json.dumps(res))
    dist.destroy_process_group()


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", required=True)
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    torch.cuda.set_device(0)
    {"force1": force1, "world2": world2, "variants": variants, "factors2": factors2}[a.mode](a.out)
