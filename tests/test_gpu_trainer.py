"""GPU: the fused train step behind the reference's trainer API (eager, hipGraph replay, pipelined)."""
import numpy as np
import pytest
import torch

from tests.helpers import assert_digest_close, load_npz

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _trainer(tmp_path, B, use_graph, n=64, **kw):
    from ultrafnd_git_amd.trainer import ForensicTrainer, TrainConfig, synthetic_cache
    cfg = TrainConfig(data_root="", ocr_phrase_pkl=None, out_dir=str(tmp_path), batch_size=B, device=DEV,
                      use_graph=use_graph, **kw)
    return ForensicTrainer(cfg, cache=synthetic_cache(n, seed=3))


def _golden_batch(z):
    b = {k: torch.from_numpy(z[f"in/{k}"]).to(DEV) for k in
         ("text_features", "audio_features", "visual_features", "temporal_features", "gnn_feat", "aux", "label")}
    return b


@pytest.mark.parametrize("use_graph", [False, True])
def test_fused_step_matches_reference(tmp_path, use_graph):
    """ForensicTrainer.train_step x3 on the B=32 golden batch == the reference's 3 optimizer steps."""
    from oracle import tier_a as O
    z = load_npz("tier_a_B32.npz")
    tr = _trainer(tmp_path, 32, use_graph)
    fus_sd, clf_sd = O.seeded_params(int(z["param_seed"]))
    tr.fusion.load_state_dict(fus_sd)
    tr.clf.load_state_dict(clf_sd)
    tr.fusion.dropout = tr.clf.dropout = tr.clf.node_dropout = 0.0
    tr.fusion.train(); tr.clf.train()
    batch = _golden_batch(z)
    for step in (1, 2, 3):
        out = tr.train_step(batch)
        loss = float(out["loss"].cpu())
        assert abs(loss - float(z[f"step{step}/loss"])) <= 5e-5, (step, loss)
        assert np.abs(out["logits"].cpu().numpy() - z[f"step{step}/logits"]).max() <= 1e-4
    for k, p in list(("fusion." + k, p) for k, p in tr.fusion.named_parameters()) + \
            list(("clf." + k, p) for k, p in tr.clf.named_parameters()):
        if p.dim():
            assert_digest_close(z, f"param_step3/{k}", p.detach(), rtol=2e-5, atol=2e-7)


def test_graph_replay_is_bit_identical_to_eager(tmp_path):
    res = []
    for use_graph in (False, True):
        torch.manual_seed(5)
        tr = _trainer(tmp_path, 16, use_graph)
        tr.fusion.train(); tr.clf.train()
        it = iter(tr.train_loader)
        for _ in range(2):
            out = tr.train_step(next(it))
        res.append((out["logits"].clone(), tr.arena.data.clone()))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])


def test_fit_and_test_contract(tmp_path, capsys):
    """fit() / test() keep the reference's contract: returns, printed lines, best.pt keys."""
    from ultrafnd_git_amd.trainer import ForensicTrainer, TrainConfig, synthetic_cache
    cache = synthetic_cache(200, seed=9)
    # make the labels plainly learnable (a label-dependent shift of 64 text dims) so AUC must move
    y = cache["labels"].astype(np.float32)
    t = cache["text"].copy()
    t[:, :64] += (2 * y[:, None] - 1) * 0.15
    cache["text"] = t / np.linalg.norm(t, axis=1, keepdims=True)
    cfg = TrainConfig(data_root="", ocr_phrase_pkl=None, out_dir=str(tmp_path), batch_size=16, epochs=8, device=DEV, lr=1e-3,
                      early_stop_patience=8)
    tr = ForensicTrainer(cfg, cache=cache)
    l0, _ = tr._epoch_loop(tr.val_loader, "val")
    best = tr.fit()
    res = tr.test()
    assert set(res) == {"test_loss", "test_acc", "test_auc", "test_precision", "test_recall", "test_f1", "test_cmcs", "test_dfdr"}
    print(out_tail := capsys.readouterr().out[-1500:])
    assert 0.8 < best <= 1.0 and res["test_auc"] > 0.8, (best, res)
    ck = torch.load(tr.ckpt_path, map_location="cpu", weights_only=True)
    assert set(ck) == {"fusion", "clf", "gnn", "cfg"} and ck["gnn"] is None
    assert "semantic.text_proj.0.weight" in ck["fusion"] and "node.trees.5.thresh.3" in ck["clf"]
    out = out_tail
    assert "train_loss=" in out and "[val]" in out and "[Test] loss=" in out
    o = tr._forward_batch(next(iter(tr.test_loader)), "test")
    assert set(o) == {"logits", "probs", "y", "forensic"} and set(o["forensic"]) == {"emotion_intensity", "semantic_conflict", "temporal_delay"}
    assert abs(tr.scheduler.get_last_lr()[0] - 1e-3 * 0.7 ** (tr.scheduler.last_epoch // 3)) < 1e-12


def test_ragged_last_batch_and_tiny_batches(tmp_path):
    tr = _trainer(tmp_path, 24, True, n=50)       # train split 35 -> batches of 24 and 11
    tr.fusion.train(); tr.clf.train()
    sizes = [int(b["label"].shape[0]) for b in tr.train_loader]
    assert sizes == [24, 11]
    loss, m = tr._epoch_loop(tr.train_loader, "train")
    assert np.isfinite(loss) and 0.0 <= m["accuracy"] <= 1.0
    tr1 = _trainer(tmp_path, 1, False, n=20)
    loss, _ = tr1._epoch_loop(tr1.train_loader, "train")
    assert np.isfinite(loss)


def test_pipelined_step_equals_plain_step(tmp_path):
    """encode_inline: all-reduce/encoder overlap schedule changes no arithmetic."""
    from oracle import encoders_ref as E
    from ultrafnd_git_amd.encoders import BertTextEncoder, ClipVisualEncoder
    from ultrafnd_git_amd.trainer import ForensicTrainer, TrainConfig, synthetic_cache
    tenc, venc = BertTextEncoder(layers=1, vocab_size=300).to(DEV), ClipVisualEncoder(layers=1).to(DEV)
    B = 8
    ids, mask = E.synthetic_tokens(3, B, 32, vocab=300, min_len=4)
    g = torch.Generator().manual_seed(1)
    batch = {"input_ids": ids.to(DEV), "attention_mask": mask.to(torch.int32).to(DEV),
             "frames": torch.randn(B, 2, 3, 224, 224, generator=g).to(DEV), "audio_features": torch.randn(B, 128, generator=g).to(DEV),
             "temporal_features": torch.randn(B, 256, generator=g).to(DEV), "gnn_feat": torch.randn(B, 128, generator=g).to(DEV),
             "aux": torch.rand(B, 2, generator=g).to(DEV), "label": torch.randint(0, 2, (B,), generator=g).to(DEV)}
    outs = []
    for mode in ("plain", "pipelined"):
        torch.manual_seed(7)
        cfg = TrainConfig(data_root="", ocr_phrase_pkl=None, out_dir=str(tmp_path), batch_size=B, device=DEV, encode_inline=True)
        tr = ForensicTrainer(cfg, cache=synthetic_cache(16, seed=1), text_encoder=tenc, visual_encoder=venc)
        tr.fusion.train(); tr.clf.train()
        if mode == "plain":
            for _ in range(3):
                out = tr.train_step(batch)
        else:
            tr.prefetch_features(batch)
            for i in range(3):
                out = tr.train_step_pipelined(batch, batch if i < 2 else None)
        outs.append((out["logits"].clone(), tr.arena.data.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    ms, n = tr.measure_gemm_time(batch, steps=1)
    assert n == 2 * 4 + 2 and ms > 0


def test_indexed_batches_gather_in_one_launch_and_equal_collated_ones(tmp_path):
    """The loader's IndexedBatch (row indices of the device-resident split) is the reference's collated dict on demand
    -- same keys, same tensors -- and the trainer's one-launch gather (ufnd_gather_rows: six cached tensors + gnn_Z)
    fills the step buffers with exactly what the per-tensor route copies: a step from either is bit-identical."""
    from ultrafnd_git_amd.trainer import ForensicTrainer, IndexedBatch, TrainConfig, synthetic_cache
    cache = synthetic_cache(96, seed=3)
    cfg = TrainConfig(data_root="", ocr_phrase_pkl=None, out_dir=str(tmp_path), batch_size=16, device="cuda", seed=7)
    a, b = ForensicTrainer(cfg, cache=cache), ForensicTrainer(cfg, cache=cache)
    ds = a.train_loader.dataset
    lazy = next(iter(a.train_loader))
    assert type(lazy) is IndexedBatch and "text_features" in lazy and "gnn_feat" not in lazy
    idx = lazy["index"]
    full = ds.gather(idx)
    assert set(full.keys()) <= set(lazy.keys())
    for k, v in full.items():
        assert torch.equal(lazy[k], v), k
    # ragged tail + repeated + out-of-order indices through the kernel
    for sel in (idx, idx[:5], torch.tensor([3, 3, 0, len(ds) - 1], device="cuda")):
        ra = a.train_step(IndexedBatch(ds, sel))
        la = float(ra["loss"])
        assert torch.equal(ra["y"], ds.y[sel])
        rb = b.train_step(b.train_loader.dataset.gather(sel))
        assert la == float(rb["loss"]) and torch.equal(ra["logits"], rb["logits"])
    assert torch.equal(a.arena.data, b.arena.data)


def test_two_launch_optimizer_is_bit_identical_to_the_four_launch_form(tmp_path):
    """ufnd_clip_adamw_step (sum of squares + step counter; norm / clip / bias corrections re-derived in every AdamW block)
    against ufnd_grad_norm + ufnd_adamw_step + ufnd_step_advance: parameters, both moments and the published scalars,
    three steps with the clip active (max_norm 0.05)."""
    res = []
    for fused in (True, False):
        torch.manual_seed(11)
        tr = _trainer(tmp_path, 2, False, grad_clip=0.05)
        tr.optim.fused = fused
        tr.fusion.train(); tr.clf.train()
        it = iter(tr.train_loader)
        scal = []
        for _ in range(3):
            tr.train_step(next(it))
            st = tr.optim.state.read()
            scal.append((int(st.step), float(st.grad_norm), float(st.clip_coef), float(st.bc1), float(st.bc2_sqrt)))
        res.append((tr.arena.data.clone(), tr.arena.exp_avg.clone(), tr.arena.exp_avg_sq.clone(), scal))
    assert res[0][3] == res[1][3] and [s[0] for s in res[0][3]] == [1, 2, 3]
    assert any(s[2] < 1.0 for s in res[0][3])                 # the clip really was active
    for a, b in zip(res[0][:3], res[1][:3]):
        assert torch.equal(a, b)


def test_two_phase_backward_writes_the_same_gradients_as_one_phase(tmp_path):
    """ufnd_fusion_backward_phase(FUSE_MLP) + (REST) == ufnd_fusion_backward, every gradient bit for bit (the bucketed
    exchange relies on it), and after the first phase alone the [classifier | fuse_mlp] bucket is already final."""
    import ctypes as C
    from ultrafnd_git_amd import _lib as L
    torch.manual_seed(21)
    tr = _trainer(tmp_path, 8, False)
    tr.fusion.train(); tr.clf.train()
    batch = next(iter(tr.train_loader))
    B = 8
    b = tr.head.bufs(B, True)
    tr._load_batch(b, batch, "train")
    tr.head.enqueue_forward(b, B, True, True)
    tr.arena.grad.fill_(float("nan"))
    tr.head.enqueue_backward(b, B, 0)
    torch.cuda.synchronize()
    whole = tr.arena.grad.clone()
    live = torch.isfinite(whole)                      # (alignment padding between parameter groups is never written)
    assert int(live.sum()) == 12_745_949
    cut = tr.reducer.buckets[0][1]
    tr.arena.grad.fill_(float("nan"))
    tr.head.enqueue_backward(b, B, 1)
    torch.cuda.synchronize()
    g1 = tr.arena.grad.clone()
    assert torch.equal(g1[:cut][live[:cut]], whole[:cut][live[:cut]])       # bucket 0 complete after phase 1
    assert torch.isnan(g1[cut:]).all()                                       # ... and phase 1 did not touch the rest
    tr.head.enqueue_backward(b, B, 2)
    torch.cuda.synchronize()
    assert torch.equal(tr.arena.grad[live], whole[live]) and torch.equal(torch.isfinite(tr.arena.grad), live)
    rc = L.lib().ufnd_fusion_backward_phase(C.byref(b["dims"]), C.byref(tr.fusion.param_table()), C.byref(tr.fusion.grad_table()),
                                            b["text"].data_ptr(), b["audio"].data_ptr(), b["visual"].data_ptr(), b["temporal"].data_ptr(),
                                            b["gnn"].data_ptr(), B, 1, b["fws"].data_ptr(), b["dfused"].data_ptr(), tr.fusion.hidden, None,
                                            tr.optim.state.ptr, None, None, 1, 7)
    assert rc == 1 and b"phase" in L.lib().ufnd_last_error()


def test_fit_with_encoders_inside_the_step_and_the_fold_guard_of_every_pass(tmp_path):
    """fit() over a cache of RAW inputs (token ids, masks, frames) with small encoders inside the step: the epoch loop's
    lookahead groups, plain train_step path and evaluation path; every encoder pass (captured graphs included) ends with the
    fold guard's launch and the scheduler reads the word asynchronously after every pass (not once per epoch)."""
    from oracle import encoders_ref as E
    from ultrafnd_git_amd.encoders import BertTextEncoder, ClipVisualEncoder
    from ultrafnd_git_amd.trainer import ForensicTrainer, TrainConfig, synthetic_cache
    tenc, venc = BertTextEncoder(layers=2, vocab_size=500), ClipVisualEncoder(layers=2)
    tenc.load_state_dict(E.seeded_weights(E.bert_shapes(layers=2, vocab=500), 11))
    venc.load_state_dict(E.seeded_weights(E.vit_shapes(layers=2), 12))
    tenc, venc = tenc.to(DEV), venc.to(DEV)
    cache = synthetic_cache(48, seed=2, with_raw=True, seq_len=128, vocab=500)
    cfg = TrainConfig(data_root="", ocr_phrase_pkl=None, out_dir=str(tmp_path), batch_size=8, epochs=2, device=DEV, encode_inline=True)
    tr = ForensicTrainer(cfg, cache=cache, text_encoder=tenc, visual_encoder=venc)
    best = tr.fit()
    res = tr.test()
    assert 0.0 <= best <= 1.0 and np.isfinite(res["test_loss"])
    # forward-only batches (validation / test) see the encoders' features too (they are written to the train step's input slot)
    vb = next(iter(tr.val_loader))
    tr._forward_batch(vb, "val")
    torch.cuda.synchronize()
    nb = len(dict.__getitem__(vb, "index"))
    assert torch.equal(tr.head.bufs(nb, False)["text"], tenc(vb["input_ids"], vb["attention_mask"]))
    assert torch.equal(tr.head.bufs(nb, False)["visual"], venc(vb["frames"]))
    assert tenc.fold_ln and venc.fold_ln and 0.0 < tenc.fold_ratio() < tenc.FOLD_GUARD_MAX       # the guard ran in every pass and did not trip
    assert tr.pipe.stats["fold_trips"] == 0 and tr.pipe.stats["pinned_replays"] > 0 and tr.pipe.stats["staged_replays"] > 0
    # a tripped guard switches the encoder to materialised LayerNorms and drops its captured graphs; new weights drop them too
    w = tenc.state_dict()
    w["embeddings.LayerNorm.bias"] = w["embeddings.LayerNorm.bias"] + 30.0
    w["encoder.layer.0.attention.output.dense.weight"] = w["encoder.layer.0.attention.output.dense.weight"] * 0.01
    tenc.load_state_dict(w)
    import warnings
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        tr._epoch_loop(tr.train_loader, "train")
    assert not tenc.fold_ln and venc.fold_ln and any("materialised" in str(c.message) for c in caught)
    assert tr.pipe.stats["fold_trips"] == 1
    tr._epoch_loop(tr.train_loader, "train")              # and training goes on, unfolded
    assert tr.pipe.stats["fold_trips"] == 1 and not tenc.fold_ln


def test_encoder_lookahead_groups_equal_plain_steps(tmp_path):
    """train_group_pipelined: the frozen encoders run ONCE over G = 3 consecutive batches (128-token samples: the fused
    projection + attention path), the head steps batch by batch.  Two groups (6 optimizer steps, the second group partial:
    2 of its 3 batches) leave logits and the whole parameter arena bit-identical to 5 plain train_step calls on the same
    batches; the per-step losses agree too."""
    from oracle import encoders_ref as E
    from ultrafnd_git_amd.encoders import BertTextEncoder, ClipVisualEncoder
    from ultrafnd_git_amd.temporal import TemporalSyncNet
    from ultrafnd_git_amd.trainer import ForensicTrainer, TrainConfig, synthetic_cache
    tenc, venc = BertTextEncoder(layers=2, vocab_size=300).to(DEV), ClipVisualEncoder(layers=1).to(DEV)
    B, G = 4, 3
    g = torch.Generator().manual_seed(2)

    def macro(seed):
        ids, mask = E.synthetic_tokens(seed, G * B, 128, vocab=300, min_len=4)
        return {"input_ids": ids.to(DEV), "attention_mask": mask.to(torch.int32).to(DEV),
                "frames": torch.randn(G * B, 1, 3, 224, 224, generator=g).to(DEV), "audio_features": torch.randn(G * B, 128, generator=g).to(DEV),
                "temporal_features": torch.randn(G * B, 256, generator=g).to(DEV), "gnn_feat": torch.randn(G * B, 128, generator=g).to(DEV),
                "aux": torch.rand(G * B, 2, generator=g).to(DEV), "label": torch.randint(0, 2, (G * B,), generator=g).to(DEV)}
    groups = [macro(31), macro(32)]
    outs = []
    for mode in ("plain", "lookahead"):
        torch.manual_seed(7)
        cfg = TrainConfig(data_root="", ocr_phrase_pkl=None, out_dir=str(tmp_path), batch_size=B, device=DEV, encode_inline=True)
        tsync = TemporalSyncNet(in_dim=768, out_dim=256).to(DEV).eval()
        tr = ForensicTrainer(cfg, cache=synthetic_cache(16, seed=1), text_encoder=tenc, visual_encoder=venc, temporal_net=tsync)
        tr.fusion.train(); tr.clf.train()
        losses = []
        if mode == "plain":
            for gi, steps in ((0, 3), (1, 2)):
                for k in range(steps):
                    out = tr.train_step({key: v[k * B:(k + 1) * B].contiguous() for key, v in groups[gi].items()})
                    losses.append(float(out["loss"].cpu()))
        else:
            tr.prefetch_features(groups[0], group=True)
            out = tr.train_group_pipelined(groups[0], groups[1])
            losses += [float(x.cpu()) for x in out["losses"]]
            out = tr.train_group_pipelined(groups[1], None, steps=2)
            losses += [float(x.cpu()) for x in out["losses"]]
        torch.cuda.synchronize()
        outs.append((out["logits"].clone(), tr.arena.data.clone(), losses, int(tr.optim.state.read().step)))
    assert outs[0][3] == outs[1][3] == 5 and outs[0][2] == outs[1][2]
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


def test_epoch_loop_with_encoder_lookahead_equals_the_plain_epoch_loop(tmp_path):
    """fit()'s training epochs with the encoders inside the step: encoder_lookahead = 3 (groups of 3 batches per encoder pass, a
    shorter tail group, the ragged last batch as a plain step) against encoder_lookahead = 1 (one train_step per batch): the same
    batches in the same order -- epoch losses, epoch metrics and the whole parameter arena are bit-identical after two epochs."""
    from oracle import encoders_ref as E
    from ultrafnd_git_amd.encoders import BertTextEncoder, ClipVisualEncoder
    from ultrafnd_git_amd.trainer import ForensicTrainer, TrainConfig, synthetic_cache
    tenc, venc = BertTextEncoder(layers=2, vocab_size=500), ClipVisualEncoder(layers=1)
    tenc.load_state_dict(E.seeded_weights(E.bert_shapes(layers=2, vocab=500), 11))
    venc.load_state_dict(E.seeded_weights(E.vit_shapes(layers=1), 12))
    tenc, venc = tenc.to(DEV), venc.to(DEV)
    cache = synthetic_cache(48, seed=2, with_raw=True, seq_len=128, vocab=500)      # train split: 33 rows = 8 batches of 4 + 1 row
    res = []
    for la in (1, 3):
        torch.manual_seed(5)
        cfg = TrainConfig(data_root="", ocr_phrase_pkl=None, out_dir=str(tmp_path / f"la{la}"), batch_size=4, epochs=2, device=DEV,
                          encode_inline=True, encoder_lookahead=la)
        tr = ForensicTrainer(cfg, cache=cache, text_encoder=tenc, visual_encoder=venc)
        ep = [tr._epoch_loop(tr.train_loader, "train") for _ in range(2)]
        torch.cuda.synchronize()
        res.append((ep, tr.arena.data.clone(), int(tr.optim.state.read().step)))
        if la == 3:
            assert sorted(k[0] for k in tr.pipe.grp_in) == [8, 8, 12, 12]            # groups of 3 and 2 batches, two slots each
    assert res[0][2] == res[1][2] == 18
    for (l0, m0), (l1, m1) in zip(res[0][0], res[1][0]):
        assert l0 == l1 and m0 == m1
    assert torch.equal(res[0][1], res[1][1])


@pytest.mark.parametrize("B", [32, 7, 130])
def test_two_call_head_step_is_bit_identical_to_the_five_call_sequence(tmp_path, B):
    """ufnd_head_forward_loss + ufnd_head_backward (TrainConfig.fused_head, the default: 22 launches at B = 32) against
    ufnd_fusion_forward -> ufnd_classifier_forward -> ufnd_softmax_ce -> ufnd_classifier_backward -> ufnd_fusion_backward_phase (26): the
    same kernels and arithmetic, so logits, probabilities, the forensic scalars, the loss, every gradient and the parameters after
    three dropout-on steps must not differ in a bit -- eager and captured, below and above the row-sliced parameter reductions (B > 64)."""
    res = {}
    for fused in (True, False):
        for graph in (False, True):
            torch.manual_seed(5)
            tr = _trainer(tmp_path, B, graph, n=max(64, 4 * B), fused_head=fused)
            assert tr.head.fused_entries == fused
            tr.fusion.train(); tr.clf.train()
            it = iter(tr.train_loader)
            for _ in range(3):
                out = tr.train_step(next(it))
            torch.cuda.synchronize()
            st = tr.optim.state.read()
            res[(fused, graph)] = (out["logits"].clone(), out["probs"].clone(), out["forensic"].clone(), float(st.loss), float(st.grad_norm),
                                   tr.arena.grad.clone(), tr.arena.data.clone())
    ref = res[(False, False)]
    for key, got in res.items():
        for a, b in zip(got, ref):
            assert (torch.equal(a, b) if isinstance(a, torch.Tensor) else a == b), key


@pytest.mark.parametrize("B", [32, 130])
def test_forty_steps_twice_leave_the_same_bits(tmp_path, B):
    """Run-to-run determinism over a long chain (dropout on, captured graphs, below and above the row-sliced parameter reductions): no kernel of
    the step may depend on which workgroup finishes first -- every reduction has a fixed order and nothing is summed by atomics."""
    res = []
    for _ in range(2):
        torch.manual_seed(5)
        tr = _trainer(tmp_path, B, True, n=max(64, 4 * B))
        tr.fusion.train(); tr.clf.train()
        it = iter(tr.train_loader)
        batches = [next(it) for _ in range(2)]
        losses = []
        for k in range(40):
            out = tr.train_step(batches[k % 2])
            losses.append(out["loss"].clone())
        torch.cuda.synchronize()
        res.append((torch.stack(losses), tr.arena.data.clone(), out["logits"].clone()))
    for a, b in zip(*res):
        assert torch.equal(a, b)
    assert bool(torch.isfinite(res[0][0]).all())


@pytest.mark.parametrize("hidden,B", [(256, 32), (1024, 16), (1024, 96)])
def test_two_call_head_step_at_other_hidden_widths(tmp_path, hidden, B):
    """The two-call head step against the five-call sequence at the other widths the kernels are built for (256: one 256-column chunk per row
    kernel, 1024: four), through HeadStep on modules built from YAMLs of that width: logits, probabilities, forensic scalars, loss and the
    whole gradient arena bit-identical (dropout on)."""
    from types import SimpleNamespace
    from ultrafnd_git_amd.arena import rehome
    from ultrafnd_git_amd.classifier import DeepTruthClassifier
    from ultrafnd_git_amd.dp import GradReducer
    from ultrafnd_git_amd.fusion import CrossModalTransformer
    from ultrafnd_git_amd.head_step import HeadStep
    from ultrafnd_git_amd.optim import FusedAdamW
    fy, cy = tmp_path / "fusion.yaml", tmp_path / "classifier.yaml"
    fy.write_text(f"hidden_dim: {hidden}\ndropout: 0.1\nuse_gnn: true\ngnn_dim: 128\n")
    cy.write_text(f"input_dim: {hidden}\nhidden_dim: {hidden}\ndropout: 0.1\nnum_classes: 2\nuse_aux: true\naux_dim: 2\nnode_trees: 6\nnode_depth: 4\n"
                  "node_tau: 10.0\ntemperature: 1.0\n")
    res = []
    for fused in (True, False):
        torch.manual_seed(11)
        fusion, clf = CrossModalTransformer(str(fy)).to(DEV), DeepTruthClassifier(str(cy)).to(DEV)
        arena = rehome([clf, fusion], ["clf.", "fusion."])
        optim = FusedAdamW(arena, seed=7)
        cfg = SimpleNamespace(use_graph=False, head_graph=False, label_smoothing=0.0, class_weighting=False, fused_head=fused)
        hs = HeadStep(cfg, torch.device(DEV), fusion, clf, optim, GradReducer(arena.ensure_grad()))
        assert hs.fused_entries == fused
        fusion.train(); clf.train()
        b = hs.bufs(B, True)
        g = torch.Generator().manual_seed(3)
        for k, d in (("text", 768), ("audio", 128), ("visual", 512), ("temporal", 256), ("gnn", 128)):
            b[k].copy_(torch.randn(B, d, generator=g))
        b["aux"].copy_(torch.rand(B, 2, generator=g))
        b["label"].copy_(torch.randint(0, 2, (B,), generator=g))
        arena.grad.fill_(float("nan"))
        hs.fwd_bwd(b, B)
        torch.cuda.synchronize()
        res.append((b["logits"].clone(), b["probs"].clone(), b["forensic"].clone(), optim.state.float_view("loss").clone(), arena.grad.clone()))
    for a, r in zip(*res):
        assert torch.equal(torch.nan_to_num(a, nan=-7.0), torch.nan_to_num(r, nan=-7.0))
    assert bool(torch.isfinite(res[0][0]).all()) and float(res[0][4][torch.isfinite(res[0][4])].abs().sum()) > 0
