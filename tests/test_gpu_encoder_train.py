"""GPU: trainable encoders (ultrafnd_git_amd/encoder_train.py; TrainConfig.train_encoders) -- every encoder gradient against the
oracle's autograd (oracle/encoders_ref.py, pinned to the installed third-party classes' own autograd by
tests/golden/tier_b_grads.npz), then the whole step through ForensicTrainer.

The reference never trains its encoders (src/core_blocks/text_blocks.py:52 `.eval()`, :63 `inference_mode`): "parity unpinned by
the reference".  Tolerances are bf16-derived: every GEMM operand (activations, upstream gradients, weights) is rounded to bf16
(2^-9 relative), accumulation is fp32; a gradient tensor after n layers agrees with the fp32 autograd to about
2^-9 * sqrt(4 n) in relative L2 (2 layers: ~6e-3, 12 layers: ~1.4e-2); the bound is 2.5 x that, plus an absolute floor of 2e-3 of
the LARGEST gradient tensor's scale for tensors whose own gradient is (nearly) zero -- the key bias's exactly is, by softmax's
shift invariance."""
import json

import pytest
import torch

from tests.helpers import load_npz

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _standalone(bp_cls, enc):
    """An encoder with a private arena (no trainer): masters bound, gradient buffer allocated."""
    from ultrafnd_git_amd.arena import FlatArena
    bp = bp_cls(enc)
    arena = FlatArena([list(g) for g in bp.groups()], [], torch.device(DEV, torch.cuda.current_device()))
    bp.bind(arena, "")
    arena.ensure_grad().fill_(float("nan"))
    return bp, arena


def _compare(arena, ref, rel_bound, what):
    """Every gradient tensor: relative L2 against the oracle's, with the absolute floor the module docstring states."""
    top = max(g.norm().item() / max(1, g.numel()) ** 0.5 for g in ref.values())          # the largest per-element gradient scale
    worst = ("", 0.0)
    for k, r in ref.items():
        got = arena.grad_view(k).cpu()
        assert torch.isfinite(got).all(), (what, k)
        err = (got - r).norm().item()
        rel = err / max(r.norm().item(), 1e-30)
        floor = 2e-3 * top * max(1, r.numel()) ** 0.5
        if err > floor and rel > worst[1]:
            worst = (k, rel)
        assert err <= rel_bound * r.norm().item() + floor, (what, k, rel, err, floor)
    print(f"{what}: worst relative-L2 gradient error {worst[1]:.3e} ({worst[0]}), bound {rel_bound:.1e}")


@pytest.mark.parametrize("tag", ["bert2_L64", "bert2_L128"])
def test_text_encoder_gradients_vs_oracle_autograd(tag):
    from oracle import encoders_ref as E
    from ultrafnd_git_amd.encoder_train import TextBackprop
    from ultrafnd_git_amd.encoders import BertTextEncoder
    z = load_npz("tier_b_grads.npz")
    meta = json.loads(str(z[f"{tag}/meta"]))
    w = E.seeded_weights(E.bert_shapes(layers=meta["layers"], vocab=meta["vocab"]), meta["weight_seed"])
    ids, mask = torch.from_numpy(z[f"{tag}/ids"]), torch.from_numpy(z[f"{tag}/mask"])
    enc = BertTextEncoder(layers=meta["layers"], vocab_size=meta["vocab"])
    enc.load_state_dict(w)
    enc = enc.to(DEV)
    bp, arena = _standalone(TextBackprop, enc)
    feat = bp.forward_train(ids, mask).clone()
    ref_feat, ref = E.text_feature_grads(w, ids, mask, meta["loss_seed"])
    assert (feat.cpu() - ref_feat).abs().max().item() <= 1.2e-3
    g = torch.Generator().manual_seed(meta["loss_seed"])
    bp.backward(torch.randn(ref_feat.shape, generator=g).to(DEV))          # d probe_loss / d features
    _compare(arena, ref, 1.6e-2, f"{tag} (2 layers)")
    first = arena.grad.clone()
    arena.grad.fill_(float("nan"))
    bp.forward_train(ids, mask)
    bp.backward(torch.randn(ref_feat.shape, generator=torch.Generator().manual_seed(meta["loss_seed"])).to(DEV))
    assert torch.equal(torch.nan_to_num(arena.grad, nan=-7.0), torch.nan_to_num(first, nan=-7.0))      # no atomics: identical bits (padding stays NaN)
    # the weight-gradient products on the backward's own stream instead of the second one (the default: double-buffered operand copies, events): same bits
    assert bp.overlap_wgrad
    bp.overlap_wgrad = False
    arena.grad.fill_(float("nan"))
    bp.forward_train(ids, mask)
    bp.backward(torch.randn(ref_feat.shape, generator=torch.Generator().manual_seed(meta["loss_seed"])).to(DEV))
    torch.cuda.synchronize()
    assert torch.equal(torch.nan_to_num(arena.grad, nan=-7.0), torch.nan_to_num(first, nan=-7.0))


@pytest.mark.parametrize("tag", ["vit2_F1", "vit2_F2"])
def test_visual_encoder_gradients_vs_oracle_autograd(tag):
    from oracle import encoders_ref as E
    from ultrafnd_git_amd.encoder_train import VisualBackprop
    from ultrafnd_git_amd.encoders import ClipVisualEncoder
    z = load_npz("tier_b_grads.npz")
    meta = json.loads(str(z[f"{tag}/meta"]))
    w = E.seeded_weights(E.vit_shapes(layers=meta["layers"]), meta["weight_seed"])
    frames = E.synthetic_frames(meta["frame_seed"], meta["B"], meta["F"])
    enc = ClipVisualEncoder(layers=meta["layers"])
    enc.load_state_dict(w)
    enc = enc.to(DEV)
    bp, arena = _standalone(VisualBackprop, enc)
    feat = bp.forward_train(frames).clone()
    ref_feat, ref = E.visual_feature_grads(w, frames, meta["loss_seed"])
    assert (feat.cpu() - ref_feat).abs().max().item() <= 1.5e-3
    g = torch.Generator().manual_seed(meta["loss_seed"])
    bp.backward(torch.randn(ref_feat.shape, generator=g).to(DEV))
    _compare(arena, ref, 1.6e-2, f"{tag} (2 layers)")


def test_full_depth_text_encoder_gradients_sampled_tensors():
    """12 layers (BERT-base geometry, small vocabulary so that the CPU autograd stays quick): first, middle and last layer's
    tensors and the embeddings."""
    from oracle import encoders_ref as E
    from ultrafnd_git_amd.encoder_train import TextBackprop
    from ultrafnd_git_amd.encoders import BertTextEncoder
    w = E.seeded_weights(E.bert_shapes(layers=12, vocab=1000), 43)
    ids, mask = E.synthetic_tokens(143, 4, 128, vocab=1000)
    enc = BertTextEncoder(layers=12, vocab_size=1000)
    enc.load_state_dict(w)
    enc = enc.to(DEV)
    bp, arena = _standalone(TextBackprop, enc)
    feat = bp.forward_train(ids, mask).clone()
    ref_feat, ref = E.text_feature_grads(w, ids, mask, 5)
    bp.backward(torch.randn(ref_feat.shape, generator=torch.Generator().manual_seed(5)).to(DEV))
    keep = {k: v for k, v in ref.items() if k.startswith(("encoder.layer.0.", "encoder.layer.6.", "encoder.layer.11.", "embeddings."))}
    _compare(arena, keep, 3.0e-2, "BERT 12 layers, sampled tensors")      # (measured 1.44e-2; was 4e-2)
    per = _per_layer(arena, ref, "encoder.layer.{}.", (0, 6, 11))
    print("BERT 12 layers, per-layer relative L2:", {k: f"{v:.3e}" for k, v in per.items()})
    assert per[11] <= BERT12_LAYER_BOUNDS[11] and per[6] <= BERT12_LAYER_BOUNDS[6] and per[0] <= BERT12_LAYER_BOUNDS[0], per


def _per_layer(arena, ref, prefix_fmt, layers):
    """Relative L2 gradient error of all tensors of a layer taken together, for the layers named (localises a regression)."""
    out = {}
    for li in layers:
        pre = prefix_fmt.format(li)
        num = den = 0.0
        for k, r in ref.items():
            if k.startswith(pre):
                got = arena.grad_view(k).cpu().double()
                num += float((got - r.double()).pow(2).sum())
                den += float(r.double().pow(2).sum())
        out[li] = (num / max(den, 1e-300)) ** 0.5
    return out


def test_full_depth_visual_encoder_gradients_sampled_tensors():
    """12 layers of ViT-B/32 (the check the text encoder already had: VERDICT r3 weak #1a): first, middle and last layer's tensors,
    the embeddings and the projection; per-layer errors are reported and bounded separately."""
    from oracle import encoders_ref as E
    from ultrafnd_git_amd.encoder_train import VisualBackprop
    from ultrafnd_git_amd.encoders import ClipVisualEncoder
    w = E.seeded_weights(E.vit_shapes(layers=12), 44)
    frames = E.synthetic_frames(144, 4, 1)
    enc = ClipVisualEncoder(layers=12)
    enc.load_state_dict(w)
    enc = enc.to(DEV)
    bp, arena = _standalone(VisualBackprop, enc)
    feat = bp.forward_train(frames).clone()
    ref_feat, ref = E.visual_feature_grads(w, frames, 6)
    assert (feat.cpu() - ref_feat).abs().max().item() <= 2.5e-3
    bp.backward(torch.randn(ref_feat.shape, generator=torch.Generator().manual_seed(6)).to(DEV))
    keep = {k: v for k, v in ref.items() if ".layers.0." in k or ".layers.6." in k or ".layers.11." in k or "embeddings" in k or "projection" in k
            or "layrnorm" in k or "post_layernorm" in k}
    assert len(keep) >= 3 * 16
    _compare(arena, keep, 1.5e-2, "ViT 12 layers, sampled tensors")      # (measured 7.0e-3)
    per = _per_layer(arena, ref, "vision_model.encoder.layers.{}.", (0, 6, 11))
    print("ViT 12 layers, per-layer relative L2:", {k: f"{v:.3e}" for k, v in per.items()})
    assert per[11] <= VIT12_LAYER_BOUNDS[11] and per[6] <= VIT12_LAYER_BOUNDS[6] and per[0] <= VIT12_LAYER_BOUNDS[0], per


# per-layer bounds = 2 x measured on MI355X (the error grows with the distance from the loss: bf16 rounding of every upstream gradient)
VIT12_LAYER_BOUNDS = {11: 1.35e-2, 6: 1.45e-2, 0: 1.6e-2}       # measured 6.7e-3 / 7.1e-3 / 7.9e-3
BERT12_LAYER_BOUNDS = {11: 1.85e-2, 6: 2.4e-2, 0: 3.0e-2}       # measured 9.1e-3 / 1.17e-2 / 1.47e-2


def test_outlier_shaped_weights_through_the_backward_path():
    """Outlier-shaped weights (four hidden dimensions with 5x LayerNorm gains, every LayerNorm bias offset by half a standard
    deviation) through forward_train + backward of a 4-layer BERT: the gradients hold the bound of the Gaussian-weight tests.
    Why 5x and not the forward test's 20x (tests/test_gpu_fullsize.py): with Gaussian projections behind them, 20x gains in EVERY
    LayerNorm let the hot dimensions dominate each next LayerNorm's variance, the other dimensions shrink 16-fold per sublayer and
    the attention logits reach ~100 (measured on this construction: |q.k| / 8 ~ 120): the softmax saturates and its gradient
    P (dP - delta) becomes a small difference of rounded numbers -- the q / k weight gradients of ANY implementation with bf16
    operands are then 60 % off the fp32 autograd (measured here: 0.64 relative L2; the pooled forward features still agree to
    9e-4).  That is a property of the synthetic network (trained encoders keep their logits at O(10)), recorded in DESIGN section 2,
    not a case a bf16 backward can be held to."""
    from oracle import encoders_ref as E
    from ultrafnd_git_amd.encoder_train import TextBackprop
    from ultrafnd_git_amd.encoders import BertTextEncoder
    w = E.seeded_weights(E.bert_shapes(layers=4, vocab=800), 93)
    hot = [7, 300, 511, 640]
    for k in list(w):
        if k.endswith("LayerNorm.weight"):
            w[k] = w[k].clone()
            w[k][hot] *= 5.0
        if k.endswith("LayerNorm.bias"):
            w[k] = w[k] + 0.5
    ids, mask = E.synthetic_tokens(193, 4, 96, vocab=800)
    enc = BertTextEncoder(layers=4, vocab_size=800)
    enc.load_state_dict(w)
    enc = enc.to(DEV)
    bp, arena = _standalone(TextBackprop, enc)
    feat = bp.forward_train(ids, mask).clone()
    ref_feat, ref = E.text_feature_grads(w, ids, mask, 7)
    from tests.helpers import feature_errors
    fe = feature_errors(feat.cpu(), ref_feat)
    print("outlier-shaped weights, training forward:", fe)
    assert fe["rel_l2"] <= 3.0e-3 and fe["one_minus_cos"] <= 3.0e-6, fe      # (measured 1.2e-3 / 7.6e-7)
    bp.backward(torch.randn(ref_feat.shape, generator=torch.Generator().manual_seed(7)).to(DEV))
    _compare(arena, ref, OUTLIER_BWD_BOUND, "outlier-shaped weights (4 layers, 5x gains), backward")


OUTLIER_BWD_BOUND = 6.0e-3      # 3 x measured on MI355X (2.1e-3)


def test_trainer_step_with_trainable_encoders_vs_oracle(tmp_path):
    """ForensicTrainer(train_encoders=True): one step -- encoder forwards, head forward / CE / backward, feature gradients,
    encoder backwards, ONE global-norm clip + AdamW over the joint arena -- against the oracle's autograd over the same
    composition (dropout off): loss, global gradient norm, head and encoder gradients, and the update's direction."""
    import torch.nn.functional as F
    from oracle import encoders_ref as E
    from oracle import tier_a as O
    from ultrafnd_git_amd.encoders import BertTextEncoder, ClipVisualEncoder
    from ultrafnd_git_amd.trainer import ForensicTrainer, TrainConfig, synthetic_cache
    B, Lq = 4, 64
    wt = E.seeded_weights(E.bert_shapes(layers=2, vocab=500), 11)
    wv = E.seeded_weights(E.vit_shapes(layers=2), 12)
    tenc, venc = BertTextEncoder(layers=2, vocab_size=500), ClipVisualEncoder(layers=2)
    tenc.load_state_dict(wt); venc.load_state_dict(wv)
    tenc, venc = tenc.to(DEV), venc.to(DEV)
    ids, mask = E.synthetic_tokens(13, B, Lq, vocab=500, min_len=8)
    frames = E.synthetic_frames(14, B, 1)
    batch = O.seeded_batch(15, B)
    fus_sd, clf_sd = O.seeded_params(1234)
    cfg = TrainConfig(data_root="", ocr_phrase_pkl=None, out_dir=str(tmp_path), batch_size=B, device=DEV, use_graph=False, encode_inline=True,
                      train_encoders=True, grad_clip=1e9)
    tr = ForensicTrainer(cfg, cache=synthetic_cache(16, seed=1), text_encoder=tenc, visual_encoder=venc)
    tr.fusion.load_state_dict(fus_sd); tr.clf.load_state_dict(clf_sd)
    tr.fusion.dropout = tr.clf.dropout = tr.clf.node_dropout = 0.0
    tr.head.step_bufs.clear()
    tr.fusion.train(); tr.clf.train()
    gb = {k: v.to(DEV) for k, v in batch.items()}
    gb.update({"input_ids": ids.to(DEV), "attention_mask": mask.to(torch.int32).to(DEV), "frames": frames.to(DEV)})
    before = tr.arena.data.clone()
    out = tr.train_step(gb)
    st = tr.optim.state.read()
    # oracle: autograd through encoders + head
    wtl = {k: v.clone().requires_grad_(True) for k, v in wt.items()}
    wvl = {k: v.clone().requires_grad_(True) for k, v in wv.items()}
    fl = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in fus_sd.items()}
    cl = {k: v.clone().requires_grad_(v.is_floating_point() and not k.endswith("tau")) for k, v in clf_sd.items()}
    rb = dict(batch)
    rb["text_features"], rb["visual_features"] = E.text_features(wtl, ids, mask), E.visual_features(wvl, frames)
    ro = O.forward_batch(fl, cl, rb)
    loss = F.cross_entropy(ro["logits"], batch["label"])
    loss.backward()
    assert abs(float(st.loss) - float(loss)) <= 1e-3
    grads = {"text." + k: v.grad for k, v in wtl.items() if v.grad is not None}
    grads.update({"vis." + k: v.grad for k, v in wvl.items() if v.grad is not None})
    enc_norm = sum(float(g.double().pow(2).sum()) for g in grads.values()) ** 0.5
    head_norm = sum(float(v.grad.double().pow(2).sum()) for d in (fl, cl) for v in d.values() if v.requires_grad and v.grad is not None) ** 0.5
    total = (enc_norm ** 2 + head_norm ** 2) ** 0.5
    print(f"loss {float(st.loss):.6f} (oracle {float(loss):.6f}); grad norm {float(st.grad_norm):.5f} (oracle {total:.5f}: head {head_norm:.5f}, encoders {enc_norm:.5f})")
    assert abs(float(st.grad_norm) - total) <= 2e-2 * total
    _compare(tr.arena, grads, 2.5e-2, "trainer step, encoder gradients (2 + 2 layers behind the fp32 head)")
    # every parameter moved against its gradient by about lr (AdamW's first step): compare the signs where the gradient is clear
    moved = (tr.arena.data - before).cpu()
    for k in ("text.encoder.layer.1.output.dense.weight", "vis.vision_model.encoder.layers.0.mlp.fc1.weight", "text.embeddings.position_embeddings.weight"):
        o, shape = tr.arena.offsets[k]
        n = 1
        for d in shape:
            n *= d
        step, gk = moved[o:o + n], grads[k].flatten()
        clear = gk.abs() > 0.1 * gk.abs().max()
        assert clear.sum() > 0 and (torch.sign(step[clear]) == -torch.sign(gk[clear])).float().mean().item() >= 0.999, k
        assert (step.abs().max().item() - cfg.lr) <= 0.05 * cfg.lr
    # the frozen fast path sees the updated masters afterwards (evaluation after training steps)
    f_train = tr.text_bp.forward_train(ids, mask).clone()
    tr._load_batch(tr.head.bufs(B, False), gb, "val")
    torch.cuda.synchronize()
    assert (tr.head.bufs(B, False)["text"] - f_train).abs().max().item() <= 2e-3


def test_best_checkpoint_carries_the_trained_encoders(tmp_path):
    """with train_encoders the best-epoch head is only meaningful with that epoch's encoders.  The checkpoint written
    at the best epoch holds both; after further steps have moved head AND encoders, _load_checkpoint() (what test() runs first)
    brings back exactly the logits of the saved state -- through the frozen fast path, whose packed operands must be rebuilt."""
    from oracle import encoders_ref as E
    from oracle import tier_a as O
    from ultrafnd_git_amd.dp import save_checkpoint
    from ultrafnd_git_amd.encoders import BertTextEncoder, ClipVisualEncoder
    from ultrafnd_git_amd.trainer import ForensicTrainer, TrainConfig, synthetic_cache
    B, Lq = 4, 64
    tenc, venc = BertTextEncoder(layers=1, vocab_size=300), ClipVisualEncoder(layers=1)
    tenc.load_state_dict(E.seeded_weights(E.bert_shapes(layers=1, vocab=300), 21))
    venc.load_state_dict(E.seeded_weights(E.vit_shapes(layers=1), 22))
    tenc, venc = tenc.to(DEV), venc.to(DEV)
    ids, mask = E.synthetic_tokens(23, B, Lq, vocab=300, min_len=8)
    frames = E.synthetic_frames(24, B, 1)
    cfg = TrainConfig(data_root="", ocr_phrase_pkl=None, out_dir=str(tmp_path), batch_size=B, device=DEV, use_graph=False, encode_inline=True,
                      train_encoders=True, lr=5e-3)
    tr = ForensicTrainer(cfg, cache=synthetic_cache(16, seed=1), text_encoder=tenc, visual_encoder=venc)
    gb = {k: v.to(DEV) for k, v in O.seeded_batch(25, B).items()}
    gb.update({"input_ids": ids.to(DEV), "attention_mask": mask.to(torch.int32).to(DEV), "frames": frames.to(DEV)})

    def logits():
        tr.fusion.eval(); tr.clf.eval()
        out = tr._forward_batch(gb, "val")      # the frozen fast path (packed operands), as validation / test() run it
        torch.cuda.synchronize()
        tr.fusion.train(); tr.clf.train()
        return out["logits"].float().clone()

    tr.fusion.train(); tr.clf.train()
    for _ in range(2):
        tr.train_step(gb)
    want = logits()
    enc_before = tr.arena.view("text.encoder.layer.0.output.dense.weight").clone()
    save_checkpoint(tr._checkpoint_state(), tr.ckpt_path, tr.comm)
    ck = torch.load(tr.ckpt_path, map_location="cpu", weights_only=True)
    assert "text_encoder" in ck and "visual_encoder" in ck and set(ck["text_encoder"]) == set(tenc.state_dict())
    for _ in range(3):      # "later epochs": head and encoders both move
        tr.train_step(gb)
    assert (tr.arena.view("text.encoder.layer.0.output.dense.weight") - enc_before).abs().max().item() > 1e-4
    moved = logits()
    assert (moved - want).abs().max().item() > 1e-4
    assert tr._load_checkpoint()
    got = logits()
    assert torch.equal(tr.arena.view("text.encoder.layer.0.output.dense.weight"), enc_before)
    assert (got - want).abs().max().item() <= 1e-6, (got - want).abs().max().item()
