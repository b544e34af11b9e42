"""GPU: BASELINE.json's other configurations and the batch sizes the golden fixtures do not cover.
Tier A is compared with the oracle evaluated on the fly (it is pinned to the reference by the fixtures);
the encoder configs are checked through size-independent properties (row independence, unit norm) and
against the oracle on a few sampled rows."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"
FEATS = ("text_features", "audio_features", "visual_features", "temporal_features", "gnn_feat")


@pytest.mark.parametrize("B", [1, 33, 100, 130, 200, 256])
def test_tier_a_forward_backward_vs_oracle(B):
    """M-tiling paths of the fp32 MFMA kernels: B=1 (one row), 33 / 100 (ragged 32-row tiles), 256
    (config 3's global batch on one GPU: two row tiles per wave, 8 row groups); from 128 rows up the weight-gradient
    GEMM splits the batch rows over a workgroup's four waves and the gate / NODE parameter reductions are row-sliced:
    130 and 200 leave the last wave / slice a ragged remainder (10 and 32 rows)."""
    from oracle import tier_a as O
    from ultrafnd_git_amd.classifier import DeepTruthClassifier
    from ultrafnd_git_amd.fusion import CrossModalTransformer
    fus_sd, clf_sd = O.seeded_params(77)
    batch = O.seeded_batch(B + 5, B)
    fusion, clf = CrossModalTransformer(), DeepTruthClassifier()
    fusion.load_state_dict(fus_sd); clf.load_state_dict(clf_sd)
    fusion, clf = fusion.to(DEV).train(), clf.to(DEV).train()
    fusion.dropout = clf.dropout = clf.node_dropout = 0.0
    gb = {k: v.to(DEV) for k, v in batch.items()}
    fo = fusion({k: gb[k] for k in FEATS})
    co = clf(fo["fused"], gb["aux"])
    loss = F.cross_entropy(co["logits"], gb["label"])
    loss.backward()
    out, rloss, gf, gc = O.loss_and_grads(fus_sd, clf_sd, batch)
    assert (co["logits"].detach().cpu() - out["logits"].detach()).abs().max().item() <= 2e-5
    assert (fo["fused"].detach().cpu() - out["fused"].detach()).abs().max().item() <= 2e-5
    assert abs(loss.item() - float(rloss)) <= 2e-5
    worst = 0.0
    for name, mod, ref in (("fusion", fusion, gf), ("clf", clf, gc)):
        for k, p in mod.named_parameters():
            if ref[k] is None:
                assert p.grad is None, k
                continue
            scale = max(ref[k].abs().max().item(), ref[k].norm().item() / max(1.0, ref[k].numel() ** 0.5), 1e-9)
            err = (p.grad.cpu() - ref[k]).abs().max().item() / scale
            worst = max(worst, err)
            assert err <= 5e-4, (name, k, err)
    print(f"B={B}: worst relative gradient error {worst:.2e}")


# three criteria (max-abs, relative L2, 1 - cos), bounds = 2 x measured on MI355X (these two were 4e-3 max-abs only)
CFG4_TEXT_BOUNDS = (7.5e-4, 5.7e-3, 8.0e-6)       # measured 3.7e-4 / 2.8e-3 / 4.0e-6
CFG5_VIS_BOUNDS = (1.3e-3, 7.0e-3, 1.25e-5)       # measured 6.5e-4 / 3.5e-3 / 6.1e-6


def _encoders(layers=2, vocab=1000):
    from oracle import encoders_ref as E
    from ultrafnd_git_amd.encoders import BertTextEncoder, ClipVisualEncoder
    wt = E.seeded_weights(E.bert_shapes(layers=layers, vocab=vocab), 61)
    wv = E.seeded_weights(E.vit_shapes(layers=layers), 62)
    tenc, venc = BertTextEncoder(layers=layers, vocab_size=vocab), ClipVisualEncoder(layers=layers)
    tenc.load_state_dict(wt); venc.load_state_dict(wv)
    return wt, wv, tenc.to(DEV), venc.to(DEV)


def test_config4_text_only_seq512_batch128():
    """BASELINE config 4: text-only ablation, L=512, B=128 (65,536 tokens through the encoder), visual and
    temporal inputs zero.  Properties: unit-norm features, row independence (a sample's feature does not
    depend on its batch), sampled rows equal the oracle, the head accepts the zero branches."""
    from oracle import encoders_ref as E
    from ultrafnd_git_amd.classifier import DeepTruthClassifier
    from ultrafnd_git_amd.fusion import CrossModalTransformer
    wt, _, tenc, _ = _encoders()
    B, Lq = 128, 512
    ids, mask = E.synthetic_tokens(46, B, Lq, vocab=1000)
    feat = tenc(ids, mask).clone()
    assert torch.isfinite(feat).all()
    assert (feat.norm(dim=1) - 1.0).abs().max().item() <= 1e-4
    rows = [0, 57, 127]
    sub = tenc(ids[rows], mask[rows]).clone()
    assert torch.equal(sub, feat[rows]), "row independence must be exact: same per-row arithmetic order in any batch"
    ref = E.text_features(wt, ids[rows], mask[rows])
    from tests.helpers import assert_features_close
    assert_features_close(sub.cpu(), ref, *CFG4_TEXT_BOUNDS, what="configs[3] geometry (2-layer encoder, L = 512, B = 128), sampled rows")
    from oracle import tier_a as O
    fus_sd, clf_sd = O.seeded_params(78)
    fusion, clf = CrossModalTransformer(), DeepTruthClassifier()
    fusion.load_state_dict(fus_sd); clf.load_state_dict(clf_sd)
    fusion, clf = fusion.to(DEV).eval(), clf.to(DEV).eval()
    g = torch.Generator().manual_seed(1)
    audio, gnn, aux = torch.randn(B, 128, generator=g), torch.randn(B, 128, generator=g), torch.rand(B, 2, generator=g)
    with torch.no_grad():
        fo = fusion({"text_features": feat, "audio_features": audio.to(DEV), "visual_features": torch.zeros(B, 512, device=DEV),
                     "temporal_features": torch.zeros(B, 256, device=DEV), "gnn_feat": gnn.to(DEV)})
        co = clf(fo["fused"], aux.to(DEV))
    assert torch.isfinite(co["logits"]).all()
    # the "visual branch off" ablation (zero visual / temporal inputs, SURVEY 8d cfg 4) vs the oracle on sampled rows
    ob = {"text_features": ref, "audio_features": audio[rows], "visual_features": torch.zeros(3, 512),
          "temporal_features": torch.zeros(3, 256), "gnn_feat": gnn[rows], "aux": aux[rows], "label": torch.zeros(3, dtype=torch.long)}
    oo = O.forward_batch(fus_sd, clf_sd, ob)
    assert (co["logits"][rows].cpu() - oo["logits"]).abs().max().item() <= 1e-3
    assert (fo["forensic"]["temporal_delay"][rows].cpu() - oo["forensic"]["temporal_delay"]).abs().max().item() <= 1e-3


def test_config5_multiframe_8x224():
    """BASELINE config 5: 8 frames x 224^2 per sample, L=128, 8 samples per GPU."""
    from oracle import encoders_ref as E
    _, wv, _, venc = _encoders()
    B, Fr = 8, 8
    frames = E.synthetic_frames(47, B, Fr)
    feat = venc(frames).clone()
    assert (feat.norm(dim=1) - 1.0).abs().max().item() <= 1e-4
    ref = E.visual_features(wv, frames[:2])
    from tests.helpers import assert_features_close
    assert_features_close(feat[:2].cpu(), ref, *CFG5_VIS_BOUNDS, what="configs[4] shard (2-layer encoder, 8 frames, B = 8), samples 0, 1")
    # frame order does not matter (mean over frames); a single frame repeated equals that frame's feature
    perm = frames[:, torch.randperm(Fr, generator=torch.Generator().manual_seed(0))]
    assert (venc(perm) - feat).abs().max().item() <= 2e-6
    rep = frames[:, :1].expand(B, Fr, 3, 224, 224).contiguous()
    assert (venc(rep) - venc(frames[:, :1])).abs().max().item() <= 2e-6


def test_adamw_is_linear_in_nothing_but_matches_reference_formula_at_scale():
    """Optimizer arena at full size: one clip+AdamW step on random grads == the hand formula (oracle.AdamWState)."""
    from oracle import tier_a as O
    from ultrafnd_git_amd.arena import FlatArena
    from ultrafnd_git_amd.optim import FusedAdamW
    n = 12_746_240
    arena = FlatArena([[("w", (n,))]], [], torch.device(DEV))
    g = torch.Generator().manual_seed(5)
    p0 = torch.randn(n, generator=g) * 0.05
    g0 = torch.randn(n, generator=g) * 0.01
    arena.view("w").copy_(p0.to(DEV))
    opt = FusedAdamW(arena, lr=2e-4, weight_decay=1e-4, max_norm=5.0)
    ref_p = {"w": p0.clone()}
    st = O.AdamWState()
    for _ in range(2):
        arena.grad.copy_(g0.to(DEV))
        opt.step()
        grads = {"w": g0.clone()}
        O.clip_grads_(grads, 5.0)
        st.step(ref_p, grads)
    s = opt.state.read()
    assert abs(s.grad_norm - g0.double().norm().item()) <= 1e-4 * s.grad_norm and s.clip_coef < 1.0
    assert (arena.view("w").cpu() - ref_p["w"]).abs().max().item() <= 2e-7
