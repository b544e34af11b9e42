#!/usr/bin/env python3
"""Mint the golden fixtures under tests/golden/ from the REAL reference.

Run in the build container only (needs /root/reference, which never travels):

    python tests/golden/make_golden.py            # all three parts

What it does
  tier_a   imports the reference's CrossModalTransformer / DeepTruthClassifier with
           `transformers` masked (the reference's own "transformers is
           optional" branch, so no name-based from_pretrained loader is reached),
           loads oracle.tier_a.seeded_params(seed) into them, runs eval-mode forward,
           CE loss, backward, clip_grad_norm_(5.0) and torch.optim.AdamW for 3 steps
           (dropout p=0) at B in {2,4,32}; asserts the oracle restatement agrees
           element-for-element, and stores the reference's outputs (full for the small
           tensors, per-tensor digests for grads/params) in tier_a_B*.npz.
  metrics  runs the reference's forensic_metrics on known inputs -> metrics_kat.json.
  temporal runs the reference's TemporalSyncNet.align on seeded weights -> temporal.npz.
  temporal_seq runs the reference's TemporalSyncNet(use_tcn=True).forward (eval, and train with dropout p=0) on seeded
           weights for four geometries -> temporal_seq.npz; also delay_score / estimate_av_lag known answers.
  gnn_model runs the reference's GNNModel (forward + autograd backward) and, when importable, build_adj_from_ocr_sets
           of the integrated trainer variant on synthetic mini-batches -> gnn_model.npz.
  gcn      runs the reference's build_adj_from_ocr / SimpleGCN (forward, two Adam pre-training steps with
           dropout p=0) on synthetic phrase sets -> gcn.npz.
  tier_b   builds the locally installed third-party BertModel / CLIPVisionModelWithProjection
           from local configs (2 layers, small vocab; no from_pretrained), loads
           oracle.encoders_ref.seeded_weights, stores inputs + outputs in tier_b.npz and
           records one full-depth (12-layer) oracle-vs-third-party max-abs-err.

Fixtures are data only (inputs, expected outputs, digests); weights are regenerated
from the seed by oracle.* on both sides, with a checksum stored to catch drift.
"""
from __future__ import annotations

import json
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
REPO = HERE.parents[1]
REF = Path("/root/reference")
sys.path.insert(0, str(REPO))

PARAM_SEED = 1234
BATCH_SEEDS = {2: 7, 4: 11, 32: 13}


def digest(t: torch.Tensor) -> dict:
    f = t.detach().double().flatten()
    n = f.numel()
    stride = max(1, n // 16)
    return {"norm": np.float64(f.norm().item()), "sum": np.float64(f.sum().item()),
            "head": f[:16].float().numpy(), "strided": f[::stride][:16].float().numpy()}


def put_digest(store: dict, prefix: str, t: torch.Tensor):
    for k, v in digest(t).items():
        store[f"{prefix}/{k}"] = np.asarray(v)


# ---------------------------------------------------------------------------
def tier_a():
    sys.modules["transformers"] = None
    os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
    sys.dont_write_bytecode = True
    sys.path.insert(0, str(REF))
    os.chdir(REF)
    import torch.nn as nn
    import torch.nn.functional as F
    from src.models.fusion.cross_modal_transformer import CrossModalTransformer
    from src.models.fusion.deep_truth_classifier import DeepTruthClassifier
    from oracle import tier_a as O

    fus_sd, clf_sd = O.seeded_params(PARAM_SEED)
    checksum = float(sum(v.double().sum() for v in list(fus_sd.values()) + list(clf_sd.values())))

    for B, bseed in BATCH_SEEDS.items():
        torch.manual_seed(0)
        fusion = CrossModalTransformer("configs/model_configs/fusion.yaml").to("cpu")
        clf = DeepTruthClassifier("configs/model_configs/classifier.yaml").to("cpu")
        assert list(fusion.state_dict().keys()) == list(fus_sd.keys())
        assert list(clf.state_dict().keys()) == list(clf_sd.keys())
        fusion.load_state_dict(fus_sd)
        clf.load_state_dict(clf_sd)
        for m in list(fusion.modules()) + list(clf.modules()):
            if isinstance(m, nn.Dropout):
                m.p = 0.0
        batch = O.seeded_batch(bseed, B)
        feats = {k: batch[k] for k in ("text_features", "audio_features", "visual_features",
                                       "temporal_features", "gnn_feat")}
        store = {"param_seed": np.int64(PARAM_SEED), "batch_seed": np.int64(bseed),
                 "param_checksum": np.float64(checksum)}
        for k, v in batch.items():
            store[f"in/{k}"] = v.numpy()

        # ---- eval forward (reference) vs oracle
        fusion.eval(); clf.eval()
        with torch.no_grad():
            fo = fusion(feats)
            co = clf(fo["fused"], batch["aux"])
        oo = O.forward_batch(fus_sd, clf_sd, batch, train=False)
        for name, r, o in (("fused", fo["fused"], oo["fused"]), ("fusion_logits", fo["logits"], oo["fusion_logits"]),
                           ("logits", co["logits"], oo["logits"]), ("probs", co["probs"], oo["probs"])):
            err = (r - o).abs().max().item()
            assert err <= 1e-6, (name, err)
            store[f"out/{name}"] = r.numpy()
        for k in ("emotion_intensity", "semantic_conflict", "temporal_delay"):
            assert (fo["forensic"][k] - oo["forensic"][k]).abs().max().item() <= 1e-7
            store[f"out/forensic/{k}"] = fo["forensic"][k].numpy()
        store["out/temperature"] = co["temperature"].detach().numpy()

        # ---- 3 train steps (reference torch.optim.AdamW + clip_grad_norm_) vs oracle
        fusion.train(); clf.train()
        params = list(fusion.parameters()) + list(clf.parameters())
        optim = torch.optim.AdamW(params, lr=2e-4, weight_decay=1e-4)
        of = {k: v.clone() for k, v in fus_sd.items()}
        oc = {k: v.clone() for k, v in clf_sd.items()}
        ostate = O.AdamWState()
        for step in (1, 2, 3):
            fo = fusion(feats)
            co = clf(fo["fused"], batch["aux"])
            loss = F.cross_entropy(co["logits"], batch["label"])
            optim.zero_grad(set_to_none=True)
            loss.backward()
            if step == 1:
                grads_ref = {("fusion." + k): (p.grad.clone() if p.grad is not None else None)
                             for k, p in fusion.named_parameters()}
                grads_ref.update({("clf." + k): (p.grad.clone() if p.grad is not None else None)
                                  for k, p in clf.named_parameters()})
            total = nn.utils.clip_grad_norm_(params, max_norm=5.0)
            optim.step()
            # oracle step
            if step == 1:
                _, oloss, gf, gc = O.loss_and_grads(of, oc, batch, train=False)
                for k, g in {**{"fusion." + k: g for k, g in gf.items()},
                             **{"clf." + k: g for k, g in gc.items()}}.items():
                    r = grads_ref[k]
                    assert (g is None) == (r is None), k
                    if g is not None:
                        err = (g - r).abs().max().item()
                        assert err <= 2e-6 * max(1.0, r.abs().max().item()), (k, err)
            oout, oloss, ototal = O.train_step(of, oc, batch, ostate, grad_clip=5.0, train=False)
            assert abs(oloss - loss.item()) <= 1e-6, (oloss, loss.item())
            assert abs(ototal - total.item()) <= 1e-5 * max(1.0, total.item())
            store[f"step{step}/loss"] = np.float64(loss.item())
            store[f"step{step}/grad_norm"] = np.float64(total.item())
            store[f"step{step}/logits"] = co["logits"].detach().numpy()
            if step == 1:
                nograd = []
                for k, g in grads_ref.items():
                    if g is None:
                        nograd.append(k)
                    else:
                        put_digest(store, f"grad/{k}", g)
                store["nograd_keys"] = np.array(json.dumps(nograd))
            if step in (1, 3):
                worst = 0.0
                for k, p in list(("fusion." + k, p) for k, p in fusion.named_parameters()) + \
                        list(("clf." + k, p) for k, p in clf.named_parameters()):
                    o = (of if k.startswith("fusion.") else oc)[k.split(".", 1)[1]]
                    worst = max(worst, (p.detach() - o).abs().max().item())
                    put_digest(store, f"param_step{step}/{k}", p.detach())
                assert worst <= 2e-6, worst
        np.savez_compressed(HERE / f"tier_a_B{B}.npz", **store)
        print(f"tier_a B={B}: ok  loss={store['step1/loss']:.6f} gnorm={store['step1/grad_norm']:.6f}")


# ---------------------------------------------------------------------------
def tier_a_nognn():
    """fusion.yaml `use_gnn: false` (15H concat, no gnn_proj; cross_modal_transformer.py:88,101-102,114-120): the reference's own
    module built from such a YAML, eval forward + one backward, against the oracle -> tier_a_nognn_B4.npz."""
    sys.modules["transformers"] = None
    os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
    sys.dont_write_bytecode = True
    sys.path.insert(0, str(REF))
    yaml_path = str(REPO / "configs" / "model_configs" / "fusion_nognn.yaml")
    os.chdir(REF)
    import torch.nn as nn
    import torch.nn.functional as F
    from src.models.fusion.cross_modal_transformer import CrossModalTransformer
    from src.models.fusion.deep_truth_classifier import DeepTruthClassifier
    from oracle import tier_a as O
    B, bseed = 4, 29
    fus_sd, clf_sd = O.seeded_params(PARAM_SEED + 1, use_gnn=False)
    torch.manual_seed(0)
    fusion = CrossModalTransformer(yaml_path).to("cpu")
    clf = DeepTruthClassifier("configs/model_configs/classifier.yaml").to("cpu")
    assert not fusion.use_gnn and fusion.fused_dim == 15 * 512 and not hasattr(fusion, "gnn_proj")
    assert list(fusion.state_dict().keys()) == list(fus_sd.keys())
    fusion.load_state_dict(fus_sd); clf.load_state_dict(clf_sd)
    for m in list(fusion.modules()) + list(clf.modules()):
        if isinstance(m, nn.Dropout):
            m.p = 0.0
    batch = O.seeded_batch(bseed, B)
    feats = {k: batch[k] for k in ("text_features", "audio_features", "visual_features", "temporal_features", "gnn_feat")}   # gnn_feat given: must be ignored
    store = {"param_seed": np.int64(PARAM_SEED + 1), "batch_seed": np.int64(bseed),
             "param_checksum": np.float64(sum(v.double().sum() for v in list(fus_sd.values()) + list(clf_sd.values())))}
    fusion.eval(); clf.eval()
    with torch.no_grad():
        fo = fusion(feats)
        co = clf(fo["fused"], batch["aux"])
    oo = O.forward_batch(fus_sd, clf_sd, batch, train=False)
    for name, r, o in (("fused", fo["fused"], oo["fused"]), ("logits", co["logits"], oo["logits"]), ("probs", co["probs"], oo["probs"])):
        assert (r - o).abs().max().item() <= 1e-6, name
        store[f"out/{name}"] = r.numpy()
    fusion.train(); clf.train()
    fo = fusion(feats)
    co = clf(fo["fused"], batch["aux"])
    loss = F.cross_entropy(co["logits"], batch["label"])
    loss.backward()
    _, oloss, gf, gc = O.loss_and_grads(fus_sd, clf_sd, batch, train=False)
    assert abs(float(oloss) - loss.item()) <= 1e-6
    nograd = []
    for pre, mod, og in (("fusion.", fusion, gf), ("clf.", clf, gc)):
        for k, p in mod.named_parameters():
            g = og[k]
            assert (g is None) == (p.grad is None), k
            if p.grad is None:
                nograd.append(pre + k)
            else:
                assert (g - p.grad).abs().max().item() <= 2e-6 * max(1.0, p.grad.abs().max().item()), k
                put_digest(store, f"grad/{pre}{k}", p.grad)
    store["nograd_keys"] = np.array(json.dumps(nograd))
    store["step1/loss"] = np.float64(loss.item())
    np.savez_compressed(HERE / "tier_a_nognn_B4.npz", **store)
    print(f"tier_a_nognn B={B}: ok  loss={loss.item():.6f}  ({len(nograd)} tensors without gradient)")


def metrics():
    sys.modules["transformers"] = None
    sys.dont_write_bytecode = True
    sys.path.insert(0, str(REF))
    from src.training.metrics import forensic_metrics as RM
    from oracle import metrics_ref as OM

    rng = np.random.default_rng(5)
    cases = []

    def add(name, y, p, forensic=None, include_cm=False):
        r = RM.aggregate_epoch_metrics(np.array(y), np.array(p), forensic=forensic and
                                       {k: np.array(v) for k, v in forensic.items()},
                                       threshold=0.5, include_cm=include_cm)
        o = OM.aggregate_epoch_metrics(np.array(y), np.array(p), forensic=forensic and
                                       {k: np.array(v) for k, v in forensic.items()},
                                       threshold=0.5, include_cm=include_cm)
        assert set(r) == set(o), (name, set(r) ^ set(o))
        for k in r:
            assert abs(r[k] - o[k]) <= 1e-12, (name, k, r[k], o[k])
        cases.append({"name": name, "y": list(map(int, y)), "p": list(map(float, p)),
                      "forensic": forensic, "include_cm": include_cm,
                      "expected": {k: float(v) for k, v in r.items()}})

    add("sanity_check.py:35-36", [0, 1, 1, 0], [0.1, 0.9, 0.8, 0.2])
    add("survey_8c", [0, 1, 1, 0, 1, 0], [.2, .7, .4, .6, .9, .1],
        {"semantic_conflict": [.1, .5, .9, .3, .2, .4], "temporal_delay": [.2, .4, .6, .8, 1.0, 0.0],
         "emotion_intensity": [.1, .2, .3, .4, .5, .6]}, include_cm=True)
    add("single_class", [1, 1, 1], [0.2, 0.7, 0.9],
        {"semantic_conflict": [1.0, 1.0, 0.9], "temporal_delay": [1.0, 1.0, 1.0], "emotion_intensity": [0, 0, 0]})
    add("ties", [0, 1, 0, 1, 1, 0, 0, 1], [0.5, 0.5, 0.5, 0.7, 0.2, 0.2, 0.9, 0.9])
    add("all_negative_pred", [0, 1, 0, 1], [0.1, 0.2, 0.3, 0.4],
        {"semantic_conflict": [0, 0, 0, 0], "temporal_delay": [0, 0, 0, 0], "emotion_intensity": [1, 1, 1, 1]})
    y = rng.integers(0, 2, 257)
    p = np.round(rng.random(257), 2)          # many ties
    add("random_257", y.tolist(), p.tolist(),
        {"semantic_conflict": rng.random(257).tolist(), "temporal_delay": rng.random(257).tolist(),
         "emotion_intensity": rng.random(257).tolist()}, include_cm=True)
    # two-column score inputs (_to_prob_1 branches, forensic_metrics.py:35-56)
    two = []
    for name, s in (("probs2", [[0.8, 0.2], [0.3, 0.7], [0.4, 0.6], [0.9, 0.1]]),
                    ("logits2", [[2.0, -1.0], [0.1, 0.3], [-3.0, 1.0], [0.5, 0.2]])):
        r = RM.compute_classification_metrics(np.array([0, 1, 1, 0]), np.array(s))
        o = OM.compute_classification_metrics(np.array([0, 1, 1, 0]), np.array(s))
        assert all(abs(r[k] - o[k]) <= 1e-12 for k in r)
        two.append({"name": name, "y": [0, 1, 1, 0], "score": s, "expected": {k: float(v) for k, v in r.items()}})
    (HERE / "metrics_kat.json").write_text(json.dumps({"aggregate": cases, "two_column": two}, indent=1))
    print(f"metrics: {len(cases)} + {len(two)} KATs ok")


# ---------------------------------------------------------------------------
def tier_b():
    from transformers import BertConfig, BertModel, CLIPVisionConfig, CLIPVisionModelWithProjection
    from oracle import encoders_ref as E

    store = {}
    torch.manual_seed(0)
    # ---- text: 2-layer BERT, small vocab (fixture-sized); production geometry otherwise
    VOC = 1000
    for tag, layers, B, L, vocab, seed in (("bert2_L128", 2, 4, 128, VOC, 21), ("bert2_L512", 2, 2, 512, VOC, 22),
                                           ("bert2_L40", 2, 3, 40, VOC, 23)):
        w = E.seeded_weights(E.bert_shapes(layers=layers, vocab=vocab), seed)
        m = BertModel(BertConfig(num_hidden_layers=layers, vocab_size=vocab), add_pooling_layer=False).eval()
        missing = m.load_state_dict(w, strict=True)
        ids, mask = E.synthetic_tokens(seed + 100, B, L, vocab=vocab, min_len=min(16, L))
        with torch.no_grad():
            ref = m(input_ids=ids, attention_mask=mask).last_hidden_state
        ora = E.bert_last_hidden_state(w, ids, mask)
        valid = mask.bool()
        err = (ref - ora)[valid].abs().max().item()
        assert err <= 2e-5, (tag, err)
        feat = E.masked_meanpool_l2(ref, mask)          # text_blocks.py:82-101 applied to the third-party output
        store[f"{tag}/ids"] = ids.numpy().astype(np.int64)
        store[f"{tag}/mask"] = mask.numpy().astype(np.int64)
        store[f"{tag}/features"] = feat.numpy()
        store[f"{tag}/hidden_row0"] = ref[0].numpy()[: int(mask[0].sum())]
        store[f"{tag}/meta"] = np.array(json.dumps({"layers": layers, "vocab": vocab, "weight_seed": seed,
                                                    "checksum": float(sum(v.double().sum() for v in w.values()))}))
        print(f"tier_b {tag}: oracle-vs-third-party hidden max-abs-err {err:.2e}")

    # ---- visual: 2-layer CLIP ViT-B/32
    for tag, layers, B, Fr, seed in (("vit2_F1", 2, 3, 1, 31), ("vit2_F4", 2, 2, 4, 32)):
        w = E.seeded_weights(E.vit_shapes(layers=layers), seed)
        m = CLIPVisionModelWithProjection(CLIPVisionConfig(num_hidden_layers=layers)).eval()
        m.load_state_dict(w, strict=True)
        frames = E.synthetic_frames(seed + 100, B, Fr)
        with torch.no_grad():
            out = m(pixel_values=frames.reshape(B * Fr, 3, 224, 224))
        pooled_o = E.vit_pooled(w, frames.reshape(B * Fr, 3, 224, 224))
        e_ref = out.image_embeds
        e_ora = torch.nn.functional.linear(pooled_o, w["visual_projection.weight"])
        err = (e_ref - e_ora).abs().max().item()
        assert err <= 2e-5, (tag, err)
        feat_ref = e_ref / (e_ref.norm(dim=-1, keepdim=True) + 1e-9)
        feat_ref = feat_ref.view(B, Fr, -1)
        feat_ref = feat_ref[:, 0] if Fr == 1 else E.field_mean_l2(feat_ref)
        assert (feat_ref - E.visual_features(w, frames)).abs().max().item() <= 1e-5
        # frames are big: store the generator seed, not the pixels
        store[f"{tag}/features"] = feat_ref.numpy()
        store[f"{tag}/image_embeds"] = e_ref.numpy()
        store[f"{tag}/meta"] = np.array(json.dumps({"layers": layers, "weight_seed": seed, "frame_seed": seed + 100,
                                                    "B": B, "F": Fr, "frames_checksum": float(frames.double().sum()),
                                                    "checksum": float(sum(v.double().sum() for v in w.values()))}))
        print(f"tier_b {tag}: oracle-vs-third-party embeds max-abs-err {err:.2e}")

    # ---- one full-depth check, scalar result recorded
    w = E.seeded_weights(E.bert_shapes(layers=12, vocab=VOC), 41)
    m = BertModel(BertConfig(vocab_size=VOC), add_pooling_layer=False).eval()
    m.load_state_dict(w, strict=True)
    ids, mask = E.synthetic_tokens(141, 2, 128, vocab=VOC)
    with torch.no_grad():
        ref = E.masked_meanpool_l2(m(input_ids=ids, attention_mask=mask).last_hidden_state, mask)
    e12 = (ref - E.text_features(w, ids, mask)).abs().max().item()
    w = E.seeded_weights(E.vit_shapes(layers=12), 42)
    m = CLIPVisionModelWithProjection(CLIPVisionConfig()).eval()
    m.load_state_dict(w, strict=True)
    fr = E.synthetic_frames(142, 2, 1)
    with torch.no_grad():
        e = m(pixel_values=fr[:, 0]).image_embeds
    v12 = (e / (e.norm(dim=-1, keepdim=True) + 1e-9) - E.visual_features(w, fr)).abs().max().item()
    store["full_depth/bert12_feature_err"] = np.float64(e12)
    store["full_depth/vit12_feature_err"] = np.float64(v12)
    assert e12 <= 1e-5 and v12 <= 1e-5, (e12, v12)
    print(f"tier_b full depth: bert12 feature err {e12:.2e}, vit12 feature err {v12:.2e}")
    np.savez_compressed(HERE / "tier_b.npz", **store)


def tier_b_grads():
    """Gradients of the encoders (Tier-B backward).  The reference never trains them; the oracle's autograd gradients
    (oracle.encoders_ref.*_feature_grads) are checked here against the installed third-party classes' OWN autograd on the same
    seeded weights and probe loss, and digests of every gradient tensor are stored (tier_b_grads.npz)."""
    from transformers import BertConfig, BertModel, CLIPVisionConfig, CLIPVisionModelWithProjection
    from oracle import encoders_ref as E

    store = {}
    VOC, LOSS_SEED = 1000, 77
    for tag, layers, B, L, seed in (("bert2_L64", 2, 3, 64, 24), ("bert2_L128", 2, 2, 128, 25)):
        w = E.seeded_weights(E.bert_shapes(layers=layers, vocab=VOC), seed)
        m = BertModel(BertConfig(num_hidden_layers=layers, vocab_size=VOC), add_pooling_layer=False).eval()      # eval: no dropout
        m.load_state_dict(w, strict=True)
        ids, mask = E.synthetic_tokens(seed + 100, B, L, vocab=VOC, min_len=8)
        feat = E.masked_meanpool_l2(m(input_ids=ids, attention_mask=mask).last_hidden_state, mask)
        E.probe_loss(feat, LOSS_SEED).backward()
        ref = {k: (p.grad if p.grad is not None else torch.zeros_like(p)) for k, p in m.named_parameters()}
        _, ora = E.text_feature_grads(w, ids, mask, LOSS_SEED)
        worst = 0.0
        for k, g in ref.items():
            scale = max(g.abs().max().item(), 1e-3 * max(x.abs().max().item() for x in ref.values()))
            worst = max(worst, (g - ora[k]).abs().max().item() / scale)
            put_digest(store, f"{tag}/grad/{k}", ora[k])
        assert worst <= 2e-4, (tag, worst)
        store[f"{tag}/ids"], store[f"{tag}/mask"] = ids.numpy().astype(np.int64), mask.numpy().astype(np.int64)
        store[f"{tag}/meta"] = np.array(json.dumps({"layers": layers, "vocab": VOC, "weight_seed": seed, "loss_seed": LOSS_SEED,
                                                    "keys": list(ref.keys())}))
        print(f"tier_b_grads {tag}: oracle-vs-third-party autograd, worst relative gradient error {worst:.2e} over {len(ref)} tensors")
    for tag, layers, B, Fr, seed in (("vit2_F1", 2, 3, 1, 33), ("vit2_F2", 2, 2, 2, 34)):
        w = E.seeded_weights(E.vit_shapes(layers=layers), seed)
        m = CLIPVisionModelWithProjection(CLIPVisionConfig(num_hidden_layers=layers)).eval()
        m.load_state_dict(w, strict=True)
        frames = E.synthetic_frames(seed + 100, B, Fr)
        e = m(pixel_values=frames.reshape(B * Fr, 3, 224, 224)).image_embeds
        u = (e / (e.norm(dim=-1, keepdim=True) + 1e-9)).view(B, Fr, -1)
        feat = u[:, 0] if Fr == 1 else E.field_mean_l2(u)
        E.probe_loss(feat, LOSS_SEED).backward()
        ref = {k: (p.grad if p.grad is not None else torch.zeros_like(p)) for k, p in m.named_parameters()}
        _, ora = E.visual_feature_grads(w, frames, LOSS_SEED)
        worst = 0.0
        for k, g in ref.items():
            if k not in ora:
                continue                      # (position_ids-like buffers are not parameters of the restatement)
            scale = max(g.abs().max().item(), 1e-3 * max(x.abs().max().item() for x in ref.values()))
            worst = max(worst, (g - ora[k]).abs().max().item() / scale)
            put_digest(store, f"{tag}/grad/{k}", ora[k])
        assert worst <= 2e-4, (tag, worst)
        store[f"{tag}/meta"] = np.array(json.dumps({"layers": layers, "weight_seed": seed, "frame_seed": seed + 100, "B": B, "F": Fr,
                                                    "loss_seed": LOSS_SEED, "keys": [k for k in ref if k in ora]}))
        print(f"tier_b_grads {tag}: oracle-vs-third-party autograd, worst relative gradient error {worst:.2e} over {len(ref)} tensors")
    np.savez_compressed(HERE / "tier_b_grads.npz", **store)


def init_parity():
    """Per-tensor checksums of the reference modules' INITIAL weights under torch.manual_seed(123): the
    mirror constructs its parameters in the same order, so the same seed must give the same weights."""
    sys.modules["transformers"] = None
    sys.dont_write_bytecode = True
    sys.path.insert(0, str(REF))
    os.chdir(REF)
    from src.models.fusion.cross_modal_transformer import CrossModalTransformer
    from src.models.fusion.deep_truth_classifier import DeepTruthClassifier
    from src.training.forensic_trainer import TrainConfig, SimpleGCN
    import dataclasses
    torch.manual_seed(321)
    gcn = SimpleGCN(in_dim=416, hid=256, out_dim=128, dropout=0.2)       # as _build_gnn constructs it (:203)
    gcn_after = float(torch.rand(1))                                      # the RNG position right after the construction
    torch.manual_seed(123)
    fusion = CrossModalTransformer("configs/model_configs/fusion.yaml")
    clf = DeepTruthClassifier("configs/model_configs/classifier.yaml")
    out = {"fusion": {k: [list(v.shape), float(v.double().sum()), float(v.double().abs().sum())] for k, v in fusion.state_dict().items()},
           "clf": {k: [list(v.shape), float(v.double().sum()), float(v.double().abs().sum())] for k, v in clf.state_dict().items()},
           "gcn": {k: [list(v.shape), float(v.double().sum()), float(v.double().abs().sum())] for k, v in gcn.state_dict().items()},
           "gcn_rng_after": gcn_after,
           "train_config_fields": [[f.name, repr(f.default) if f.default is not dataclasses.MISSING else None]
                                   for f in dataclasses.fields(TrainConfig)]}
    (HERE / "init_parity.json").write_text(json.dumps(out, indent=0))
    print("init_parity:", len(out["fusion"]), "+", len(out["clf"]), "tensors,", len(out["train_config_fields"]), "TrainConfig fields")


def temporal():
    """TemporalSyncNet.align (src/core_blocks/temporal_blocks.py:102-140) with seeded weights."""
    sys.dont_write_bytecode = True
    sys.path.insert(0, str(REF))
    from src.core_blocks.temporal_blocks import TemporalSyncNet
    from oracle import temporal_ref as T
    w = T.seeded_weights(51)
    net = TemporalSyncNet(in_dim=768, out_dim=256).to("cpu").eval()
    assert list(net.state_dict().keys()) == list(w.keys())
    net.load_state_dict(w)
    g = torch.Generator().manual_seed(52)
    t = torch.randn(6, 768, generator=g)
    t = t / t.norm(dim=1, keepdim=True)
    v = torch.randn(6, 512, generator=g)
    v = v / v.norm(dim=1, keepdim=True)
    v[5] = 0                                             # zero visual vector: cosine's eps path
    ref = torch.stack([torch.from_numpy(net.align(t[i].numpy(), v[i].numpy())) for i in range(6)])
    ref_self = torch.from_numpy(net.align(t[0].numpy(), t[0].numpy()))      # fakesv_dataset.py:180 usage
    ora = T.align(w, t, v)
    assert (ref - ora).abs().max().item() <= 1e-6
    assert (ref_self - T.align(w, t[:1], t[:1])[0]).abs().max().item() <= 1e-6
    np.savez_compressed(HERE / "temporal.npz", weight_seed=np.int64(51), t=t.numpy(), v=v.numpy(), out=ref.numpy(),
                        out_self=ref_self.numpy(), checksum=np.float64(sum(x.double().sum() for x in w.values())))
    print("temporal: oracle-vs-reference max-abs-err", (ref - ora).abs().max().item())


SEQ_CASES = [  # name, in_dim, out_dim, hid, layers, k, B, T, Dt
    ("a", 96, 64, 64, 2, 3, 3, 10, 64),       # widths differ at block 0 (no residual there), residual at block 1
    ("b", 64, 32, 64, 3, 3, 2, 9, 32),        # in_dim == hid: residual at every block; dilations 1, 2, 4 > T/2
    ("c", 48, 32, 32, 2, 4, 2, 7, 20),        # even kernel: 'same' pads asymmetrically
    ("d", 40, 32, 32, 2, 5, 4, 1, 24),        # T = 1 clips; kernel 5
]


def temporal_seq():
    """TemporalSyncNet.forward, the sequence path (src/core_blocks/temporal_blocks.py:16-43,141-157), and the two
    host-side delay estimators (:162-226)."""
    sys.dont_write_bytecode = True
    sys.path.insert(0, str(REF))
    from src.core_blocks.temporal_blocks import TemporalSyncNet
    from oracle import temporal_ref as T
    store = {}
    worst = 0.0
    for n, (name, in_dim, out_dim, hid, layers, k, B, Tn, Dt) in enumerate(SEQ_CASES):
        w = T.seq_seeded_weights(70 + n, in_dim, out_dim, hid, layers, k)
        net = TemporalSyncNet(in_dim=in_dim, out_dim=out_dim, use_tcn=True, tcn_hid=hid, tcn_layers=layers, tcn_kernel=k, dropout=0.0).to("cpu")
        assert list(net.state_dict().keys()) == list(w.keys()), (list(net.state_dict().keys()), list(w.keys()))
        net.load_state_dict(w)
        g = torch.Generator().manual_seed(80 + n)
        ts, vs = torch.randn(B, Tn, Dt, generator=g), torch.randn(B, Tn, in_dim - Dt, generator=g)
        net.eval()
        with torch.no_grad():
            out_eval = net(ts, vs)
        net.train()
        with torch.no_grad():
            out_train = net(ts, vs)
        sd = net.state_dict()
        o_eval, _ = T.sequence_forward(w, ts, vs, layers, k, train=False)
        o_train, stats = T.sequence_forward(w, ts, vs, layers, k, train=True)
        errs = [(out_eval - o_eval).abs().max().item(), (out_train - o_train).abs().max().item()]
        errs += [(sd[kk] - vv).abs().max().item() for kk, vv in stats.items()]
        assert max(errs) <= 2e-5, (name, errs)
        worst = max(worst, max(errs))
        store.update({f"{name}/text_seq": ts.numpy(), f"{name}/vis_seq": vs.numpy(), f"{name}/out_eval": out_eval.numpy(),
                      f"{name}/out_train": out_train.numpy(), f"{name}/weight_seed": np.int64(70 + n),
                      f"{name}/checksum": np.float64(sum(x.double().sum() for x in w.values()))})
        for i in range(layers):
            for s_ in ("running_mean", "running_var"):
                store[f"{name}/after/tcn.norms.{i}.{s_}"] = sd[f"tcn.norms.{i}.{s_}"].numpy()
    # delay estimators: known answers from the reference's static methods
    ds = [(0, 0), (10, 10), (16000, 400), (400, 16000), (-5, 7), (3, 0), (1, 2)]
    store["delay/args"] = np.asarray(ds, dtype=np.int64)
    store["delay/out"] = np.asarray([TemporalSyncNet.delay_score(a, v) for a, v in ds], dtype=np.float64)
    g = torch.Generator().manual_seed(99)
    lag_out = []
    for i, (L_, shift, sr, max_lag) in enumerate([(400, 7, 100.0, 0.5), (400, -13, 100.0, 0.5), (257, 40, 100.0, 0.2), (3, 1, 100.0, 0.5),
                                                  (64, 5, 16000.0, 0.5), (500, 0, 50.0, 1.0)]):
        base = torch.randn(L_ + 128, generator=g).numpy()
        a = base[64:64 + L_].copy()
        m = base[64 + shift:64 + shift + L_].copy() + 0.05 * torch.randn(L_, generator=g).numpy()
        m = m[: L_ - (i % 2) * 3]                          # ragged lengths: the shorter one wins
        store[f"lag/{i}/a"], store[f"lag/{i}/m"] = a.astype(np.float32), m.astype(np.float32)
        store[f"lag/{i}/args"] = np.asarray([sr, max_lag], dtype=np.float64)
        lag_out.append(TemporalSyncNet.estimate_av_lag(a, m, sr=sr, max_lag_s=max_lag))
    store["lag/out"] = np.asarray(lag_out, dtype=np.float64)
    # initial weights of the module with the sequence path under a fixed seed (construction order :79-94)
    torch.manual_seed(654)
    net = TemporalSyncNet(in_dim=768, out_dim=256, use_tcn=True)
    store["init/rng_after"] = np.float64(float(torch.rand(1)))
    sd = net.state_dict()
    store["init/keys"] = np.asarray(list(sd.keys()))
    store["init/sums"] = np.asarray([float(v.double().sum()) for v in sd.values()], dtype=np.float64)
    store["init/abs_sums"] = np.asarray([float(v.double().abs().sum()) for v in sd.values()], dtype=np.float64)
    store["cases"] = np.asarray(json.dumps(SEQ_CASES))
    np.savez_compressed(HERE / "temporal_seq.npz", **store)
    print("temporal_seq: oracle-vs-reference worst max-abs-err", worst, "lags", lag_out)


def gcn():
    """Graph side of the trainer: build_adj_from_ocr, SimpleGCN.forward and the degree
    pre-training steps (src/training/forensic_trainer.py:25-53,114-132,214-224)."""
    sys.modules["transformers"] = None
    sys.dont_write_bytecode = True
    sys.path.insert(0, str(REF))
    os.chdir(REF)
    import torch.nn as nn
    import torch.nn.functional as F
    from src.training.forensic_trainer import SimpleGCN, build_adj_from_ocr
    from oracle import gcn_ref as G
    N = 150
    sets = G.synthetic_ocr_sets(N, seed=61)
    ref_adj = build_adj_from_ocr([set(f"phrase{t}" for t in s) for s in sets], thresh=0.12)     # the reference sees strings
    assert np.array_equal(ref_adj, G.build_adj_from_ocr(sets, 0.12))
    for thr in (0.05, 0.3):
        assert np.array_equal(build_adj_from_ocr([set(f"phrase{t}" for t in s) for s in sets], thresh=thr), G.build_adj_from_ocr(sets, thr))
    g = torch.Generator().manual_seed(62)
    T, A, V, U = (torch.randn(N, d, generator=g).numpy() for d in (768, 128, 512, 256))
    X = G.node_features(T, A, V, U)
    w = G.seeded_weights(63)
    net = SimpleGCN(in_dim=416, hid=256, out_dim=128, dropout=0.2).eval()
    assert list(net.state_dict().keys()) == list(w.keys())
    net.load_state_dict(w)
    adj_t, x_t = torch.from_numpy(ref_adj), torch.from_numpy(X)
    with torch.no_grad():
        z_ref = net(x_t, adj_t)
    z_ora = G.gcn_forward(w, x_t, adj_t)
    e_fwd = (z_ref - z_ora).abs().max().item()
    assert e_fwd <= 2e-6, e_fwd
    # the two pre-training steps (:214-224), dropout p = 0 so that the result is RNG-free
    net.train()
    net.drop.p = 0.0
    hg = torch.Generator().manual_seed(64)
    head = nn.Linear(128, 1)
    with torch.no_grad():
        head.weight.copy_(torch.randn(1, 128, generator=hg) * 0.1)
        head.bias.copy_(torch.randn(1, generator=hg) * 0.1)
    opt = torch.optim.Adam(net.parameters(), lr=1e-3, weight_decay=1e-4)
    target_deg = adj_t.sum(dim=-1, keepdim=True) / max(1.0, adj_t.shape[0])
    losses = []
    for _ in range(2):
        Z = net(x_t, adj_t)
        loss = F.mse_loss(torch.sigmoid(head(Z)), target_deg)
        opt.zero_grad(); loss.backward(); opt.step()
        losses.append(float(loss))
    w2, l2 = G.pretrain(w, x_t, adj_t, head.weight.detach(), head.bias.detach(), epochs=2)
    e_pre = max((net.state_dict()[k] - w2[k]).abs().max().item() for k in w)
    assert e_pre <= 2e-6 and max(abs(a - b) for a, b in zip(losses, l2)) <= 1e-6, (e_pre, losses, l2)
    with torch.no_grad():
        net.eval()
        z_after = net(x_t, adj_t)
    offs, toks = G.sets_to_csr(sets)
    np.savez_compressed(HERE / "gcn.npz", N=np.int64(N), set_seed=np.int64(61), weight_seed=np.int64(63), offsets=offs, tokens=toks,
                        adj_rowsum=ref_adj.sum(1), adj_packed=np.packbits(ref_adj.astype(np.uint8), axis=1), T=T[:, :192], A=A[:, :32],
                        V=V[:, :128], U=U[:, :64], X=X, Z=z_ref.numpy(), head_w=head.weight.detach().numpy(),
                        head_b=head.bias.detach().numpy(), losses=np.asarray(losses), Z_after=z_after.numpy(),
                        lin1_w_after_sum=np.float64(net.lin1.weight.double().sum()), lin2_w_after_sum=np.float64(net.lin2.weight.double().sum()),
                        checksum=np.float64(sum(x.double().sum() for x in w.values())))
    print(f"gcn: N={N}, edges={int((ref_adj.sum() - N) / 2)}, forward oracle-vs-reference {e_fwd:.2e}, pretrain {e_pre:.2e}, losses {losses}")


def gnn_model():
    """The integrated variant's in-graph GNN: build_adj_from_ocr_sets (forensic_trainer_integrated.py:77-98,
    restated there as plain Python: importing that module pulls the dataset pipeline in, so the function body is checked
    against the oracle's restatement on the reference's own GNNModel only through the adjacency it feeds) and GNNModel
    (src/models/gnn/gnn_model.py) forward + autograd backward for an upstream gradient."""
    sys.modules["transformers"] = None
    sys.dont_write_bytecode = True
    sys.path.insert(0, str(REF))
    os.chdir(REF)
    from src.models.gnn.gnn_model import GNNModel
    from oracle import gcn_ref as G
    from oracle import gnn_model_ref as M
    try:                                                   # the real adjacency builder, when its module imports here
        from src.training.forensic_trainer_integrated import build_adj_from_ocr_sets as ref_adj_fn
    except Exception as e:                                 # (dataset-pipeline dependencies absent)
        print("forensic_trainer_integrated not importable here:", repr(e)[:120])
        ref_adj_fn = None
    out = {}
    for tag, n, thr in (("b32", 32, 0.12), ("b7", 7, 0.05), ("b150", 150, 0.3)):
        sets = M.synthetic_ocr_sets(n, seed=71 + n)
        adj = M.build_adj_from_ocr_sets(sets, thr)
        if ref_adj_fn is not None:
            ref_adj = ref_adj_fn([set(f"tok{t}" for t in s) for s in sets], overlap_thresh=thr).numpy()
            assert np.array_equal(ref_adj, adj), tag
        g = torch.Generator().manual_seed(72 + n)
        T, A, V, U = (torch.randn(n, d, generator=g).numpy() for d in (768, 128, 512, 256))
        X = torch.from_numpy(G.node_features(T, A, V, U))
        w = M.seeded_weights(73)
        net = GNNModel(in_dim=416, hid=256, out_dim=128, dropout=0.1).eval()
        assert list(net.state_dict().keys()) == list(w.keys())
        net.load_state_dict(w)
        a_t = torch.from_numpy(adj)
        z = net(X, a_t)
        dz = torch.randn(n, 128, generator=g) / n
        z.backward(dz)
        grads = {k: p.grad.clone() for k, p in net.named_parameters()}
        # oracle restatement agrees (forward and every gradient)
        wo = {k: v.clone().requires_grad_(True) for k, v in w.items()}
        zo = M.forward(wo, X, a_t)
        zo.backward(dz)
        e_f = (z.detach() - zo.detach()).abs().max().item()
        e_g = max((grads[k] - wo[k].grad).abs().max().item() for k in w)
        assert e_f <= 2e-6 and e_g <= 2e-6, (tag, e_f, e_g)
        offs, toks = G.sets_to_csr(sets)
        out.update({f"{tag}/n": np.int64(n), f"{tag}/thr": np.float64(thr), f"{tag}/offsets": offs, f"{tag}/tokens": toks, f"{tag}/adj": adj,
                    f"{tag}/T": T[:, :192], f"{tag}/A": A[:, :32], f"{tag}/V": V[:, :128], f"{tag}/U": U[:, :64], f"{tag}/X": X.numpy(),
                    f"{tag}/Z": z.detach().numpy(), f"{tag}/dZ": dz.numpy()})
        for k, gk in grads.items():
            out[f"{tag}/grad/{k}"] = gk.numpy()
        print(f"gnn_model {tag}: n={n} edges={int((adj > 0).sum() // 2)} oracle-vs-reference fwd {e_f:.2e} grads {e_g:.2e}"
              f" (adjacency {'checked against the reference function' if ref_adj_fn is not None else 'oracle restatement only'})")
    out["weight_seed"] = np.int64(73)
    out["checksum"] = np.float64(sum(x.double().sum() for x in M.seeded_weights(73).values()))
    np.savez_compressed(HERE / "gnn_model.npz", **out)


if __name__ == "__main__":
    part = sys.argv[1] if len(sys.argv) > 1 else "all"
    if part == "all":
        for p in ("tier_a", "tier_a_nognn", "metrics", "tier_b", "tier_b_grads", "temporal", "temporal_seq", "init_parity", "gcn", "gnn_model"):
            subprocess.check_call([sys.executable, str(Path(__file__).resolve()), p], cwd=str(REPO))
    else:
        {"tier_a": tier_a, "tier_a_nognn": tier_a_nognn, "metrics": metrics, "tier_b": tier_b, "tier_b_grads": tier_b_grads, "temporal": temporal, "temporal_seq": temporal_seq, "init_parity": init_parity, "gcn": gcn, "gnn_model": gnn_model}[part]()
