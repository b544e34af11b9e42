"""Not a test: prints the measured Tier-B parity numbers on a GPU box (`python tests/measure_parity.py`); the bounds in the tests are set to <= 2x these."""
import json, sys
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parents[1])); sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parent))
import numpy as np, torch
from helpers import feature_errors, hidden_errors, load_npz
from oracle import encoders_ref as E
from ultrafnd_git_amd.encoders import BertTextEncoder, ClipVisualEncoder
DEV = "cuda"
z = load_npz("tier_b.npz")
for tag in ["bert2_L128", "bert2_L512", "bert2_L40"]:
    meta = json.loads(str(z[f"{tag}/meta"]))
    w = E.seeded_weights(E.bert_shapes(layers=meta["layers"], vocab=meta["vocab"]), meta["weight_seed"])
    enc = BertTextEncoder(layers=meta["layers"], vocab_size=meta["vocab"]); enc.load_state_dict(w); enc = enc.to(DEV)
    ids, mask = torch.from_numpy(z[f"{tag}/ids"]), torch.from_numpy(z[f"{tag}/mask"])
    print(tag, feature_errors(enc(ids, mask).cpu().numpy(), z[f"{tag}/features"]))
for tag in ["vit2_F1", "vit2_F4"]:
    meta = json.loads(str(z[f"{tag}/meta"]))
    w = E.seeded_weights(E.vit_shapes(layers=meta["layers"]), meta["weight_seed"])
    enc = ClipVisualEncoder(layers=meta["layers"]); enc.load_state_dict(w); enc = enc.to(DEV)
    frames = E.synthetic_frames(meta["frame_seed"], meta["B"], meta["F"])
    print(tag, feature_errors(enc(frames).cpu().numpy(), z[f"{tag}/features"]))
# 12 layers small vocab
w = E.seeded_weights(E.bert_shapes(layers=12, vocab=1000), 41)
enc = BertTextEncoder(layers=12, vocab_size=1000); enc.load_state_dict(w); enc = enc.to(DEV)
ids, mask = E.synthetic_tokens(141, 4, 128, vocab=1000)
col = {}
ref_h = E.bert_last_hidden_state(w, ids, mask, collect=col)
print("bert12 feat", feature_errors(enc(ids, mask).cpu(), E.text_features(w, ids, mask)))
for k in (1, 6, 12):
    h = enc.last_hidden_state(ids, mask, n_layers=k).clone().cpu()
    worst = max((hidden_errors(h[i, :int(mask[i].sum())], col[k][i, :int(mask[i].sum())])["rel_rms"] for i in range(4)))
    print("bert12 hidden after layer", k, worst)
w = E.seeded_weights(E.vit_shapes(layers=12), 42)
venc = ClipVisualEncoder(layers=12); venc.load_state_dict(w); venc = venc.to(DEV)
fr = E.synthetic_frames(142, 2, 1)
print("vit12 feat", feature_errors(venc(fr).cpu(), E.visual_features(w, fr)))
col = {}
E.vit_pooled(w, fr.reshape(-1, 3, 224, 224), collect=col)
for k in (1, 6, 12):
    h = venc.hidden_state(fr, n_layers=k).clone().cpu()
    print("vit12 hidden after layer", k, hidden_errors(h, col[k]))
# full size
B, Lq = 128, 512
w = E.seeded_weights(E.bert_shapes(), 51)
enc = BertTextEncoder(); enc.load_state_dict(w); enc = enc.to(DEV)
ids, mask = E.synthetic_tokens(151, B, Lq); mask[5] = 1
rows = [0, 5, 77, 127]
feat = enc(ids, mask).clone()
print("bert full L512 feat", feature_errors(feat[rows].cpu(), E.text_features(w, ids[rows], mask[rows])))
col = {}
E.bert_last_hidden_state(w, ids[rows], mask[rows], collect=col)
for k in (1, 6, 12):
    h = enc.last_hidden_state(ids, mask, n_layers=k)[rows].clone().cpu()
    worst = max(hidden_errors(h[i, :int(mask[r].sum())], col[k][i, :int(mask[r].sum())])["rel_rms"] for i, r in enumerate(rows))
    wa = max(hidden_errors(h[i, :int(mask[r].sum())], col[k][i, :int(mask[r].sum())])["max_abs"] for i, r in enumerate(rows))
    print("bert full hidden after layer", k, worst, wa)
B, Fr = 8, 8
w = E.seeded_weights(E.vit_shapes(), 52)
venc = ClipVisualEncoder(); venc.load_state_dict(w); venc = venc.to(DEV)
frames = E.synthetic_frames(152, B, Fr)
print("vit full F8 feat", feature_errors(venc(frames)[[0, 7]].cpu(), E.visual_features(w, frames[[0, 7]])))
col = {}
E.vit_pooled(w, frames[[0, 7]].reshape(-1, 3, 224, 224), collect=col)
for k in (1, 6, 12):
    h = venc.hidden_state(frames, n_layers=k).clone().cpu().view(B, Fr, 50, 768)[[0, 7]].reshape(-1, 50, 768)
    print("vit full hidden after layer", k, hidden_errors(h, col[k]))

# ---- bf16 residual stream: the same full-size cases
B, Lq = 128, 512
w = E.seeded_weights(E.bert_shapes(), 51)
enc = BertTextEncoder(residual_dtype="bf16"); enc.load_state_dict(w); enc = enc.to(DEV)
ids, mask = E.synthetic_tokens(151, B, Lq); mask[5] = 1
rows = [0, 5, 77, 127]
print("RB16 bert full L512 feat", feature_errors(enc(ids, mask)[rows].cpu(), E.text_features(w, ids[rows], mask[rows])))
col = {}
E.bert_last_hidden_state(w, ids[rows], mask[rows], collect=col)
for k in (1, 6, 12):
    h = enc.last_hidden_state(ids, mask, n_layers=k)[rows].clone().cpu()
    worst = max(hidden_errors(h[i, :int(mask[r].sum())], col[k][i, :int(mask[r].sum())])["rel_rms"] for i, r in enumerate(rows))
    print("RB16 bert full hidden after layer", k, worst)
w = E.seeded_weights(E.vit_shapes(), 52)
venc = ClipVisualEncoder(residual_dtype="bf16"); venc.load_state_dict(w); venc = venc.to(DEV)
frames = E.synthetic_frames(152, 8, 8)
print("RB16 vit full F8 feat", feature_errors(venc(frames)[[0, 7]].cpu(), E.visual_features(w, frames[[0, 7]])))
col = {}
E.vit_pooled(w, frames[[0, 7]].reshape(-1, 3, 224, 224), collect=col)
for k in (1, 6, 12):
    h = venc.hidden_state(frames, n_layers=k).clone().cpu().view(8, 8, 50, 768)[[0, 7]].reshape(-1, 50, 768)
    print("RB16 vit full hidden after layer", k, hidden_errors(h, col[k]))
# end-to-end logits with both
from oracle import tier_a as O
from ultrafnd_git_amd.classifier import DeepTruthClassifier
from ultrafnd_git_amd.fusion import CrossModalTransformer
B, Lq = 32, 128
wt = E.seeded_weights(E.bert_shapes(), 61); wv = E.seeded_weights(E.vit_shapes(), 62)
ids, mask = E.synthetic_tokens(161, B, Lq); frames = E.synthetic_frames(162, B, 1)
batch = O.seeded_batch(163, B); fus_sd, clf_sd = O.seeded_params(1234)
fusion, clf = CrossModalTransformer(), DeepTruthClassifier()
fusion.load_state_dict(fus_sd); clf.load_state_dict(clf_sd)
fusion, clf = fusion.to(DEV).eval(), clf.to(DEV).eval()
rows = [0, 9, 20, 31]
ref_b = {k: v[rows] for k, v in batch.items()}
ref_b["text_features"] = E.text_features(wt, ids[rows], mask[rows]); ref_b["visual_features"] = E.visual_features(wv, frames[rows])
ref = O.forward_batch(fus_sd, clf_sd, ref_b)
for rd in ("fp32", "bf16"):
    tenc, venc = BertTextEncoder(residual_dtype=rd), ClipVisualEncoder(residual_dtype=rd)
    tenc.load_state_dict(wt); venc.load_state_dict(wv); tenc, venc = tenc.to(DEV), venc.to(DEV)
    feats = {k: batch[k].to(DEV) for k in ("audio_features", "temporal_features", "gnn_feat")}
    feats["text_features"] = tenc(ids, mask); feats["visual_features"] = venc(frames)
    with torch.no_grad():
        co = clf(fusion(feats)["fused"], batch["aux"].to(DEV))
    print("end-to-end logits", rd, (co["logits"][rows].cpu() - ref["logits"]).abs().max().item(),
          feature_errors(feats["text_features"][rows].cpu(), ref_b["text_features"]), feature_errors(feats["visual_features"][rows].cpu(), ref_b["visual_features"]))
