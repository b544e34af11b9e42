"""CPU: the Tier-B encoder oracle reproduces the third-party outputs stored in tests/golden/tier_b.npz
(BertModel / CLIPVisionModelWithProjection built from local configs with seeded weights)."""
import json

import numpy as np
import pytest
import torch

from oracle import encoders_ref as E
from tests.helpers import load_npz


@pytest.mark.parametrize("tag", ["bert2_L128", "bert2_L512", "bert2_L40"])
def test_text_features(tag):
    z = load_npz("tier_b.npz")
    meta = json.loads(str(z[f"{tag}/meta"]))
    w = E.seeded_weights(E.bert_shapes(layers=meta["layers"], vocab=meta["vocab"]), meta["weight_seed"])
    assert abs(float(sum(v.double().sum() for v in w.values())) - meta["checksum"]) < 1e-6
    ids, mask = torch.from_numpy(z[f"{tag}/ids"]), torch.from_numpy(z[f"{tag}/mask"])
    hid = E.bert_last_hidden_state(w, ids, mask)
    n0 = int(mask[0].sum())
    assert np.abs(hid[0, :n0].numpy() - z[f"{tag}/hidden_row0"]).max() <= 2e-5
    assert np.abs(E.masked_meanpool_l2(hid, mask).numpy() - z[f"{tag}/features"]).max() <= 2e-6


@pytest.mark.parametrize("tag", ["vit2_F1", "vit2_F4"])
def test_visual_features(tag):
    z = load_npz("tier_b.npz")
    meta = json.loads(str(z[f"{tag}/meta"]))
    w = E.seeded_weights(E.vit_shapes(layers=meta["layers"]), meta["weight_seed"])
    frames = E.synthetic_frames(meta["frame_seed"], meta["B"], meta["F"])
    assert abs(float(frames.double().sum()) - meta["frames_checksum"]) < 1e-6
    assert np.abs(E.visual_features(w, frames).numpy() - z[f"{tag}/features"]).max() <= 2e-6


def test_pooling_edge_cases():
    """text_blocks.py:82-86,100: all-masked row -> denom clamp 1e-6 -> zero vector, no NaN."""
    h = torch.randn(2, 5, 8)
    m = torch.tensor([[1, 1, 0, 0, 0], [0, 0, 0, 0, 0]])
    f = E.masked_meanpool_l2(h, m)
    assert torch.isfinite(f).all() and f[1].abs().max() == 0
    assert abs(f[0].norm().item() - 1.0) < 1e-5
    assert torch.allclose(E.field_mean_l2(f[None, :1]), f[:1], atol=1e-6)


def test_temporal_align_matches_reference():
    """oracle/temporal_ref.py vs the reference's TemporalSyncNet.align outputs (tests/golden/temporal.npz)."""
    from oracle import temporal_ref as T
    z = load_npz("temporal.npz")
    w = T.seeded_weights(int(z["weight_seed"]))
    assert abs(float(sum(x.double().sum() for x in w.values())) - float(z["checksum"])) < 1e-9
    t, v = torch.from_numpy(z["t"]), torch.from_numpy(z["v"])
    assert np.abs(T.align(w, t, v).numpy() - z["out"]).max() <= 1e-6
    assert np.abs(T.align(w, t[:1], t[:1])[0].numpy() - z["out_self"]).max() <= 1e-6


def test_temporal_sequence_path_matches_reference():
    """oracle/temporal_ref.py::sequence_forward vs the reference's TemporalSyncNet(use_tcn=True).forward in eval and
    in train mode (dropout p=0), including the BatchNorm running statistics the train-mode call leaves behind
    (tests/golden/temporal_seq.npz; four geometries: residual / no residual at block 0, even kernel, T=1)."""
    from oracle import temporal_ref as T
    g = load_npz("temporal_seq.npz")
    for n, (name, in_dim, out_dim, hid, layers, k, B, Tn, Dt) in enumerate(json.loads(str(g["cases"]))):
        w = T.seq_seeded_weights(int(g[f"{name}/weight_seed"]), in_dim, out_dim, hid, layers, k)
        assert abs(float(sum(x.double().sum() for x in w.values())) - float(g[f"{name}/checksum"])) < 1e-6
        ts, vs = torch.from_numpy(g[f"{name}/text_seq"]), torch.from_numpy(g[f"{name}/vis_seq"])
        o_eval, _ = T.sequence_forward(w, ts, vs, layers, k, train=False)
        o_train, stats = T.sequence_forward(w, ts, vs, layers, k, train=True)
        assert (o_eval - torch.from_numpy(g[f"{name}/out_eval"])).abs().max().item() <= 2e-5, name
        assert (o_train - torch.from_numpy(g[f"{name}/out_train"])).abs().max().item() <= 2e-5, name
        for kk, v in stats.items():
            assert (v - torch.from_numpy(g[f"{name}/after/{kk}"])).abs().max().item() <= 2e-5, (name, kk)


def test_oracle_encoder_gradients_match_the_third_party_autograd_fixture():
    """tier_b_grads.npz: digests of every encoder gradient for a seeded probe loss, stored after make_golden.py checked the
    oracle's autograd against the installed third-party classes' own autograd (2e-6 .. 7e-6 relative).  The reference never
    trains its encoders (text_blocks.py:52,63): 'parity unpinned by the reference'."""
    import json
    from tests.helpers import assert_digest_close, load_npz
    from oracle import encoders_ref as E
    z = load_npz("tier_b_grads.npz")
    meta = json.loads(str(z["bert2_L64/meta"]))
    w = E.seeded_weights(E.bert_shapes(layers=meta["layers"], vocab=meta["vocab"]), meta["weight_seed"])
    ids, mask = torch.from_numpy(z["bert2_L64/ids"]), torch.from_numpy(z["bert2_L64/mask"])
    _, g = E.text_feature_grads(w, ids, mask, meta["loss_seed"])
    assert len(meta["keys"]) == 37
    for k in meta["keys"]:
        assert_digest_close(z, f"bert2_L64/grad/{k}", g[k], 1e-4, 1e-7, k)
    meta = json.loads(str(z["vit2_F2/meta"]))
    w = E.seeded_weights(E.vit_shapes(layers=meta["layers"]), meta["weight_seed"])
    _, g = E.visual_feature_grads(w, E.synthetic_frames(meta["frame_seed"], meta["B"], meta["F"]), meta["loss_seed"])
    for k in meta["keys"]:
        assert_digest_close(z, f"vit2_F2/grad/{k}", g[k], 1e-4, 1e-7, k)
