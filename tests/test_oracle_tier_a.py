"""CPU: the Tier-A oracle reproduces the reference's golden vectors (tests/golden/tier_a_B*.npz,
minted from the real reference by tests/golden/make_golden.py)."""
import numpy as np
import pytest
import torch

from oracle import tier_a as O
from tests.helpers import assert_digest_close, load_npz, nograd_keys


def _batch(z):
    return {k: torch.from_numpy(z[f"in/{k}"]) for k in
            ("text_features", "audio_features", "visual_features", "temporal_features", "gnn_feat", "aux", "label")}


@pytest.mark.parametrize("B", [2, 4, 32])
def test_forward_matches_reference(B):
    z = load_npz(f"tier_a_B{B}.npz")
    fus, clf = O.seeded_params(int(z["param_seed"]))
    chk = float(sum(v.double().sum() for v in list(fus.values()) + list(clf.values())))
    assert abs(chk - float(z["param_checksum"])) < 1e-9, "seeded parameter generator drifted"
    batch = _batch(z)
    ref_batch = O.seeded_batch(int(z["batch_seed"]), B)
    for k in batch:
        assert torch.equal(batch[k], ref_batch[k])
    out = O.forward_batch(fus, clf, batch, train=False)
    for name, key in (("fused", "fused"), ("fusion_logits", "fusion_logits"), ("logits", "logits"), ("probs", "probs")):
        assert np.abs(out[key].numpy() - z[f"out/{name}"]).max() <= 1e-6
    for k in ("emotion_intensity", "semantic_conflict", "temporal_delay"):
        assert np.abs(out["forensic"][k].numpy() - z[f"out/forensic/{k}"]).max() <= 1e-7


@pytest.mark.parametrize("B", [2, 4, 32])
def test_train_steps_match_reference(B):
    z = load_npz(f"tier_a_B{B}.npz")
    fus, clf = O.seeded_params(int(z["param_seed"]))
    batch = _batch(z)
    _, loss, gf, gc = O.loss_and_grads(fus, clf, batch)
    grads = {**{"fusion." + k: g for k, g in gf.items()}, **{"clf." + k: g for k, g in gc.items()}}
    assert sorted(k for k, g in grads.items() if g is None) == sorted(nograd_keys(z))
    assert sum(g.numel() for g in grads.values() if g is not None) == 12_745_949
    for k, g in grads.items():
        if g is not None:
            assert_digest_close(z, f"grad/{k}", g, rtol=2e-5, atol=1e-9)
    opt = O.AdamWState()
    for step in (1, 2, 3):
        out, l, total = O.train_step(fus, clf, batch, opt, grad_clip=5.0)
        assert abs(l - float(z[f"step{step}/loss"])) <= 1e-6
        assert abs(total - float(z[f"step{step}/grad_norm"])) <= 1e-5 * max(1.0, total)
        assert np.abs(out["logits"].detach().numpy() - z[f"step{step}/logits"]).max() <= 1e-5
        if step in (1, 3):
            for k, p in {**{"fusion." + k: v for k, v in fus.items()}, **{"clf." + k: v for k, v in clf.items()}}.items():
                if p.dim() == 0:
                    continue
                assert_digest_close(z, f"param_step{step}/{k}", p, rtol=1e-5, atol=1e-7)


def test_clip_branches_covered():
    """B=2,4 clip (norm > 5), B=32 does not -- both branches of clip_grad_norm_ are pinned."""
    assert float(load_npz("tier_a_B2.npz")["step1/grad_norm"]) > 5.0
    assert float(load_npz("tier_a_B32.npz")["step1/grad_norm"]) < 5.0


def test_node_is_exercised():
    """Zero-init leaves/gates would make NODE contribute nothing: the seeded
    parameters must make it matter."""
    fus, clf = O.seeded_params(1234)
    b = O.seeded_batch(3, 8)
    a = O.forward_batch(fus, clf, b)["logits"]
    clf2 = {k: (torch.zeros_like(v) if "leaf_logits" in k else v) for k, v in clf.items()}
    assert (a - O.forward_batch(fus, clf2, b)["logits"]).abs().max() > 1e-2


def test_missing_gnn_feat_changes_width():
    """cross_modal_transformer.py:184-195: without gnn_feat the concat is 7680 wide and
    fuse_mlp (8192 in) rejects it -- the restatement keeps that behaviour."""
    fus, clf = O.seeded_params(1234)
    b = O.seeded_batch(3, 2)
    b["gnn_feat"] = None
    with pytest.raises(RuntimeError):
        O.fusion_forward(fus, b)


def test_oracle_without_the_gnn_slot_matches_the_reference_fixture():
    """fusion.yaml `use_gnn: false`: the oracle's 15H head against the outputs of the reference's own module built from such a
    YAML (tests/golden/tier_a_nognn_B4.npz; make_golden.py tier_a_nognn)."""
    import numpy as np
    from tests.helpers import assert_digest_close, load_npz, nograd_keys
    from oracle import tier_a as O
    z = load_npz("tier_a_nognn_B4.npz")
    fus, clf = O.seeded_params(int(z["param_seed"]), use_gnn=False)
    assert "gnn_proj.weight" not in fus and tuple(fus["fuse_mlp.0.weight"].shape) == (1024, 7680)
    assert abs(float(sum(v.double().sum() for v in list(fus.values()) + list(clf.values()))) - float(z["param_checksum"])) < 1e-6
    batch = O.seeded_batch(int(z["batch_seed"]), 4)
    out = O.forward_batch(fus, clf, batch, train=False)
    for name in ("fused", "logits", "probs"):
        assert np.abs(out[name].numpy() - z[f"out/{name}"]).max() <= 1e-6, name
    _, loss, gf, gc = O.loss_and_grads(fus, clf, batch, train=False)
    assert abs(float(loss) - float(z["step1/loss"])) <= 1e-6
    grads = {**{"fusion." + k: g for k, g in gf.items()}, **{"clf." + k: g for k, g in gc.items()}}
    assert sorted(k for k, g in grads.items() if g is None) == sorted(nograd_keys(z))
    for k, g in grads.items():
        if g is not None:
            assert_digest_close(z, f"grad/{k}", g, rtol=1e-5, atol=1e-9)
