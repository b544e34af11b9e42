"""GPU parity (through the C ABI): Tier-A HIP path vs the reference's golden vectors and the oracle.

Tolerances (fp32 everywhere; only the summation order differs from the CPU reference):
forward outputs 5e-5 abs, gradients 2e-4 relative to the tensor's scale, parameters after
AdamW steps 1e-5 relative -- all far inside north_star's 1e-3 logit bound.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests.helpers import assert_digest_close, load_npz, nograd_keys

pytestmark = pytest.mark.gpu

FEATS = ("text_features", "audio_features", "visual_features", "temporal_features", "gnn_feat")


def _modules(z, dropout_off=True):
    from oracle import tier_a as O
    from ultrafnd_git_amd.classifier import DeepTruthClassifier
    from ultrafnd_git_amd.fusion import CrossModalTransformer
    fus_sd, clf_sd = O.seeded_params(int(z["param_seed"]))
    fusion, clf = CrossModalTransformer(), DeepTruthClassifier()
    fusion.load_state_dict(fus_sd)
    clf.load_state_dict(clf_sd)
    fusion, clf = fusion.to("cuda"), clf.to("cuda")
    if dropout_off:
        fusion.dropout = 0.0
        clf.dropout = 0.0
        clf.node_dropout = 0.0
    return fusion, clf


def _batch(z, dev="cuda"):
    return {k: torch.from_numpy(z[f"in/{k}"]).to(dev) for k in FEATS + ("aux", "label")}


@pytest.mark.parametrize("B", [2, 4, 32])
def test_forward_matches_reference(B):
    z = load_npz(f"tier_a_B{B}.npz")
    fusion, clf = _modules(z)
    fusion.eval(); clf.eval()
    b = _batch(z)
    with torch.no_grad():
        fo = fusion({k: b[k] for k in FEATS})
        co = clf(fo["fused"], b["aux"])
    errs = {}
    for name, t in (("fused", fo["fused"]), ("fusion_logits", fo["logits"]), ("logits", co["logits"]), ("probs", co["probs"])):
        errs[name] = float(np.abs(t.cpu().numpy() - z[f"out/{name}"]).max())
    for k in ("emotion_intensity", "semantic_conflict", "temporal_delay"):
        errs[k] = float(np.abs(fo["forensic"][k].cpu().numpy() - z[f"out/forensic/{k}"]).max())
    print("forward max-abs-err", B, errs)
    assert max(errs.values()) <= 5e-5, errs
    assert abs(float(co["temperature"]) - float(z["out/temperature"])) < 1e-7
    assert clf.predict(fo["fused"], b["aux"]).shape == (B,)


@pytest.mark.parametrize("B", [2, 4, 32])
def test_backward_matches_reference(B):
    z = load_npz(f"tier_a_B{B}.npz")
    fusion, clf = _modules(z)
    fusion.train(); clf.train()
    b = _batch(z)
    fo = fusion({k: b[k] for k in FEATS})
    co = clf(fo["fused"], b["aux"])
    loss = F.cross_entropy(co["logits"], b["label"])
    loss.backward()
    assert abs(loss.item() - float(z["step1/loss"])) <= 2e-5
    grads = {**{"fusion." + k: p.grad for k, p in fusion.named_parameters()},
             **{"clf." + k: p.grad for k, p in clf.named_parameters()}}
    assert sorted(k for k, g in grads.items() if g is None) == sorted(nograd_keys(z))
    bad = []
    for k, g in grads.items():
        if g is None:
            continue
        try:
            assert_digest_close(z, f"grad/{k}", g, rtol=2e-4, atol=1e-8)
        except AssertionError as e:
            bad.append(str(e)[:300])
    assert not bad, "\n".join(bad)


def test_native_cross_entropy_and_aux_head():
    """ufnd_softmax_ce == F.cross_entropy; a loss on the fusion's own logits reaches classifier.*"""
    from oracle import tier_a as O
    from ultrafnd_git_amd.functional import cross_entropy
    z = load_npz("tier_a_B4.npz")
    fusion, clf = _modules(z)
    fusion.train()
    b = _batch(z)
    fo = fusion({k: b[k] for k in FEATS})
    loss = cross_entropy(fo["logits"], b["label"])
    loss.backward()
    fus_sd, _ = O.seeded_params(int(z["param_seed"]))
    p = {k: v.clone().requires_grad_(True) for k, v in fus_sd.items()}
    cb = {k: v.cpu() for k, v in b.items()}
    ref = O.fusion_forward(p, cb)
    rl = F.cross_entropy(ref["logits"], cb["label"])
    rl.backward()
    assert abs(loss.item() - rl.item()) < 2e-5
    for k in ("classifier.weight", "classifier.bias", "fuse_mlp.3.weight", "text_proj.weight", "attn_tv.q.weight",
              "attn_vu.evidence_proj.0.weight"):
        g = dict(fusion.named_parameters())[k].grad.cpu()
        err = (g - p[k].grad).abs().max().item()
        assert err <= 2e-4 * max(1e-3, p[k].grad.abs().max().item()), (k, err)


@pytest.mark.parametrize("B", [2, 32])
def test_train_steps_match_reference(B):
    """forward + CE + backward + clip_grad_norm_(5) + AdamW, 3 steps, vs the reference's numbers."""
    from ultrafnd_git_amd.arena import rehome
    from ultrafnd_git_amd.functional import cross_entropy
    from ultrafnd_git_amd.optim import FusedAdamW
    z = load_npz(f"tier_a_B{B}.npz")
    fusion, clf = _modules(z)
    arena = rehome([clf, fusion], ["clf.", "fusion."])
    assert arena.n_grad >= 12_745_949
    opt = FusedAdamW(arena, lr=2e-4, weight_decay=1e-4, max_norm=5.0)
    fusion.train(); clf.train()
    b = _batch(z)
    for step in (1, 2, 3):
        fo = fusion({k: b[k] for k in FEATS})
        co = clf(fo["fused"], b["aux"])
        loss = cross_entropy(co["logits"], b["label"])
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        st = opt.state.read()
        print(f"B={B} step {step}: loss {loss.item():.6f} (ref {float(z[f'step{step}/loss']):.6f}) "
              f"gnorm {st.grad_norm:.6f} (ref {float(z[f'step{step}/grad_norm']):.6f}) clip {st.clip_coef:.4f}")
        assert abs(loss.item() - float(z[f"step{step}/loss"])) <= 5e-5
        # The reference's clip_grad_norm_ reduces 8.4M-element tensors in fp32 on the CPU and lands
        # ~2.5e-4..6e-4 LOW (torch.linalg.vector_norm vs its own float64 result); the HIP reduction
        # (fp32 per thread, float64 across blocks) matches the float64 norm of the reference's own
        # gradients.  Bound: 1e-3 vs the reference's number, 5e-5 vs the exact norm (step 1, from
        # the per-tensor digests of the reference's gradients).
        assert abs(st.grad_norm - float(z[f"step{step}/grad_norm"])) <= 1e-3 * float(z[f"step{step}/grad_norm"])
        if step == 1:
            exact = float(np.sqrt(sum(float(z[k]) ** 2 for k in z.files if k.startswith("grad/") and k.endswith("/norm"))))
            assert abs(st.grad_norm - exact) <= 5e-5 * exact, (st.grad_norm, exact)
        # B=2 clips at step 1 with a coefficient that differs from the reference's by its own fp32
        # norm error (see below), which shifts later logits by ~3e-4 (|logits| ~ 8); B=32 never clips
        # and is held to 1e-4.
        assert np.abs(co["logits"].detach().cpu().numpy() - z[f"step{step}/logits"]).max() <= (1e-3 if B == 2 else 1e-4)
        if step in (1, 3):
            for k, p in list(("fusion." + k, p) for k, p in fusion.named_parameters()) + \
                    list(("clf." + k, p) for k, p in clf.named_parameters()):
                if p.dim() == 0:
                    continue
                rt, at = (2e-4, 2e-6) if B == 2 else (2e-5, 2e-7)
                assert_digest_close(z, f"param_step{step}/{k}", p.detach(), rtol=rt, atol=at, what=f"step{step}")
    assert int(opt.state.read().step) == 3


def test_dropout_is_deterministic_and_calibrated():
    """Train-mode dropout cannot be bit-matched across devices: check it statistically --
    keep-rate of fuse_mlp's output, mean preserved, and backward regenerates the same mask."""
    z = load_npz("tier_a_B32.npz")
    fusion, clf = _modules(z, dropout_off=False)
    fusion.train()
    b = _batch(z)
    feats = {k: b[k] for k in FEATS}
    fused_t = fusion(feats)["fused"]
    fusion.eval()
    with torch.no_grad():
        fused_e = fusion(feats)["fused"]
    kept = (fused_t != 0) | (fused_e == 0)
    rate = kept.float().mean().item()
    assert 0.80 <= rate <= 0.97, rate            # two stacked p=0.1 dropouts feed it
    fusion.train()
    a = fusion(feats)["fused"].detach().clone()
    c = fusion(feats)["fused"].detach()
    assert not torch.equal(a, c), "mask must change between steps"
    # gradient w.r.t. a dropped output unit is exactly zero in fuse_mlp.3.bias' contribution
    out = fusion(feats)["fused"]
    out.sum().backward()
    gb = fusion.fuse_mlp[3].bias.grad
    assert torch.isfinite(gb).all() and gb.abs().sum() > 0


def test_cpu_use_fails_loudly():
    from ultrafnd_git_amd._lib import UltrafndHipError
    from ultrafnd_git_amd.fusion import CrossModalTransformer
    m = CrossModalTransformer()
    with pytest.raises(UltrafndHipError):
        m({k: torch.zeros(2, d) for k, d in zip(FEATS, (768, 128, 512, 256, 128))})


def test_temporal_align_matches_reference():
    """TemporalSyncNet.align on the HIP path vs the reference's outputs (tests/golden/temporal.npz); fp32, 2e-6."""
    from oracle import temporal_ref as T
    from ultrafnd_git_amd.temporal import TemporalSyncNet
    z = load_npz("temporal.npz")
    net = TemporalSyncNet(in_dim=768, out_dim=256)
    net.load_state_dict(T.seeded_weights(int(z["weight_seed"])))
    net = net.to("cuda").eval()                    # the golden was minted from the reference in eval mode (dropout off)
    t, v = torch.from_numpy(z["t"]), torch.from_numpy(z["v"])
    got = net.align_batch(t, v).cpu().numpy()
    err = np.abs(got - z["out"]).max()
    one = net.align(z["t"][0], z["t"][0])
    print("temporal align max-abs-err", err)
    assert err <= 2e-6 and np.abs(one - z["out_self"]).max() <= 2e-6 and one.dtype == np.float32
    wide = net.align_batch(t, torch.cat([torch.from_numpy(z["t"]), torch.ones(6, 40)], dim=1))    # Dv > D: truncated
    assert np.abs(wide.cpu().numpy() - T.align(T.seeded_weights(51), t, t).numpy()).max() <= 2e-6


def test_temporal_align_train_mode_applies_the_projections_dropout():
    """The reference's align() is under torch.inference_mode, which does not switch dropout off, and the cache builder
    never calls .eval(): in (default) train mode the Dropout(0.1) between the two Linears is live.  With an identity-like
    second Linear the output shows the mask: every hidden unit is either dropped or scaled by 1 / (1 - p); the keep
    fraction is 1 - p; consecutive calls draw different masks; eval mode is deterministic."""
    from ultrafnd_git_amd.temporal import TemporalSyncNet
    torch.manual_seed(3)
    net = TemporalSyncNet(in_dim=768, out_dim=256)
    with torch.no_grad():
        net.proj[3].weight.zero_()
        net.proj[3].weight[:, :256].copy_(torch.eye(256))           # out[:, i] = dropped hidden unit i
        net.proj[3].bias.zero_()
    net = net.to("cuda")
    g = torch.Generator().manual_seed(4)
    t, v = torch.randn(64, 768, generator=g), torch.randn(64, 512, generator=g)
    assert net.training
    a, b = net.align_batch(t, v).clone(), net.align_batch(t, v).clone()
    net.eval()
    e1, e2 = net.align_batch(t, v).clone(), net.align_batch(t, v).clone()
    assert torch.equal(e1, e2)
    p = net.proj[2].p
    live = e1.abs() > 1e-4
    for x in (a, b):
        ratio = (x / e1)[live]
        kept = ratio.abs() > 0.5
        assert (ratio[kept] - 1.0 / (1.0 - p)).abs().max().item() <= 1e-5 and (ratio[~kept]).abs().max().item() == 0.0
        frac = kept.float().mean().item()
        assert abs(frac - (1.0 - p)) <= 4.0 * (p * (1 - p) / kept.numel()) ** 0.5 + 1e-3, frac
    assert not torch.equal(a, b)                                    # a fresh mask per call
    net.train()
    one = net.align(t[0].numpy(), v[0].numpy())
    assert one.shape == (256,) and one.dtype == np.float32


DEV = "cuda"


@pytest.mark.parametrize("w,eps", [((1.0, 1.0), 0.05), ((0.7, 1.9), 0.0), ((0.6, 1.4), 0.05)])
def test_weighted_label_smoothed_ce_matches_torch(w, eps):
    """The integrated variant's criterion (forensic_trainer_integrated.py:154-166): nn.CrossEntropyLoss(weight,
    label_smoothing) -- loss and d loss / d logits against torch autograd."""
    from ultrafnd_git_amd import _lib as L
    from ultrafnd_git_amd.state import StepStateBuffer
    g = torch.Generator().manual_seed(3)
    for B in (1, 5, 300):
        logits = (torch.randn(B, 2, generator=g) * 2).requires_grad_(True)
        y = torch.randint(0, 2, (B,), generator=g)
        ref = torch.nn.CrossEntropyLoss(weight=torch.tensor(w), label_smoothing=eps)(logits, y)
        ref.backward()
        lg, yd = logits.detach().to(DEV), y.to(DEV)
        d = torch.empty(B, 2, device=DEV)
        rows = torch.empty(B, device=DEV)
        st = StepStateBuffer(torch.device(DEV))
        L.check(L.lib().ufnd_softmax_ce_weighted(lg.data_ptr(), yd.data_ptr(), B, w[0], w[1], eps, rows.data_ptr(), d.data_ptr(), st.ptr,
                                                 L.stream_ptr(lg.device)), "ce")
        assert abs(float(st.read().loss) - float(ref)) <= 2e-6 * max(1.0, abs(float(ref)))
        assert abs(float(rows.sum()) - float(ref)) <= 1e-5
        assert (d.cpu() - logits.grad).abs().max().item() <= 2e-7 + 1e-5 * logits.grad.abs().max().item()


def test_cosine_schedule_and_criterion_options_in_the_trainer():
    """TrainConfig(label_smoothing, class_weighting, use_cosine): the integrated variant's options on the same step."""
    import math
    from ultrafnd_git_amd.trainer import ForensicTrainer, TrainConfig, synthetic_cache
    cfg = TrainConfig(data_root="", ocr_phrase_pkl=None, out_dir="/tmp/ufnd_int_t", batch_size=16, epochs=4, device=DEV, seed=3,
                      label_smoothing=0.05, class_weighting=True, use_cosine=True, min_lr_scale=0.1)
    cache = synthetic_cache(80, seed=2)
    tr = ForensicTrainer(cfg, cache=cache)
    y = cache["labels"]
    assert abs(tr.head.ce_w[0] - 0.5 * len(y) / max(1, (y == 0).sum())) < 1e-9
    loss, _ = tr._epoch_loop(tr.train_loader, "train")
    assert math.isfinite(loss)
    lrs = []
    for _ in range(4):
        tr.scheduler.step()
        lrs.append(tr.scheduler.get_last_lr()[0])
    want = [2e-5 + (2e-4 - 2e-5) * (1 + math.cos(math.pi * e / 4)) / 2 for e in (1, 2, 3, 4)]
    assert max(abs(a - b) for a, b in zip(lrs, want)) <= 1e-9


def test_temporal_sequence_path_matches_reference():
    """TemporalSyncNet(use_tcn=True).forward on the HIP path vs the reference's outputs (tests/golden/temporal_seq.npz):
    eval mode (running statistics) and train mode with dropout p=0 (batch statistics + the running-statistics update
    and num_batches_tracked), four geometries (residual / none at block 0, even kernel, T=1); fp32, 2e-5."""
    import json
    from oracle import temporal_ref as T
    from ultrafnd_git_amd.temporal import TemporalSyncNet
    g = load_npz("temporal_seq.npz")
    for name, in_dim, out_dim, hid, layers, k, B, Tn, Dt in json.loads(str(g["cases"])):
        net = TemporalSyncNet(in_dim=in_dim, out_dim=out_dim, use_tcn=True, tcn_hid=hid, tcn_layers=layers, tcn_kernel=k, dropout=0.0)
        net.load_state_dict(T.seq_seeded_weights(int(g[f"{name}/weight_seed"]), in_dim, out_dim, hid, layers, k))
        net = net.to("cuda")
        ts, vs = torch.from_numpy(g[f"{name}/text_seq"]), torch.from_numpy(g[f"{name}/vis_seq"])
        net.eval()
        e_eval = np.abs(net(ts, vs).cpu().numpy() - g[f"{name}/out_eval"]).max()
        net.train()
        e_train = np.abs(net(ts.cuda(), vs.cuda()).cpu().numpy() - g[f"{name}/out_train"]).max()
        sd = net.state_dict()
        e_stats = max(np.abs(sd[f"tcn.norms.{i}.{s}"].cpu().numpy() - g[f"{name}/after/tcn.norms.{i}.{s}"]).max()
                      for i in range(layers) for s in ("running_mean", "running_var"))
        print(f"temporal seq {name}: eval {e_eval:.2e} train {e_train:.2e} running stats {e_stats:.2e}")
        assert e_eval <= 2e-5 and e_train <= 2e-5 and e_stats <= 2e-5, name
        assert all(int(sd[f"tcn.norms.{i}.num_batches_tracked"]) == 1 for i in range(layers))


def test_temporal_sequence_path_properties():
    """Size-independent checks at a FakeSV-like shape (B=32 clips x 64 frames, 384+384 channels, default TCN): a clip's
    eval-mode output does not depend on the other clips of the batch; the module default (train mode, dropout 0.1)
    draws a fresh mask per call; a wrong channel count is refused as nn.Conv1d refuses it."""
    from ultrafnd_git_amd.temporal import TemporalSyncNet
    torch.manual_seed(5)
    net = TemporalSyncNet(in_dim=768, out_dim=256, use_tcn=True).to("cuda")
    ts, vs = torch.randn(32, 64, 384, device="cuda"), torch.randn(32, 64, 384, device="cuda")
    net.eval()
    full = net(ts, vs)
    assert full.shape == (32, 256) and torch.isfinite(full).all()
    assert torch.equal(net(ts[5:9], vs[5:9]), full[5:9])
    net.train()
    a, b = net(ts, vs), net(ts, vs)
    assert not torch.equal(a, b) and torch.isfinite(a).all()
    with pytest.raises(RuntimeError):
        net(ts, vs[:, :, :100])


def test_fusion_without_the_gnn_slot_matches_the_reference():
    """fusion.yaml `use_gnn: false` (cross_modal_transformer.py:88,101-102,114-120): no gnn_proj, fuse_mlp.0 over the 15H concat,
    gnn_feat ignored -- forward and every gradient against the reference's own module built from such a YAML
    (tests/golden/tier_a_nognn_B4.npz), through dims.gnn_dim == 0 of the C ABI."""
    from oracle import tier_a as O
    from ultrafnd_git_amd.classifier import DeepTruthClassifier
    from ultrafnd_git_amd.fusion import CrossModalTransformer
    z = load_npz("tier_a_nognn_B4.npz")
    fus_sd, clf_sd = O.seeded_params(int(z["param_seed"]), use_gnn=False)
    fusion, clf = CrossModalTransformer("configs/model_configs/fusion_nognn.yaml"), DeepTruthClassifier()
    assert not fusion.use_gnn and fusion.fused_dim == 15 * 512 and not hasattr(fusion, "gnn_proj")
    assert list(fusion.state_dict().keys()) == list(fus_sd.keys())
    fusion.load_state_dict(fus_sd); clf.load_state_dict(clf_sd)
    fusion, clf = fusion.to("cuda"), clf.to("cuda")
    fusion.dropout = clf.dropout = clf.node_dropout = 0.0
    batch = O.seeded_batch(int(z["batch_seed"]), 4)
    b = {k: v.to("cuda") for k, v in batch.items()}
    fusion.eval(); clf.eval()
    with torch.no_grad():
        fo = fusion({k: b[k] for k in FEATS})                      # gnn_feat present: ignored, as in the reference
        co = clf(fo["fused"], b["aux"])
        fo2 = fusion({k: b[k] for k in FEATS if k != "gnn_feat"})  # ... and absent
    assert torch.equal(fo["fused"], fo2["fused"])
    for name, t in (("fused", fo["fused"]), ("logits", co["logits"]), ("probs", co["probs"])):
        assert float(np.abs(t.cpu().numpy() - z[f"out/{name}"]).max()) <= 5e-5, name
    fusion.train(); clf.train()
    fo = fusion({k: b[k] for k in FEATS})
    co = clf(fo["fused"], b["aux"])
    loss = F.cross_entropy(co["logits"], b["label"])
    loss.backward()
    assert abs(loss.item() - float(z["step1/loss"])) <= 2e-5
    grads = {**{"fusion." + k: p.grad for k, p in fusion.named_parameters()}, **{"clf." + k: p.grad for k, p in clf.named_parameters()}}
    assert sorted(k for k, g in grads.items() if g is None) == sorted(nograd_keys(z))
    for k, g in grads.items():
        if g is not None:
            assert_digest_close(z, f"grad/{k}", g, rtol=2e-4, atol=1e-8)
    # the trainer refuses a head without the slot while its own use_gnn is true (it would feed gnn_feat to nothing), and
    # use_gnn=False with the 16H head, where the reference itself raises (7,680 columns into an 8,192-wide Linear)
    from ultrafnd_git_amd.trainer import ForensicTrainer, TrainConfig, synthetic_cache
    with pytest.raises(ValueError):
        ForensicTrainer(TrainConfig(data_root="", ocr_phrase_pkl=None, out_dir="/tmp/ufnd_nognn", device="cuda", use_gnn=False), cache=synthetic_cache(32, seed=1))
