"""CPU: the graph-side oracle (oracle/gcn_ref.py) against the fixture minted from the REAL reference
(tests/golden/make_golden.py gcn: build_adj_from_ocr, SimpleGCN.forward, two Adam pre-training steps)."""
import numpy as np
import torch

from oracle import gcn_ref as G
from tests.helpers import load_npz


def _fixture():
    z = load_npz("gcn.npz")
    n = int(z["N"])
    sets = G.synthetic_ocr_sets(n, int(z["set_seed"]))
    adj = np.unpackbits(z["adj_packed"], axis=1)[:, :n].astype(np.float32)
    return z, n, sets, adj


def test_adjacency_matches_reference_and_csr_roundtrip():
    z, n, sets, adj = _fixture()
    assert np.array_equal(G.build_adj_from_ocr(sets, 0.12), adj)
    assert np.array_equal(adj.sum(1), z["adj_rowsum"])
    offs, toks = G.sets_to_csr(sets)
    assert np.array_equal(offs, z["offsets"]) and np.array_equal(toks, z["tokens"])
    assert G.jaccard(set(), set()) == 0.0 and G.jaccard({1}, set()) == 0.0 and G.jaccard({1, 2}, {2, 3}) == 1 / (3 + 1e-9)


def test_node_features_and_forward_match_reference():
    z, n, sets, adj = _fixture()
    w = G.seeded_weights(int(z["weight_seed"]))
    assert abs(float(sum(x.double().sum() for x in w.values())) - float(z["checksum"])) < 1e-9
    X = G.node_features(np.pad(z["T"], ((0, 0), (0, 0))), z["A"], z["V"], z["U"])
    assert np.abs(X - z["X"]).max() <= 1e-7
    out = G.gcn_forward(w, torch.from_numpy(z["X"]), torch.from_numpy(adj))
    assert (out - torch.from_numpy(z["Z"])).abs().max().item() <= 2e-6


def test_pretrain_steps_match_reference():
    z, n, sets, adj = _fixture()
    w = G.seeded_weights(int(z["weight_seed"]))
    w2, losses = G.pretrain(w, torch.from_numpy(z["X"]), torch.from_numpy(adj), torch.from_numpy(z["head_w"]), torch.from_numpy(z["head_b"]),
                            epochs=2)
    assert np.abs(np.asarray(losses) - z["losses"]).max() <= 1e-6
    assert abs(float(w2["lin1.weight"].double().sum()) - float(z["lin1_w_after_sum"])) <= 1e-4
    assert abs(float(w2["lin2.weight"].double().sum()) - float(z["lin2_w_after_sum"])) <= 1e-4
    out = G.gcn_forward(w2, torch.from_numpy(z["X"]), torch.from_numpy(adj))
    assert (out - torch.from_numpy(z["Z_after"])).abs().max().item() <= 5e-6


def test_gnn_model_oracle_matches_the_reference_fixture():
    """oracle.gnn_model_ref (the integrated variant's GNNModel + weighted adjacency) against tests/golden/gnn_model.npz,
    which make_golden.py minted from the reference's own GNNModel / build_adj_from_ocr_sets."""
    import torch
    from oracle import gcn_ref as G
    from oracle import gnn_model_ref as M
    z = load_npz("gnn_model.npz")
    w = M.seeded_weights(int(z["weight_seed"]))
    assert abs(sum(x.double().sum() for x in w.values()).item() - float(z["checksum"])) <= 1e-9
    for tag in ("b32", "b7", "b150"):
        n, thr = int(z[f"{tag}/n"]), float(z[f"{tag}/thr"])
        offs, toks = z[f"{tag}/offsets"], z[f"{tag}/tokens"]
        sets = [set(int(t) for t in toks[offs[i]:offs[i + 1]]) for i in range(n)]
        adj = M.build_adj_from_ocr_sets(sets, thr)
        assert np.array_equal(adj, z[f"{tag}/adj"]) and np.array_equal(adj, adj.T) and np.all(np.diag(adj) == 0)
        X = G.node_features(*(np.pad(z[f"{tag}/{k}"], ((0, 0), (0, 8))) for k in "TAVU"))      # extra columns are never read
        assert np.abs(X - z[f"{tag}/X"]).max() <= 1e-7
        wo = {k: v.clone().requires_grad_(True) for k, v in w.items()}
        out = M.forward(wo, torch.from_numpy(z[f"{tag}/X"]), torch.from_numpy(adj))
        assert np.abs(out.detach().numpy() - z[f"{tag}/Z"]).max() <= 2e-6
        out.backward(torch.from_numpy(z[f"{tag}/dZ"]))
        for k in w:
            assert np.abs(wo[k].grad.numpy() - z[f"{tag}/grad/{k}"]).max() <= 2e-6, (tag, k)
