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
