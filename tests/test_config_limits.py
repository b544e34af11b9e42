"""CPU: every YAML the reference would accept but the HIP kernels do not is refused at construction, with a message that names
the limit (/root/reference/src/models/fusion/deep_truth_classifier.py:106-117 takes any value)."""
import pytest
import yaml


def _yaml(tmp_path, name, **over):
    base = {"hidden_dim": 512, "dropout": 0.1, "num_classes": 2, "use_aux": True, "aux_dim": 2, "node_trees": 6, "node_depth": 4, "node_tau": 10.0}
    base.update(over)
    p = tmp_path / name
    p.write_text(yaml.safe_dump(base))
    return str(p)


@pytest.mark.parametrize("over,needle", [
    ({"num_classes": 3}, "num_classes == 2"),
    ({"hidden_dim": 384, "input_dim": 384}, "{256, 512, 1024}"),
    ({"node_trees": 9, "node_depth": 4}, "node_trees x node_depth <= 32"),
    ({"node_trees": 2, "node_depth": 7}, "node_depth <= 6"),
    ({"aux_dim": 3}, "{0, 2, 4}"),
])
def test_classifier_yaml_outside_the_kernels_range_is_refused_by_name(tmp_path, over, needle):
    from ultrafnd_git_amd.classifier import DeepTruthClassifier
    with pytest.raises(ValueError) as e:
        DeepTruthClassifier(_yaml(tmp_path, "classifier.yaml", **over))
    assert needle in str(e.value) and "classifier.yaml" in str(e.value)


def test_shipped_and_in_range_classifier_yamls_construct(tmp_path):
    from ultrafnd_git_amd.classifier import DeepTruthClassifier
    DeepTruthClassifier()
    DeepTruthClassifier(_yaml(tmp_path, "c.yaml", hidden_dim=256, input_dim=256, node_trees=8, node_depth=4, aux_dim=4))
    DeepTruthClassifier(_yaml(tmp_path, "d.yaml", use_aux=False, aux_dim=7))      # aux off: its width is not used


def test_fusion_yaml_hidden_outside_the_kernels_range_is_refused_by_name(tmp_path):
    from ultrafnd_git_amd.fusion import CrossModalTransformer
    p = tmp_path / "fusion.yaml"
    p.write_text(yaml.safe_dump({"hidden_dim": 640, "dropout": 0.1}))
    with pytest.raises(ValueError) as e:
        CrossModalTransformer(str(p))
    assert "{256, 512, 1024}" in str(e.value)


def test_forensic_coattention_forward_names_the_parent():
    import torch
    from ultrafnd_git_amd.fusion import ForensicCoAttention
    blk = ForensicCoAttention(512)
    assert {k for k, _ in blk.named_parameters()} >= {"q.weight", "k.bias", "v.weight", "evidence_proj.0.weight", "evidence_proj.2.bias"}
    with pytest.raises(RuntimeError) as e:
        blk(torch.zeros(2, 512), torch.zeros(2, 512), torch.zeros(2, 3))
    assert "CrossModalTransformer" in str(e.value)
