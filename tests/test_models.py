import pytest
import torch

from pytorch_distributed_training_amd.models import MODELS, build_model

TORCHVISION_RESNET18_KEYS_SAMPLE = [
    "conv1.weight", "bn1.weight", "bn1.bias", "bn1.running_mean", "bn1.running_var", "bn1.num_batches_tracked",
    "layer1.0.conv1.weight", "layer1.0.bn1.weight", "layer1.1.conv2.weight", "layer2.0.downsample.0.weight",
    "layer2.0.downsample.1.running_var", "layer4.1.bn2.num_batches_tracked", "fc.weight", "fc.bias",
]


def _count(m):
    return sum(p.numel() for p in m.parameters())


def test_resnet18_matches_torchvision_layout():
    m = build_model("resnet18", num_classes=10)
    sd = m.state_dict()
    for k in TORCHVISION_RESNET18_KEYS_SAMPLE:
        assert k in sd
    assert _count(m) == 11_181_642  # torchvision resnet18(num_classes=10)
    assert len(list(m.parameters())) == 62
    assert len(sd) == 122
    assert m(torch.randn(2, 3, 32, 32)).shape == (2, 10)


def test_resnet50_parameter_count_and_names():
    m = build_model("resnet50")
    assert _count(m) == 25_557_032
    assert len(list(m.parameters())) == 161
    sd = m.state_dict()
    for k in ["layer1.0.conv3.weight", "layer1.0.downsample.0.weight", "layer3.5.bn3.bias", "layer4.2.conv2.weight"]:
        assert k in sd
    assert sd["layer2.0.conv2.weight"].shape == (128, 128, 3, 3)  # v1.5: the stride lives on the 3x3 conv
    assert m.layer2[0].conv2.stride == (2, 2) and m.layer2[0].conv1.stride == (1, 1)


def test_resnet50_forward_small_input():
    m = build_model("resnet50", num_classes=7).eval()
    with torch.no_grad():
        assert m(torch.randn(1, 3, 64, 64)).shape == (1, 7)


def test_vit_and_gpt2_param_counts():
    vit = build_model("vit_b_16")
    assert abs(_count(vit) - 86.6e6) < 0.3e6
    # GPT-2 medium is large; count without allocating by using the meta device
    with torch.device("meta"):
        gpt = build_model("gpt2_medium")
    assert abs(_count(gpt) - 354.8e6) < 0.5e6


def test_tiny_transformers_run_and_backprop():
    vit = build_model("vit_tiny")
    out = vit(torch.randn(2, 3, 32, 32))
    assert out.shape == (2, 10)
    out.sum().backward()
    gpt = build_model("gpt2_tiny")
    ids = torch.randint(0, 512, (2, 16))
    loss = gpt(ids, torch.randint(0, 512, (2, 16)))
    assert loss.ndim == 0 and abs(float(loss.detach()) - 6.24) < 0.5  # ~ln(512) at init
    loss.backward()
    assert gpt.wte.weight.grad is not None
    assert gpt(ids).shape == (2, 16, 512)


def test_registry_is_consistent():
    for name, spec in MODELS.items():
        assert spec.name == name and spec.kind in ("image", "lm")
    with pytest.raises(KeyError):
        build_model("alexnet")
