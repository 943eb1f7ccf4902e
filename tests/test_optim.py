import pytest
import torch
import torch.nn as nn

from pytorch_distributed_training_amd.optim import FusedOptimizer


def _models():
    torch.manual_seed(0)
    a = nn.Sequential(nn.Linear(8, 16), nn.Tanh(), nn.Linear(16, 3))
    b = nn.Sequential(nn.Linear(8, 16), nn.Tanh(), nn.Linear(16, 3))
    b.load_state_dict(a.state_dict())
    return a, b


@pytest.mark.parametrize(
    "mode,ref",
    [
        ("adam", lambda ps: torch.optim.Adam(ps, lr=0.01, weight_decay=0.001)),
        ("adamw", lambda ps: torch.optim.AdamW(ps, lr=0.01, weight_decay=0.1)),
        ("sgd", lambda ps: torch.optim.SGD(ps, lr=0.01, momentum=0.9, weight_decay=0.001)),
    ],
)
def test_matches_torch_optim(mode, ref):
    a, b = _models()
    wd = 0.1 if mode == "adamw" else 0.001
    ours = FusedOptimizer(a.parameters(), mode=mode, lr=0.01, weight_decay=wd, momentum=0.9)
    theirs = ref(b.parameters())
    g = torch.Generator().manual_seed(1)
    for _ in range(5):
        x = torch.randn(4, 8, generator=g)
        for m, o in ((a, ours), (b, theirs)):
            o.zero_grad()
            m(x).pow(2).sum().backward()
            o.step()
    for pa, pb in zip(a.parameters(), b.parameters()):
        assert torch.allclose(pa, pb, atol=1e-6, rtol=1e-5)


def test_state_dict_round_trip_continues_identically():
    a, b = _models()
    oa = FusedOptimizer(a.parameters(), mode="adam", lr=0.01, weight_decay=0.001)
    x = torch.randn(4, 8)
    for _ in range(3):
        oa.zero_grad(); a(x).pow(2).sum().backward(); oa.step()
    b.load_state_dict(a.state_dict())
    ob = FusedOptimizer(b.parameters(), mode="adam", lr=0.01, weight_decay=0.001)
    import copy

    ob.load_state_dict(copy.deepcopy(oa.state_dict()))  # a file round trip never aliases tensors
    for m, o in ((a, oa), (b, ob)):
        o.zero_grad(); m(x).pow(2).sum().backward(); o.step()
    for pa, pb in zip(a.parameters(), b.parameters()):
        assert torch.equal(pa, pb)


def test_rejects_bad_arguments():
    p = [nn.Parameter(torch.zeros(1))]
    with pytest.raises(ValueError):
        FusedOptimizer(p, mode="rmsprop")
    with pytest.raises(ValueError):
        FusedOptimizer(p, lr=-1.0)
