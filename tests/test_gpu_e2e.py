"""End-to-end numerics of the benchmark models: the HIP path (bf16 compute, fused kernels) against the same model
on the CPU in fp32 through ``ops.reference`` - same weights, same batch.  Compared: logits, every parameter's
gradient (cosine), and a short SGD / AdamW loss curve.

A ResNet-50 with the default initialisation (every BatchNorm weight 1) amplifies rounding noise from block to block
in training mode: with bf16 activations the logits end up ~12 % away from the fp32 ones and the gradients of the
early layers are all but uncorrelated with the fp32 gradients (median cosine 0.15) for *any* bf16 pipeline - measured
here for stock PyTorch mixed precision (the same CPU model under ``torch.autocast(bfloat16)``) as well as for the HIP
path.  That configuration can only be checked for parity with the stock-bf16 yardstick.  The strict comparison runs
on the same network with its residual branches damped (last BatchNorm weight of every block 0.1, as in a trained
network), where bf16 noise stays noise: logits within 2 % of fp32, every one of the 161 gradients within the
yardstick's distance of the fp32 gradient.  The transformers (LayerNorm, no batch statistics) get absolute bounds."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def _cos(a, b):
    a, b = a.detach().double().cpu().reshape(-1), b.detach().double().cpu().reshape(-1)
    return float(torch.dot(a, b) / (a.norm() * b.norm() + 1e-30))


def _pair(build):
    from pytorch_distributed_training_amd import ops

    torch.manual_seed(0)
    ref = build()
    gpu = copy.deepcopy(ref).to("cuda")
    if any(p.dim() == 4 for p in gpu.parameters()):
        gpu = gpu.to(memory_format=torch.channels_last)
    ops.prepare_module(gpu)
    return ref, gpu


def _resnet50(gamma3):
    from pytorch_distributed_training_amd.models import build_model

    def build():
        m = build_model("resnet50", num_classes=1000)
        for n, mod in m.named_modules():
            if n.endswith("bn3"):
                torch.nn.init.constant_(mod.weight, gamma3)
        return m

    return build


def _three_way(gamma3):
    """fp32 CPU reference, stock-bf16 CPU yardstick and the HIP model after one forward + backward on one batch."""
    from pytorch_distributed_training_amd import ops

    torch.set_num_threads(16)
    ref, gpu = _pair(_resnet50(gamma3))
    yard = copy.deepcopy(ref)
    g = torch.Generator().manual_seed(1)
    x = torch.rand(32, 3, 224, 224, generator=g)
    y = torch.randint(0, 1000, (32,), generator=g)
    xg, yg = ops.to_device_layout(x.cuda(), torch.bfloat16), y.cuda()
    for m in (ref, yard, gpu):
        m.train()
    logits_ref = ref(x)
    ops.softmax_cross_entropy(logits_ref, y).backward()
    with torch.autocast("cpu", dtype=torch.bfloat16):
        logits_yard = yard(x)
        loss_yard = ops.softmax_cross_entropy(logits_yard.float(), y)
    loss_yard.backward()
    for p in gpu.parameters():
        p.grad = torch.zeros_like(p)
    logits_gpu = gpu(xg)
    ops.softmax_cross_entropy(logits_gpu, yg).backward()
    torch.cuda.synchronize()
    return ref, yard, gpu, (x, y, xg, yg), (logits_ref, logits_yard, logits_gpu)


def test_resnet50_default_init_is_as_close_to_fp32_as_stock_bf16():
    ref, yard, gpu, _, (logits_ref, logits_yard, logits_gpu) = _three_way(1.0)
    err_gpu, err_yard = _rel(logits_gpu.float(), logits_ref), _rel(logits_yard.float(), logits_ref)
    assert err_gpu < 1.25 * err_yard + 5e-3, (err_gpu, err_yard)
    cos_gpu = sorted(_cos(a.grad, b.grad) for a, b in zip(gpu.parameters(), ref.parameters()))
    cos_yard = sorted(_cos(a.grad, b.grad) for a, b in zip(yard.parameters(), ref.parameters()))
    # a chaotic configuration: only the distribution can be compared (medians ~0.15 for both pipelines)
    assert cos_gpu[len(cos_gpu) // 2] > cos_yard[len(cos_yard) // 2] - 0.05, (cos_gpu[80], cos_yard[80])


def test_resnet50_matches_fp32_reference_logits_grads_and_loss_curve():
    from pytorch_distributed_training_amd import ops
    from pytorch_distributed_training_amd.optim import build_optimizer

    ref, yard, gpu, (x, y, xg, yg), (logits_ref, logits_yard, logits_gpu) = _three_way(0.1)
    assert _rel(logits_gpu.float(), logits_ref) < 2e-2
    assert _rel(logits_gpu.float(), logits_ref) < 1.25 * _rel(logits_yard.float(), logits_ref) + 2e-3
    names = [n for n, _ in ref.named_parameters()]
    assert len(names) == 161
    cos_gpu = {n: _cos(pg.grad, pr.grad) for n, pg, pr in zip(names, gpu.parameters(), ref.parameters())}
    cos_yard = {n: _cos(py.grad, pr.grad) for n, py, pr in zip(names, yard.parameters(), ref.parameters())}
    worst = sorted(cos_gpu.items(), key=lambda kv: kv[1])[:5]
    assert min(cos_gpu.values()) > 0.9, f"lowest gradient cosines: {worst}"
    assert sorted(cos_gpu.values())[len(names) // 2] > 0.95
    # no gradient is further from the fp32 gradient than stock bf16's is (with a margin for run-to-run noise)
    behind = {n: (cos_gpu[n], cos_yard[n]) for n in names if (1 - cos_gpu[n]) > 1.3 * (1 - cos_yard[n]) + 0.01}
    assert not behind, f"gradients further from fp32 than stock bf16's: {behind}"
    # the layers next to the loss see almost no accumulated noise
    assert cos_gpu["fc.weight"] > 0.999 and cos_gpu["fc.bias"] > 0.999
    # BatchNorm running statistics moved the same way
    for (n, bg), (_, br) in zip(gpu.named_buffers(), ref.named_buffers()):
        if n.endswith("running_mean") or n.endswith("running_var"):
            assert _rel(bg.float(), br) < 2e-2, n

    # 10 SGD steps on the fixed batch: the loss curve follows the fp32 one within 2 %
    for m in (ref, gpu):
        for p in m.parameters():
            p.grad = None
    opt_ref = torch.optim.SGD(ref.parameters(), lr=0.01, momentum=0.5, weight_decay=1e-4)
    opt_gpu = build_optimizer(gpu.parameters(), "sgd", 0.01, 1e-4, 0.5)
    curve_ref, curve_gpu = [], []
    for _ in range(10):
        opt_ref.zero_grad()
        l = ops.softmax_cross_entropy(ref(x), y)
        l.backward()
        opt_ref.step()
        curve_ref.append(float(l.detach()))
        opt_gpu.zero_grad(set_to_none=False)
        l = ops.softmax_cross_entropy(gpu(xg), yg)
        l.backward()
        opt_gpu.step()
        curve_gpu.append(float(l.detach()))
    assert curve_ref[-1] < 0.9 * curve_ref[0] and curve_gpu[-1] < 0.9 * curve_gpu[0], (curve_gpu, curve_ref)
    for a, b in zip(curve_gpu, curve_ref):
        assert abs(a - b) < 2e-2 * max(abs(b), 1.0), (curve_gpu, curve_ref)


def test_vit_real_width_reduced_depth_matches_fp32_reference():
    from pytorch_distributed_training_amd import ops
    from pytorch_distributed_training_amd.models.vit import VisionTransformer
    from pytorch_distributed_training_amd.optim import build_optimizer

    torch.set_num_threads(16)
    # ViT-B/16 geometry (width 768, 12 heads of 64, MLP 3072, 196 + 1 tokens) with 2 of the 12 layers
    def build():
        m = VisionTransformer(image_size=224, patch=16, dim=768, depth=2, heads=12, mlp_dim=3072, num_classes=1000)
        torch.nn.init.normal_(m.head.weight, std=0.02)  # the zero-initialised head would make every other gradient zero
        return m

    ref, gpu = _pair(build)
    g = torch.Generator().manual_seed(2)
    x = torch.rand(16, 3, 224, 224, generator=g)
    y = torch.randint(0, 1000, (16,), generator=g)
    xg, yg = ops.to_device_layout(x.cuda(), torch.bfloat16), y.cuda()
    logits_ref = ref(x)
    ops.softmax_cross_entropy(logits_ref, y).backward()
    for p in gpu.parameters():
        p.grad = torch.zeros_like(p)
    logits_gpu = gpu(xg)
    ops.softmax_cross_entropy(logits_gpu, yg).backward()
    torch.cuda.synchronize()
    assert _rel(logits_gpu.float(), logits_ref) < 2e-2
    cosines = {n: _cos(pg.grad, pr.grad) for (n, pr), pg in zip(ref.named_parameters(), gpu.parameters())}
    worst = sorted(cosines.items(), key=lambda kv: kv[1])[:5]
    assert all(c > 0.99 for c in cosines.values()), f"lowest gradient cosines: {worst}"
    for m in (ref, gpu):
        for p in m.parameters():
            p.grad = None
    opt_ref = torch.optim.AdamW(ref.parameters(), lr=1e-4, weight_decay=0.05)
    opt_gpu = build_optimizer(gpu.parameters(), "adamw", 1e-4, 0.05)
    for step in range(6):
        opt_ref.zero_grad()
        lr_ = ops.softmax_cross_entropy(ref(x), y)
        lr_.backward()
        opt_ref.step()
        opt_gpu.zero_grad(set_to_none=False)
        lg = ops.softmax_cross_entropy(gpu(xg), yg)
        lg.backward()
        opt_gpu.step()
        assert abs(float(lg) - float(lr_)) < 2e-2 * max(float(lr_), 1.0), (step, float(lg), float(lr_))


def test_gpt2_real_width_reduced_depth_matches_fp32_reference():
    from pytorch_distributed_training_amd.models.gpt2 import GPT2
    from pytorch_distributed_training_amd.optim import build_optimizer

    torch.set_num_threads(16)
    # GPT-2 medium geometry (width 1024, 16 heads of 64, tied 50257-entry vocabulary, 1024 positions), 2 of 24 layers
    ref, gpu = _pair(lambda: GPT2(50257, 1024, 1024, 2, 16))
    g = torch.Generator().manual_seed(3)
    t = torch.randint(0, 50257, (2, 513), generator=g)
    ids, tgt = t[:, :-1].contiguous(), t[:, 1:].contiguous()
    loss_ref = ref(ids, tgt)
    loss_ref.backward()
    for p in gpu.parameters():
        p.grad = torch.zeros_like(p)
    loss_gpu = gpu(ids.cuda(), tgt.cuda())
    loss_gpu.backward()
    torch.cuda.synchronize()
    assert abs(float(loss_gpu) - float(loss_ref)) < 1e-2 * float(loss_ref)
    cosines = {n: _cos(pg.grad, pr.grad) for (n, pr), pg in zip(ref.named_parameters(), gpu.parameters())}
    worst = sorted(cosines.items(), key=lambda kv: kv[1])[:5]
    assert all(c > 0.99 for c in cosines.values()), f"lowest gradient cosines: {worst}"
    for m in (ref, gpu):
        for p in m.parameters():
            p.grad = None
    opt_ref = torch.optim.AdamW(ref.parameters(), lr=3e-4, weight_decay=0.1)
    opt_gpu = build_optimizer(gpu.parameters(), "adamw", 3e-4, 0.1)
    for step in range(6):
        opt_ref.zero_grad()
        lr_ = ref(ids, tgt)
        lr_.backward()
        opt_ref.step()
        opt_gpu.zero_grad(set_to_none=False)
        lg = gpu(ids.cuda(), tgt.cuda())
        lg.backward()
        opt_gpu.step()
        assert abs(float(lg) - float(lr_)) < 2e-2 * max(float(lr_), 1.0), (step, float(lg), float(lr_))


def test_check_finite_sees_an_inf_that_batchnorm_and_relu_would_swallow():
    """--check-finite on the GPU path: a weight poisoned with Inf produces NaN statistics in the first BatchNorm; ReLU
    maps the NaN activations to zero and the loss stays finite, so only the kernels' non-finite flag can stop the run."""
    import pytest as _pytest

    from pytorch_distributed_training_amd.train import TrainConfig, Trainer

    cfg = TrainConfig(model="resnet18", dataset="synthetic", dataset_length=64, batch_size=16, num_workers=0,
                      learning_rate=0.01, optimizer="sgd", progress=False, log_every=1, check_finite=True, hip_graph=False,
                      device_data=True, steps=3)
    tr = Trainer(cfg)
    with torch.no_grad():
        tr.raw_model.layer1[0].conv1.weight[0, 0, 0, 0] = float("inf")
    with _pytest.raises(FloatingPointError):
        tr.fit()
    tr.close()
    ok = Trainer(TrainConfig(model="resnet18", dataset="synthetic", dataset_length=64, batch_size=16, num_workers=0,
                             learning_rate=0.01, optimizer="sgd", progress=False, log_every=1, check_finite=True,
                             hip_graph=False, device_data=True, steps=3))
    assert ok.fit().steps == 3  # and a healthy run is not stopped
    ok.close()


@pytest.mark.parametrize("name", ["resnet18", "gpt2_tiny"])
def test_deterministic_mode_is_bit_stable_across_runs(name):
    """PDT_DETERMINISTIC: two runs of the same three training steps end with bit-identical parameters (weight
    gradients through per-split slabs, BatchNorm sums through single-writer reductions, embedding gradients through
    sorted segment sums); the mode's results agree with the ordinary mode's to rounding."""
    from pytorch_distributed_training_amd import ops
    from pytorch_distributed_training_amd.models import build_model
    from pytorch_distributed_training_amd.optim import build_optimizer

    dev = torch.device("cuda", 0)

    def run(det: bool):
        ops.set_deterministic(det)
        try:
            torch.manual_seed(0)
            if name == "resnet18":
                m = build_model("resnet18", 10, 64).to(dev).to(memory_format=torch.channels_last)
                x = ops.to_device_layout(torch.randn(32, 3, 64, 64, device=dev))
                y = torch.randint(0, 10, (32,), device=dev)
            else:
                m = build_model("gpt2_tiny").to(dev)
                t = torch.randint(0, m.vocab_size, (8, 65), device=dev)
                t[:, :8] = 7  # many tokens share an id: their embedding gradients meet in one table row
                x, y = t[:, :-1].contiguous(), t[:, 1:].contiguous()
            raw = ops.prepare_module(m)
            opt = build_optimizer([p for p in raw.parameters() if p.requires_grad], "sgd", 0.05, 1e-4, 0.9)
            opt.zero_grad(set_to_none=False)
            first = None
            losses = []
            for _ in range(3):
                loss = m(x, y) if name != "resnet18" else ops.softmax_cross_entropy(m(x), y)
                loss.backward()
                losses.append(float(loss.detach()))
                if first is None:
                    first = [p.grad.detach().float().clone() for p in raw.parameters()]
                opt.step()
                opt.zero_grad(set_to_none=False)
            torch.cuda.synchronize()
            return torch.cat([p.detach().reshape(-1).float() for p in raw.parameters()]).clone(), losses, first
        finally:
            ops.set_deterministic(False)

    a, la, ga = run(True)
    b, lb, gb = run(True)
    assert torch.equal(a, b) and la == lb, float((a - b).abs().max())
    assert all(torch.equal(u, v) for u, v in zip(ga, gb))
    # The ordinary mode computes the same thing up to the order of its floating-point sums.  A default-initialised
    # ResNet at a small batch amplifies last-bit differences of the BatchNorm constants layer by layer (see
    # test_resnet50_default_init_is_as_close_to_fp32_as_stock_bf16), so there only the direction of the first step's
    # gradients and the layers next to the loss are compared; the transformer (no BatchNorm) agrees to bf16 noise.
    _, lc, gc = run(False)
    for u, v in zip(ga, gc):
        if name == "resnet18":
            assert _cos(u, v) > 0.9
        else:
            assert float((u - v).norm()) <= 3e-2 * float(v.norm()) + 1e-6
    assert float((ga[-1] - gc[-1]).norm()) <= 2e-2 * float(gc[-1].norm()) + 1e-6  # the last layer's bias / table
    # the first loss is the same forward pass; three steps on, the chaotic ResNet has drifted (the ordinary mode does not
    # even repeat itself there), the transformer has not
    assert abs(la[0] - lc[0]) < 1e-2 * max(1.0, abs(lc[0]))
    assert abs(la[-1] - lc[-1]) < (0.25 if name == "resnet18" else 5e-2) * max(1.0, abs(lc[-1]))
