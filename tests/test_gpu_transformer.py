"""Numerics of the transformer HIP kernels against PyTorch fp32 references (needs an MI355X)."""
import math

import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ht():
    from pytorch_distributed_training_amd.ops import hip_transformer as m

    return m


def _dev():
    return torch.device("cuda", 0)


def _r(t):
    return t.to(torch.bfloat16).to(torch.float32)


def _rel(a, b):
    return float((a.float() - b.float()).norm() / (b.float().norm() + 1e-12))


# (8192: wider than the register-resident kernels hold - served by the fallback path)
@pytest.mark.parametrize("rows,d", [(197 * 3, 768), (64, 1024), (50, 64), (17, 3072), (9, 8192)])
def test_layer_norm(ht, rows, d):
    torch.manual_seed(0)
    dev = _dev()
    x = (torch.randn(rows, d, device=dev) * 2 + 0.5).to(torch.bfloat16).requires_grad_(True)
    w = (torch.rand(d, device=dev) + 0.5).requires_grad_(True)
    b = torch.randn(d, device=dev).requires_grad_(True)
    y = ht.layer_norm(x, w, b, 1e-5)
    xr = x.detach().float().requires_grad_(True)
    wr, br = w.detach().clone().requires_grad_(True), b.detach().clone().requires_grad_(True)
    ref = F.layer_norm(xr, (d,), wr, br, 1e-5)
    assert _rel(y, ref) < 6e-3
    g = torch.randn_like(ref)
    y.backward(g.to(torch.bfloat16))
    ref.backward(_r(g))
    assert _rel(x.grad, xr.grad) < 1e-2
    assert _rel(w.grad, wr.grad) < 1e-2
    assert _rel(b.grad, br.grad) < 1e-2


def test_gelu_and_bias_residual(ht):
    torch.manual_seed(1)
    dev = _dev()
    x = (torch.randn(33, 256, device=dev) * 2).to(torch.bfloat16).requires_grad_(True)
    y = ht.gelu(x)
    xr = x.detach().float().requires_grad_(True)
    ref = F.gelu(xr, approximate="tanh")
    assert _rel(y, ref) < 6e-3
    y.backward(torch.ones_like(y))
    ref.backward(torch.ones_like(ref))
    assert _rel(x.grad, xr.grad) < 8e-3
    a = torch.randn(40, 128, device=dev).to(torch.bfloat16).requires_grad_(True)
    res = torch.randn(40, 128, device=dev).to(torch.bfloat16).requires_grad_(True)
    bias = torch.randn(128, device=dev).requires_grad_(True)
    out = ht.add_bias_residual(a, bias, res)
    assert _rel(out, a.float() + bias + res.float()) < 6e-3
    g = torch.randn(40, 128, device=dev).to(torch.bfloat16)
    out.backward(g)
    assert torch.equal(a.grad, g) and torch.equal(res.grad, g)
    assert _rel(bias.grad, g.float().sum(0)) < 1e-3


@pytest.mark.parametrize("d,rows", [(64, (4, 9)), (200, (3, 50)), (1024, (8, 300))])
def test_embedding(ht, d, rows):
    """Gather forward; scatter-add backward (a wave per token row) with many tokens hitting the same table rows."""
    torch.manual_seed(2)
    dev = _dev()
    w = torch.randn(100, d, device=dev).requires_grad_(True)
    ids = torch.randint(0, 100, rows, device=dev)
    out = ht.embedding(ids, w)
    assert out.shape == (*rows, d)
    assert torch.equal(out, w.detach()[ids].to(torch.bfloat16))
    g = torch.randn(*rows, d, device=dev).to(torch.bfloat16)
    out.backward(g)
    ref = torch.zeros_like(w)
    ref.index_put_((ids.reshape(-1),), g.float().reshape(-1, d), accumulate=True)
    assert _rel(w.grad, ref) < 1e-5


@pytest.mark.parametrize("b,s,h,causal", [(2, 197, 3, False), (2, 256, 4, True), (1, 130, 2, True), (3, 64, 1, False)])
def test_attention_forward_backward(ht, b, s, h, causal):
    from pytorch_distributed_training_amd.ops import reference as ref_ops

    torch.manual_seed(3)
    dev = _dev()
    d = h * 64
    qkv = torch.randn(b, s, 3 * d, device=dev).to(torch.bfloat16).requires_grad_(True)
    out = ht.attention_qkv(qkv, h, causal)
    qr = qkv.detach().float().requires_grad_(True)
    ref = ref_ops.attention_qkv(qr, h, causal)
    assert out.shape == ref.shape
    assert _rel(out, ref) < 1e-2
    g = torch.randn_like(ref)
    out.backward(g.to(torch.bfloat16))
    ref.backward(_r(g))
    dq, dk, dv = (qkv.grad.view(b, s, 3, d)[:, :, i] for i in range(3))
    rq, rk, rv = (qr.grad.view(b, s, 3, d)[:, :, i] for i in range(3))
    assert _rel(dv, rv) < 1.5e-2
    assert _rel(dk, rk) < 2e-2
    assert _rel(dq, rq) < 2e-2


@pytest.mark.parametrize("b,s,h,causal,amp", [(2, 1, 2, True, 1.0), (2, 77, 3, True, 1.0), (1, 1000, 4, False, 1.0), (2, 300, 2, True, 6.0),
                                              (2, 300, 2, False, 12.0), (1, 520, 2, True, 30.0)])
def test_attention_ragged_lengths_and_scores_hundreds_of_octaves_apart(ht, b, s, h, causal, amp):
    """Sequence ends inside a tile / inside a 128-row block, a single key, and inputs scaled so that scores reach
    +-1000: the forward kernel takes no running maximum on its common path, so the later tiles of such rows overflow the
    2^30 bound and must go through the careful path (maximum raised, output rescaled) and still match fp32."""
    from pytorch_distributed_training_amd.ops import reference as ref_ops

    torch.manual_seed(11)
    dev = _dev()
    d = h * 64
    qkv = (torch.randn(b, s, 3 * d, device=dev) * amp).to(torch.bfloat16).requires_grad_(True)
    out = ht.attention_qkv(qkv, h, causal)
    qr = qkv.detach().float().requires_grad_(True)
    ref = ref_ops.attention_qkv(qr, h, causal)
    assert torch.isfinite(out.float()).all()
    assert _rel(out, ref) < 1e-2
    g = torch.randn_like(ref)
    out.backward(g.to(torch.bfloat16))
    ref.backward(_r(g))
    assert torch.isfinite(qkv.grad.float()).all()
    floor = 1e-3 * ref.numel() ** 0.5  # a (nearly) zero reference gradient - one key - is compared absolutely
    for i, tol in ((0, 2.5e-2), (1, 2.5e-2), (2, 1.5e-2)):
        a, r = qkv.grad.view(b, s, 3, d)[:, :, i].float(), qr.grad.view(b, s, 3, d)[:, :, i]
        assert float((a - r).norm() / r.norm().clamp_min(floor)) < tol, ("dq", "dk", "dv")[i]


def test_vit_and_gpt2_tiny_train_on_gpu():
    from pytorch_distributed_training_amd import ops
    from pytorch_distributed_training_amd.models import build_model
    from pytorch_distributed_training_amd.optim import FusedOptimizer

    torch.manual_seed(4)
    dev = _dev()
    vit = build_model("vit_tiny").to(dev)
    opt = FusedOptimizer(vit.parameters(), mode="adamw", lr=3e-3, weight_decay=0.01)
    x = ops.to_device_layout(torch.randn(16, 3, 32, 32, device=dev))
    y = torch.randint(0, 10, (16,), device=dev)
    losses = []
    for _ in range(15):
        opt.zero_grad()
        l = ops.softmax_cross_entropy(vit(x), y)
        l.backward()
        opt.step()
        losses.append(float(l.detach()))
    assert losses[-1] < 0.7 * losses[0], losses
    gpt = build_model("gpt2_tiny").to(dev)
    opt = FusedOptimizer(gpt.parameters(), mode="adamw", lr=3e-3, weight_decay=0.01)
    ids = torch.randint(0, 512, (4, 64), device=dev)
    tg = torch.randint(0, 512, (4, 64), device=dev)
    losses = []
    for _ in range(15):
        opt.zero_grad()
        l = gpt(ids, tg)
        l.backward()
        opt.step()
        losses.append(float(l.detach()))
    assert abs(losses[0] - math.log(512)) < 0.7
    assert losses[-1] < 0.85 * losses[0], losses


def test_gradient_sinks_match_autograd_accumulation():
    """Kernels that add parameter gradients straight into ``.grad`` (LayerNorm, biases, embedding, linear
    weights - including the embedding matrix tied to the output projection) give the same gradients as
    returning them to autograd."""
    from pytorch_distributed_training_amd import ops
    from pytorch_distributed_training_amd.models import build_model
    from pytorch_distributed_training_amd.optim import FusedOptimizer

    dev = _dev()
    ids = torch.randint(0, 512, (4, 64), device=dev)
    tg = torch.randint(0, 512, (4, 64), device=dev)
    grads = {}
    for sink in (False, True):
        torch.manual_seed(11)
        gpt = build_model("gpt2_tiny").to(dev)
        if sink:
            assert ops.prepare_module(gpt) is gpt
            assert all(getattr(p, "_pdt_sink_ok", False) for p in gpt.parameters())
        opt = FusedOptimizer(gpt.parameters(), mode="adamw", lr=1e-3)
        opt.zero_grad()  # materialises the flat fp32 gradient buffer the sinks write into
        gpt(ids, tg).backward()
        grads[sink] = {n: p.grad.detach().clone() for n, p in gpt.named_parameters()}
    assert grads[True].keys() == grads[False].keys()
    for n in grads[True]:
        a, b = grads[True][n], grads[False][n]
        assert torch.isfinite(a).all()
        assert _rel(a, b) < 2e-3, n


@pytest.mark.parametrize("rows,d", [(197 * 8, 768), (1024, 1024), (300, 2048), (64, 4096)])
def test_skip_gradient_is_added_inside_the_layernorm_backward(ht, rows, d):
    """y = linear(layer_norm(x), residual=x) with the skip link: x.grad must equal plain autograd's sum of the
    branch gradient and the skip gradient."""
    from pytorch_distributed_training_amd import ops
    from pytorch_distributed_training_amd.ops import reference as ref

    torch.manual_seed(5)
    dev = torch.device("cuda", 0)
    ln = nn.LayerNorm(d).to(dev)
    with torch.no_grad():
        ln.weight.uniform_(0.5, 1.5)
        ln.bias.uniform_(-0.2, 0.2)
    lin = nn.Linear(d, d).to(dev)
    x0 = torch.randn(2, rows // 2, d, device=dev)
    g = torch.randn(2, rows // 2, d, device=dev).to(torch.bfloat16)

    xa = x0.to(torch.bfloat16).requires_grad_(True)
    link = ops.skip_link(xa)
    ya = ops.linear(ops.layer_norm(xa, ln.weight, ln.bias, ln.eps, link=link), lin.weight, lin.bias, residual=xa, link=link)
    ya.backward(g)
    assert link.grad is None  # consumed by the LayerNorm backward
    got = {"x": xa.grad.float(), "lnw": ln.weight.grad.clone(), "lnb": ln.bias.grad.clone(), "w": lin.weight.grad.clone()}
    for p_ in list(ln.parameters()) + list(lin.parameters()):
        p_.grad = None

    xb = x0.to(torch.bfloat16).float().requires_grad_(True)
    wr = lin.weight.detach().to(torch.bfloat16).float()
    yb = ref.linear(ref.layer_norm(xb, ln.weight, ln.bias, ln.eps), wr, lin.bias, residual=xb)
    yb.backward(g.float())
    assert _rel(ya, yb) < 1e-2
    assert _rel(got["x"], xb.grad) < 1e-2
    assert _rel(got["lnw"], ln.weight.grad) < 2e-2
    assert _rel(got["lnb"], ln.bias.grad) < 2e-2


def test_skip_link_is_bypassed_when_the_layernorm_input_needs_no_gradient(ht):
    from pytorch_distributed_training_amd import ops

    dev = torch.device("cuda", 0)
    d = 256
    ln, lin = nn.LayerNorm(d).to(dev), nn.Linear(d, d).to(dev)
    x = torch.randn(128, d, device=dev).to(torch.bfloat16)  # no grad needed
    link = ops.skip_link(x)
    y = ops.linear(ops.layer_norm(x, ln.weight, ln.bias, ln.eps, link=link), lin.weight, lin.bias, residual=x, link=link)
    y.float().sum().backward()
    assert not link.armed and link.grad is None and lin.weight.grad is not None


@pytest.mark.parametrize("rows,d,hd,res", [(197 * 16, 768, 3072, True), (4096, 1024, 4096, True), (200, 256, 512, False), (1000, 128, 512, True)])
def test_fused_gelu_mlp_matches_the_composition(ht, rows, d, hd, res):
    """Large shapes take the ring GEMM kernels (GELU, its derivative, bias gradients in the epilogues), small
    ones the register-staged kernel; both are compared with the fp32 PyTorch composition."""
    from pytorch_distributed_training_amd import ops
    from pytorch_distributed_training_amd.ops import reference as ref

    torch.manual_seed(11)
    dev = torch.device("cuda", 0)
    fc1, fc2 = nn.Linear(d, hd).to(dev), nn.Linear(hd, d).to(dev)
    x0 = torch.randn(rows, d, device=dev).to(torch.bfloat16)
    r0 = torch.randn(rows, d, device=dev).to(torch.bfloat16)
    g = torch.randn(rows, d, device=dev).to(torch.bfloat16)

    xa, ra = x0.clone().requires_grad_(True), r0.clone().requires_grad_(True)
    ya = ops.gelu_mlp(xa, fc1.weight, fc1.bias, fc2.weight, fc2.bias, residual=ra if res else None)
    ya.backward(g)
    got = [xa.grad.float(), fc1.weight.grad.clone(), fc1.bias.grad.clone(), fc2.weight.grad.clone(), fc2.bias.grad.clone()]
    if res:
        assert _rel(ra.grad, g) < 1e-6
    for p_ in list(fc1.parameters()) + list(fc2.parameters()):
        p_.grad = None

    xb, rb = x0.float().requires_grad_(True), r0.float()
    w1 = fc1.weight.detach().to(torch.bfloat16).float().requires_grad_(True)
    w2 = fc2.weight.detach().to(torch.bfloat16).float().requires_grad_(True)
    b1 = fc1.bias.detach().clone().requires_grad_(True)
    b2 = fc2.bias.detach().clone().requires_grad_(True)
    yb = ref.linear(ref.linear(xb, w1, b1, activation="gelu"), w2, b2, residual=rb if res else None)
    yb.backward(g.float())
    assert _rel(ya, yb) < 1e-2
    for a, b in zip(got, [xb.grad, w1.grad, b1.grad, w2.grad, b2.grad]):
        assert _rel(a, b) < 1.5e-2


@pytest.mark.parametrize("chunk,upstream", [(128, 1.0), (4096, 0.25)])
def test_lm_head_loss_chunked_matches_fp32_reference(chunk, upstream):
    """Output projection + cross-entropy in one operator, rows in chunks through one scratch buffer: loss, dx and dW
    against fp32 PyTorch on the bf16-rounded operands (vocabulary 50257 as in GPT-2, not a multiple of 8)."""
    from pytorch_distributed_training_amd.ops import hip

    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    M, K, V = 300, 256, 50257
    x = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16).requires_grad_(True)
    w = (torch.randn(V, K, device=dev) * 0.05).requires_grad_(True)
    y = torch.randint(0, V, (M,), device=dev)
    old = hip.LM_HEAD_CHUNK_ROWS
    hip.LM_HEAD_CHUNK_ROWS = chunk
    try:
        loss = hip.lm_head_loss(x, w, y)
        (loss * upstream).backward()
    finally:
        hip.LM_HEAD_CHUNK_ROWS = old
    xr = x.detach().float().requires_grad_(True)
    wr = w.detach().to(torch.bfloat16).float().requires_grad_(True)
    ref = F.cross_entropy(xr @ wr.t(), y)
    (ref * upstream).backward()
    assert abs(float(loss) - float(ref)) < 2e-3 * float(ref)
    rel = lambda a, b: float((a.float() - b).norm() / (b.norm() + 1e-12))
    assert rel(x.grad, xr.grad) < 2e-2
    assert rel(w.grad, wr.grad) < 2e-2
