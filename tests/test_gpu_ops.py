"""Numerics of the HIP kernels against plain PyTorch fp32 references (needs an MI355X)."""
import math

import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from pytorch_distributed_training_amd.ops import hip as h

    return h


def _dev():
    return torch.device("cuda", 0)


def _bf16_round(t):
    return t.to(torch.bfloat16).to(torch.float32)


def _rel_err(a, b):
    return float((a.float() - b.float()).norm() / (b.float().norm() + 1e-12))


CONV_CASES = [
    # (batch, cin, cout, k, stride, pad, hw)
    (8, 64, 64, 1, 1, 0, 56),
    (8, 64, 64, 3, 1, 1, 56),
    (8, 64, 256, 1, 1, 0, 56),
    (8, 256, 64, 1, 1, 0, 56),
    (4, 128, 128, 3, 2, 1, 56),
    (4, 256, 512, 1, 2, 0, 56),
    (8, 256, 256, 3, 1, 1, 14),
    (8, 512, 512, 3, 1, 1, 7),
    (3, 1024, 2048, 1, 2, 0, 14),
    (5, 64, 128, 3, 2, 1, 15),  # odd sizes: edge tiles, partial parity classes
    (2, 3, 64, 7, 2, 3, 64),    # stem (patch-matrix path)
]


@pytest.mark.parametrize("nb,cin,cout,k,stride,pad,hw", CONV_CASES)
def test_conv_forward_backward(hip, nb, cin, cout, k, stride, pad, hw):
    torch.manual_seed(0)
    dev = _dev()
    x = torch.randn(nb, cin, hw, hw, device=dev)
    w = (torch.randn(cout, cin, k, k, device=dev) / math.sqrt(cin * k * k)).contiguous(memory_format=torch.channels_last)
    w = nn.Parameter(w)
    xb = hip.to_device_layout(x).requires_grad_(cin % 64 == 0)
    y, stats = hip.conv2d_stats(xb, w, stride, pad)
    # reference: same bf16-rounded operands, fp32 math
    xr = _bf16_round(x).requires_grad_(True)
    wr = _bf16_round(w.detach()).requires_grad_(True)
    yr = F.conv2d(xr, wr, None, stride, pad)
    assert y.shape == yr.shape and y.dtype == torch.bfloat16
    assert _rel_err(y, yr) < 6e-3
    s = stats[:-1].sum(dim=0)  # (the last slot carries the pivot of the sums: zero here)
    yf = y.float()
    assert torch.allclose(s[0], yf.sum(dim=(0, 2, 3)), rtol=2e-3, atol=2e-2 * math.sqrt(yf[:, 0].numel()))
    assert torch.allclose(s[1], (yf * yf).sum(dim=(0, 2, 3)), rtol=2e-3, atol=1e-2)
    g = torch.randn_like(yr)
    gb = g.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    y.backward(gb)
    yr.backward(_bf16_round(g))
    assert _rel_err(w.grad, wr.grad) < 8e-3
    if cin % 64 == 0:
        assert _rel_err(xb.grad, xr.grad) < 8e-3


RING_CASES = [
    # (batch, cin, cout, k, stride, pad, hw): more 256x128 tiles than CUs, rows and channels that do not
    # fill the last tile, several taps, strided data gradient
    (24, 128, 192, 1, 1, 0, 56),
    (11, 64, 320, 3, 1, 1, 37),
    (9, 256, 384, 3, 2, 1, 45),
]


@pytest.mark.parametrize("tile", [0, 3, 6, 13, 16])
@pytest.mark.parametrize("nb,cin,cout,k,stride,pad,hw", RING_CASES)
def test_conv_every_wide_tile_variant(hip, monkeypatch, tile, nb, cin, cout, k, stride, pad, hw):
    """Each wide-N kernel variant (one tile per workgroup, and the persistent LDS-DMA ring) on its own."""
    monkeypatch.setattr(hip, "_TILE_WIDE", tile)
    test_conv_forward_backward(hip, nb, cin, cout, k, stride, pad, hw)


@pytest.mark.parametrize("tile", [1, 14, 15])
def test_conv_narrow_tile_variants(hip, monkeypatch, tile):
    """Layers with at most 64 output channels: one 128x64 tile per workgroup, and the 256x64 and 512x64 rings."""
    monkeypatch.setattr(hip, "_TILE_NARROW", tile)
    test_conv_forward_backward(hip, 9, 64, 64, 3, 1, 1, 37)
    test_conv_forward_backward(hip, 6, 256, 64, 1, 1, 0, 56)
    # the 3-channel stem ("packed rows": a K-step spans two kernel rows), odd size so the last tile is partial and the
    # last image's last output row reaches the spare row behind the padded input
    test_conv_forward_backward(hip, 3, 3, 64, 7, 2, 3, 61)
    if tile == 15:
        # 3x3 / stride 1 / 64 -> 64 channels with 24 <= W <= 63 runs on the kernel that shares the input rows between the
        # taps: the narrowest and widest images it takes, the ResNet size, and enough tiles per workgroup (more than 256
        # tiles) for the circular row buffer to wrap several times
        test_conv_forward_backward(hip, 7, 64, 64, 3, 1, 1, 24)
        test_conv_forward_backward(hip, 3, 64, 64, 3, 1, 1, 63)
        test_conv_forward_backward(hip, 64, 64, 64, 3, 1, 1, 56)
        test_conv_forward_backward(hip, 5, 64, 64, 3, 1, 1, 23)  # (one too narrow: the tap-table kernel)
        # the ImageNet stem on the kernel that streams the input rows through LDS once: several tiles per workgroup, so
        # the circular input window wraps, and image boundaries inside a workgroup's range
        test_conv_forward_backward(hip, 40, 3, 64, 7, 2, 3, 224)


@pytest.mark.parametrize("cin,cout,k,stride,pad,h,w", [(64, 64, 3, 1, 1, 40, 56), (64, 64, 3, 1, 1, 57, 31), (128, 128, 3, 1, 1, 19, 28),
                                                       (3, 64, 7, 2, 3, 160, 224), (3, 64, 7, 2, 3, 224, 150)])
def test_shared_rows_kernels_on_images_that_are_not_square(hip, monkeypatch, cin, cout, k, stride, pad, h, w):
    """The kernels that keep one copy of the input rows in LDS decode a row index into (image, row, column) themselves
    (forward: which taps leave the image, or where a pixel's kernel rows start; weight gradient: the padded pixel
    numbering): heights and widths that differ."""
    monkeypatch.setattr(hip, "_TILE_NARROW", 15)
    torch.manual_seed(3)
    dev = _dev()
    nb = 37 if cin >= 64 else 9
    x = torch.randn(nb, cin, h, w, device=dev)
    wt = nn.Parameter((torch.randn(cout, cin, k, k, device=dev) / math.sqrt(cin * k * k)).contiguous(memory_format=torch.channels_last))
    xb = hip.to_device_layout(x).requires_grad_(cin % 64 == 0)
    y, stats = hip.conv2d_stats(xb, wt, stride, pad)
    xr = _bf16_round(x).requires_grad_(True)
    wr = _bf16_round(wt.detach()).requires_grad_(True)
    yr = F.conv2d(xr, wr, None, stride, pad)
    assert y.shape == yr.shape
    assert _rel_err(y, yr) < 6e-3
    yf = y.float()
    ssum = stats[:-1].sum(dim=0)
    assert torch.allclose(ssum[0], yf.sum(dim=(0, 2, 3)), rtol=2e-3, atol=2e-2 * math.sqrt(yf[:, 0].numel()))
    g = torch.randn_like(yr)
    y.backward(g.to(torch.bfloat16).contiguous(memory_format=torch.channels_last))
    yr.backward(_bf16_round(g))
    assert _rel_err(wt.grad, wr.grad) < 8e-3
    if cin % 64 == 0:
        assert _rel_err(xb.grad, xr.grad) < 8e-3


@pytest.mark.parametrize("nb,hw", [(24, 28), (40, 17), (23, 29)])
def test_wgrad_of_the_128_channel_3x3_layers_with_shared_input_rows(hip, monkeypatch, nb, hw):
    """128 -> 128 channels, 3x3, stride 1: the weight-gradient kernel that reduces over padded pixel indices and serves all
    nine taps from one copy of the input rows (two workgroups per pixel range, 64 output channels each)."""
    monkeypatch.setattr(hip, "WGRAD_HALO", 1)
    test_conv_forward_backward(hip, nb, 128, 128, 3, 1, 1, hw)


@pytest.mark.parametrize("tile", [13, 16])
def test_linear_ring_variant_with_bias(hip, monkeypatch, tile):
    monkeypatch.setattr(hip, "_TILE_WIDE", tile)
    test_linear_forward_backward(hip, 3000, 256, 1000, True)


@pytest.mark.parametrize("tile,pp", [(10, 0), (11, 0), (12, 0), (16, 0), (16, 1)])
@pytest.mark.parametrize("m,n,k", [(4133, 520, 328), (16384, 1024, 768), (700, 256, 256)])
def test_weight_gradient_ring_tiles_against_fp32(hip, monkeypatch, tile, pp, m, n, k):
    """dW = dY^T X and the bias gradient (column sums of dY) from every ring tile of the weight-gradient kernel,
    the 256x256 tile on both of its schedules (pp = 1: ping-pong), on ragged row counts and partial tiles."""
    monkeypatch.setattr(hip, "WGRAD_RING", tile)
    monkeypatch.setattr(hip, "WGRAD_PP", pp)
    torch.manual_seed(5)
    dev = _dev()
    dy = torch.randn(m, n, device=dev).to(torch.bfloat16)
    x = torch.randn(m, k, device=dev).to(torch.bfloat16)
    dw = torch.zeros(n, k, device=dev)
    db = torch.zeros(n, device=dev)
    assert hip._launch_wgrad(dy, x, dw, m, n, n, k, [(0, 0, 0)], k, 1, 1, 1, 1, 1, dbias=db)
    assert _rel_err(dw, dy.float().t() @ x.float()) < 2e-3
    assert _rel_err(db, dy.float().sum(0)) < 2e-3


@pytest.mark.parametrize("m,k,n,wide", [(4096, 1024, 1024, None), (197 * 16, 768, 768, None), (3000, 256, 1024, 13), (3000, 256, 1024, 16), (500, 3072, 768, 0)])
def test_linear_with_bias_and_residual_in_the_epilogue(hip, monkeypatch, m, k, n, wide):
    """x W^T + bias + residual from one GEMM; the bias gradient comes out of the weight-gradient kernel when it
    is a ring kernel (large shapes) and out of the column-sum kernel otherwise."""
    if wide is not None:
        monkeypatch.setattr(hip, "_TILE_WIDE", wide)
    torch.manual_seed(3)
    dev = _dev()
    x = torch.randn(m, k, device=dev)
    res = torch.randn(m, n, device=dev)
    lin = nn.Linear(k, n).to(dev)
    xb = x.to(torch.bfloat16).requires_grad_(True)
    rb = res.to(torch.bfloat16).requires_grad_(True)
    out = hip.linear(xb, lin.weight, lin.bias, None, rb)
    xr = _bf16_round(x).requires_grad_(True)
    rr = _bf16_round(res).requires_grad_(True)
    wr = _bf16_round(lin.weight.detach()).requires_grad_(True)
    br = lin.bias.detach().clone().requires_grad_(True)
    ref = F.linear(xr, wr, br) + rr
    assert _rel_err(out, ref) < 8e-3
    g = torch.randn_like(ref)
    out.backward(g.to(torch.bfloat16))
    ref.backward(_bf16_round(g))
    assert _rel_err(xb.grad, xr.grad) < 8e-3
    assert _rel_err(rb.grad, rr.grad) < 1e-6
    assert _rel_err(lin.weight.grad, wr.grad) < 8e-3
    assert _rel_err(lin.bias.grad, br.grad) < 8e-3


@pytest.mark.parametrize("m,k,n,bias", [(256, 2048, 1000, True), (64, 512, 10, True), (300, 768, 3072, False), (197 * 4, 768, 768, True)])
def test_linear_forward_backward(hip, m, k, n, bias):
    torch.manual_seed(1)
    dev = _dev()
    x = torch.randn(m, k, device=dev)
    lin = nn.Linear(k, n, bias=bias).to(dev)
    xb = x.to(torch.bfloat16).requires_grad_(True)
    out = hip.linear(xb, lin.weight, lin.bias)
    xr = _bf16_round(x).requires_grad_(True)
    wr = _bf16_round(lin.weight.detach()).requires_grad_(True)
    br = lin.bias.detach().clone().requires_grad_(True) if bias else None
    ref = F.linear(xr, wr, br)
    assert out.shape == ref.shape
    assert _rel_err(out, ref) < 6e-3
    g = torch.randn_like(ref)
    out.backward(g.to(torch.bfloat16))
    ref.backward(_bf16_round(g))
    assert _rel_err(xb.grad, xr.grad) < 8e-3
    assert _rel_err(lin.weight.grad, wr.grad) < 8e-3
    if bias:
        assert _rel_err(lin.bias.grad, br.grad) < 8e-3


def _ref_bn(y, bn):
    return F.batch_norm(y, None, None, bn.weight, bn.bias, True, 0.1, bn.eps)


@pytest.mark.parametrize("nb,c,hw", [(8, 64, 56), (4, 256, 14), (6, 2048, 7)])
@pytest.mark.parametrize("variant", ["act", "add", "proj"])
def test_bn_family_forward_backward(hip, nb, c, hw, variant):
    torch.manual_seed(2)
    dev = _dev()
    conv = nn.Conv2d(64, c, 1, bias=False).to(dev).to(memory_format=torch.channels_last)
    x = hip.to_device_layout(torch.randn(nb, 64, hw, hw, device=dev))
    bn = nn.BatchNorm2d(c).to(dev)
    bn2 = nn.BatchNorm2d(c).to(dev)
    with torch.no_grad():
        for b in (bn, bn2):
            b.weight.uniform_(0.5, 1.5)
            b.bias.uniform_(-0.5, 0.5)
    y, st = hip.conv2d_stats(x, conv.weight, 1, 0)
    y = y.detach().requires_grad_(True)
    ident = hip.to_device_layout(torch.randn(nb, c, hw, hw, device=dev)).requires_grad_(True)
    yr = y.detach().float().requires_grad_(True)
    ir = ident.detach().float().requires_grad_(True)
    rbn = nn.BatchNorm2d(c).to(dev)
    rbn2 = nn.BatchNorm2d(c).to(dev)
    rbn.load_state_dict(bn.state_dict())
    rbn2.load_state_dict(bn2.state_dict())
    if variant == "act":
        out = hip.bn_act(y, st, bn, True)
        ref = F.relu(_ref_bn(yr, rbn))
    elif variant == "add":
        out = hip.bn_add_relu(y, st, bn, ident)
        ref = F.relu(_ref_bn(yr, rbn) + ir)
    else:
        # the shortcut statistics are those of `ident` itself
        ist = torch.zeros_like(st)
        idf = ident.detach().float()
        ist[0, 0] = idf.sum(dim=(0, 2, 3))
        ist[0, 1] = (idf * idf).sum(dim=(0, 2, 3))
        out = hip.bn_add_relu(y, st, bn, ident, ist, bn2)
        ref = F.relu(_ref_bn(yr, rbn) + _ref_bn(ir, rbn2))
    assert _rel_err(out, ref) < 1e-2
    # running statistics were updated with momentum 0.1 and the unbiased variance
    m = yr.detach().mean(dim=(0, 2, 3))
    v = yr.detach().var(dim=(0, 2, 3), unbiased=True)
    assert torch.allclose(bn.running_mean, 0.1 * m, atol=2e-3, rtol=1e-2)
    assert torch.allclose(bn.running_var, 0.9 + 0.1 * v, atol=2e-3, rtol=1e-2)
    assert int(bn.num_batches_tracked) == 1
    g = torch.randn_like(ref)
    out.backward(g.to(torch.bfloat16).contiguous(memory_format=torch.channels_last))
    ref.backward(_bf16_round(g))
    assert _rel_err(y.grad, yr.grad) < 2e-2
    assert _rel_err(bn.weight.grad, rbn.weight.grad) < 2e-2
    assert _rel_err(bn.bias.grad, rbn.bias.grad) < 2e-2
    if variant == "add":
        assert _rel_err(ident.grad, ir.grad) < 1e-2
    if variant == "proj":
        assert _rel_err(ident.grad, ir.grad) < 2e-2
        assert _rel_err(bn2.weight.grad, rbn2.weight.grad) < 2e-2


@pytest.mark.parametrize("cout,hw", [(64, 28), (256, 14)])  # one-tile-per-workgroup kernel / persistent ring kernel
def test_batch_statistics_of_a_channel_with_mean_far_above_its_spread(hip, cout, hw):
    """mean / std = 10^3: E[y^2] - E[y]^2 in fp32 would cancel to noise (or to a negative number clamped to 0, i.e.
    rstd = 1 / sqrt(eps)).  The sums are taken about a pivot that follows the batch mean, so from the second step on
    the variance is accurate; the first step (pivot 0) must at least not be wildly off for moderate ratios."""
    torch.manual_seed(3)
    dev = _dev()
    nb, cin = 16, 64
    x = torch.randn(nb, cin, hw, hw, device=dev) * 0.5
    x[:, 0] = 1.0  # a constant input channel: its weight sets the mean of an output channel
    w = torch.randn(cout, cin, 1, 1, device=dev) * 0.25
    w[:, 0] = 0.0
    w[1, 0] = 1000.0  # channel 1: mean 1000, spread about 1
    w[2, 0] = 100.0   # channel 2: mean 100
    w = nn.Parameter(w.contiguous(memory_format=torch.channels_last))
    xb = hip.to_device_layout(x)
    bn = nn.BatchNorm2d(cout).to(dev)
    count = nb * hw * hw
    errs = []
    for step in range(3):
        y, st = hip.conv2d_stats(xb, w, 1, 0)
        consts = hip._bn_consts(st, bn, count, y.device)
        yd = y.detach().double()
        mean = yd.mean(dim=(0, 2, 3))
        var = yd.var(dim=(0, 2, 3), unbiased=False)
        assert float(mean[1]) > 900 and float(var[1].sqrt()) < 5  # the case under test exists
        assert torch.allclose(consts[0].double(), mean, rtol=1e-5, atol=1e-4), step
        got_var = 1.0 / consts[1].double() ** 2 - bn.eps
        errs.append(((got_var - var).abs() / var).cpu())
    # pivot = the previous step's mean: every channel accurate, the extreme ones included
    assert float(errs[1].max()) < 2e-3 and float(errs[2].max()) < 2e-3, (errs[1].max(), errs[2].max())
    # first step (pivot 0): ordinary channels and the mean / std = 100 channel are fine already
    ordinary = torch.ones(cout, dtype=torch.bool)
    ordinary[1] = False
    assert float(errs[0][ordinary].max()) < 2e-2, errs[0][ordinary].max()
    # the pivot travels with the sums: a copy of the statistics tensor (which drops the Python-side link to the
    # pivot buffer) still finalizes to the same constants
    y, st = hip.conv2d_stats(xb, w, 1, 0)
    a = hip._bn_consts(st.clone(), nn.BatchNorm2d(cout).to(dev), count, y.device)
    b = hip._bn_consts(st, nn.BatchNorm2d(cout).to(dev), count, y.device)
    assert torch.equal(a, b)


def test_non_finite_statistics_raise_the_flag_that_check_finite_reads(hip):
    """BatchNorm + ReLU turns a NaN into 0 (max(NaN, 0) = 0), so an Inf that enters a convolution never reaches the
    loss; the statistics' finalize raises a device-side flag instead."""
    dev = _dev()
    torch.manual_seed(0)
    x = torch.randn(4, 64, 8, 8, device=dev)
    w = nn.Parameter((torch.randn(64, 64, 1, 1, device=dev) * 0.1).contiguous(memory_format=torch.channels_last))
    bn = nn.BatchNorm2d(64).to(dev)
    hip.nonfinite_seen(dev)  # clear
    y, st = hip.conv2d_stats(hip.to_device_layout(x), w, 1, 0)
    out = hip.bn_act(y, st, bn, True)
    assert torch.isfinite(out.float()).all() and not hip.nonfinite_seen(dev)
    x[1, 3, 2, 2] = float("inf")
    y, st = hip.conv2d_stats(hip.to_device_layout(x), w, 1, 0)
    out = hip.bn_act(y, st, bn, True)
    torch.cuda.synchronize()
    assert hip.nonfinite_seen(dev), "the Inf went unnoticed"
    assert not hip.nonfinite_seen(dev)  # reading clears


def test_activation_of_more_than_2_31_elements(hip):
    """Convolution (forward, data gradient, weight gradient) and BatchNorm + ReLU (forward, backward) on an activation
    whose element offsets do not fit a signed 32-bit integer: 11 000 images of 64 x 56 x 56 = 2.2e9 elements, the size
    of a ResNet-50 stage-1 tensor at a batch of about 2 750 per channel-quarter.  Checked against fp32 matrix products
    on the same bf16-rounded operands; the rows compared include the very last ones (highest offsets)."""
    dev = _dev()
    free, _ = torch.cuda.mem_get_info()
    if free < 80 * 2**30:
        pytest.skip("needs about 70 GiB of free device memory")
    torch.manual_seed(11)
    nb, c, hw = 11000, 64, 56
    M = nb * hw * hw
    assert M * c > 2**31
    x = torch.empty((nb, hw, hw, c), dtype=torch.bfloat16, device=dev)
    for i in range(0, nb, 1000):  # filled in pieces: no fp32 copy of the whole tensor
        x[i:i + 1000] = torch.randn((min(1000, nb - i), hw, hw, c), device=dev).to(torch.bfloat16)
    xb = x.permute(0, 3, 1, 2).requires_grad_(True)  # logical NCHW, physically NHWC
    w = nn.Parameter((torch.randn(c, c, 1, 1, device=dev) / 8).contiguous(memory_format=torch.channels_last))
    bn = nn.BatchNorm2d(c).to(dev)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.uniform_(-0.5, 0.5)
    y, st = hip.conv2d_stats(xb, w, 1, 0)
    out = hip.bn_act(y, st, bn, True)
    g = torch.empty_like(out)
    gp = g.permute(0, 2, 3, 1)
    for i in range(0, nb, 1000):
        gp[i:i + 1000] = torch.randn((min(1000, nb - i), hw, hw, c), device=dev).to(torch.bfloat16)
    out.backward(g)
    torch.cuda.synchronize()
    # ---- fp32 reference, computed on slices (head, a middle piece straddling offset 2^31, tail) and globally for the
    # reductions
    wr = _bf16_round(w.detach()).view(c, c)
    xf = x.view(M, c)
    yk = y.permute(0, 2, 3, 1).reshape(M, c)
    mean = torch.zeros(c, dtype=torch.float64, device=dev)
    sq = torch.zeros(c, dtype=torch.float64, device=dev)
    step = 4_000_000
    for i in range(0, M, step):
        yy = yk[i:i + step].double()
        mean += yy.sum(0)
        sq += (yy * yy).sum(0)
    mean /= M
    var = sq / M - mean * mean
    rstd = 1.0 / torch.sqrt(var + bn.eps)
    gamma, beta = bn.weight.detach().double(), bn.bias.detach().double()
    gk = g.permute(0, 2, 3, 1).reshape(M, c)
    ok = out.detach().permute(0, 2, 3, 1).reshape(M, c)
    # global sums of BatchNorm's backward
    sdz = torch.zeros(c, dtype=torch.float64, device=dev)
    sdzx = torch.zeros(c, dtype=torch.float64, device=dev)
    for i in range(0, M, step):
        yy = yk[i:i + step].double()
        xhat = (yy - mean) * rstd
        dz = gk[i:i + step].double() * ((xhat * gamma + beta) > 0)
        sdz += dz.sum(0)
        sdzx += (dz * xhat).sum(0)
    assert _rel_err(bn.bias.grad.double(), sdz) < 2e-2 and _rel_err(bn.weight.grad.double(), sdzx) < 2e-2
    dw_ref = torch.zeros(c, c, dtype=torch.float64, device=dev)
    half = 2**31 // c
    for lo in (0, half - 3000, M - 6000):
        sl = slice(lo, lo + 6000)
        yr = xf[sl].float() @ wr.t()
        assert _rel_err(yk[sl], yr) < 6e-3, lo
        xhat = (yk[sl].double() - mean) * rstd
        z = xhat * gamma + beta
        assert _rel_err(ok[sl], torch.relu(z)) < 1e-2, lo
        dz = gk[sl].double() * (z > 0)
        dy = gamma * rstd * (dz - sdz / M - xhat * sdzx / M)
        dx = dy.float() @ wr
        assert _rel_err(xb.grad.permute(0, 2, 3, 1).reshape(M, c)[sl], dx) < 2e-2, lo
    for i in range(0, M, step):
        yy = yk[i:i + step].double()
        xhat = (yy - mean) * rstd
        dz = gk[i:i + step].double() * ((xhat * gamma + beta) > 0)
        dy = (gamma * rstd * (dz - sdz / M - xhat * sdzx / M)).to(torch.bfloat16).double()
        dw_ref += dy.t() @ xf[i:i + step].double()
    assert _rel_err(w.grad.view(c, c).double(), dw_ref) < 2e-2


def test_bn_eval_uses_running_stats(hip):
    dev = _dev()
    bn = nn.BatchNorm2d(64).to(dev).eval()
    with torch.no_grad():
        bn.running_mean.uniform_(-1, 1)
        bn.running_var.uniform_(0.5, 2)
    y = hip.to_device_layout(torch.randn(4, 64, 8, 8, device=dev))
    out = hip.bn_act(y, None, bn, True)
    ref = F.relu(bn(y.float()))
    assert _rel_err(out, ref) < 1e-2


@pytest.mark.parametrize("c,h,w", [(64, 30, 30), (64, 29, 31), (24, 17, 16), (128, 9, 9)])
def test_stem_tail_maxpool(hip, c, h, w):
    """BN + ReLU + 3x3/2 max-pool, forward and backward (sums from pooled-size tensors, quad-wise gather), on even
    and odd image sizes and with a channel count whose per-thread constants cannot be hoisted (24)."""
    torch.manual_seed(3)
    dev = _dev()
    conv = nn.Conv2d(64, c, 1, bias=False).to(dev).to(memory_format=torch.channels_last)
    x = hip.to_device_layout(torch.randn(4, 64, h, w, device=dev))
    bn = nn.BatchNorm2d(c).to(dev)
    y, st = hip.conv2d_stats(x, conv.weight, 1, 0)
    y = y.detach().requires_grad_(True)
    out = hip.bn_relu_maxpool(y, st, bn)
    yr = y.detach().float().requires_grad_(True)
    rbn = nn.BatchNorm2d(c).to(dev)
    ref = F.max_pool2d(F.relu(_ref_bn(yr, rbn)), 3, 2, 1)
    assert out.shape == ref.shape
    assert _rel_err(out, ref) < 1e-2
    g = torch.randn_like(ref)
    out.backward(g.to(torch.bfloat16).contiguous(memory_format=torch.channels_last))
    ref.backward(_bf16_round(g))
    assert _rel_err(y.grad, yr.grad) < 3e-2


def test_global_avgpool(hip):
    dev = _dev()
    a = hip.to_device_layout(torch.randn(6, 128, 7, 7, device=dev)).requires_grad_(True)
    out = hip.global_avgpool(a)
    ref = a.detach().float().mean(dim=(2, 3))
    assert _rel_err(out, ref) < 5e-3
    out.backward(torch.ones_like(out))
    assert torch.allclose(a.grad.float(), torch.full_like(a.grad.float(), 1 / 49), rtol=1e-2)


@pytest.mark.parametrize("rows,classes,dtype", [(256, 1000, torch.bfloat16), (32, 10, torch.float32), (64, 50257, torch.bfloat16)])
def test_softmax_cross_entropy(hip, rows, classes, dtype):
    torch.manual_seed(4)
    dev = _dev()
    logits = (3 * torch.randn(rows, classes, device=dev)).to(dtype).requires_grad_(True)
    labels = torch.randint(0, classes, (rows,), device=dev)
    loss = hip.softmax_cross_entropy(logits, labels)
    lr = logits.detach().float().requires_grad_(True)
    ref = F.cross_entropy(lr, labels)
    assert abs(float(loss) - float(ref)) < 2e-3 * max(1.0, abs(float(ref)))
    (2.0 * loss).backward()
    (2.0 * ref).backward()
    assert _rel_err(logits.grad, lr.grad) < (1e-2 if dtype == torch.bfloat16 else 1e-5)


@pytest.mark.parametrize("rows,classes,ld,scale", [(96, 50257, 50304, 1.0), (96, 50257, 50304, 2.0), (64, 32000, 32000, 1.0), (40, 2048, 2048, 0.5), (16, 65536, 65536, 1.0)])
def test_one_pass_cross_entropy_for_long_rows(hip, rows, classes, ld, scale):
    """bf16 rows of more than 1024 classes with an 8-aligned stride: loss and gradient from one kernel that holds
    the row in registers; the stride's padding columns hold garbage that must be ignored and get zero gradient."""
    torch.manual_seed(6)
    dev = _dev()
    buf = (3 * torch.randn(rows, ld, device=dev)).to(torch.bfloat16)
    base = buf.clone().requires_grad_(True)
    logits = base[:, :classes]
    labels = torch.randint(0, classes, (rows,), device=dev)
    labels[0] = classes - 1
    loss = hip.softmax_cross_entropy(logits, labels)
    lr = buf[:, :classes].float().requires_grad_(True)
    ref = F.cross_entropy(lr, labels)
    assert abs(float(loss) - float(ref)) < 2e-3 * max(1.0, abs(float(ref)))
    (scale * loss).backward()
    (scale * ref).backward()
    assert _rel_err(base.grad[:, :classes], lr.grad) < 1e-2
    assert float(base.grad[:, classes:].abs().sum()) == 0.0
    with torch.no_grad():  # inference: no gradient buffer is produced
        assert abs(float(hip.softmax_cross_entropy(buf[:, :classes], labels)) - float(ref)) < 2e-3 * max(1.0, abs(float(ref)))


@pytest.mark.parametrize("mode", ["adam", "adamw", "sgd"])
def test_multi_tensor_optimizer_matches_cpu_definition(mode):
    from pytorch_distributed_training_amd.optim import FusedOptimizer

    torch.manual_seed(5)
    dev = _dev()
    shapes = [(64, 3, 7, 7), (1000, 2048), (2048,), (33,), (5, 7)]
    cpu = [nn.Parameter(torch.randn(s)) for s in shapes]
    gpu = [nn.Parameter(p.detach().to(dev)) for p in cpu]
    kw = dict(mode=mode, lr=0.01, weight_decay=0.05, momentum=0.9)
    oc, og = FusedOptimizer(cpu, **kw), FusedOptimizer(gpu, **kw)
    for _ in range(3):
        for pc, pg in zip(cpu, gpu):
            g = torch.randn_like(pc)
            pc.grad = g.clone()
            pg.grad = g.to(dev)
        oc.step()
        og.step()
    for pc, pg in zip(cpu, gpu):
        assert torch.allclose(pg.detach().cpu(), pc.detach(), atol=1e-6, rtol=1e-5)


def test_optimizer_refreshes_bf16_weight_copies(hip):
    from pytorch_distributed_training_amd.optim import FusedOptimizer

    dev = _dev()
    lin = nn.Linear(64, 64, bias=False).to(dev)
    x = torch.randn(8, 64, device=dev).to(torch.bfloat16)
    opt = FusedOptimizer(lin.parameters(), mode="sgd", lr=0.5, momentum=0.0, weight_decay=0.0)
    hip.linear(x, lin.weight).float().pow(2).sum().backward()
    opt.step()
    shadow = hip.SHADOWS.fwd(lin.weight)
    assert torch.equal(shadow, lin.weight.detach().to(torch.bfloat16))
    out = hip.linear(x, lin.weight)
    assert _rel_err(out, F.linear(x.float(), _bf16_round(lin.weight.detach()))) < 6e-3


def test_resnet18_matches_reference_model_and_learns():
    from pytorch_distributed_training_amd import ops
    from pytorch_distributed_training_amd.models import build_model
    from pytorch_distributed_training_amd.optim import FusedOptimizer

    torch.manual_seed(6)
    dev = _dev()
    ref = build_model("resnet18", num_classes=10)
    gpu = build_model("resnet18", num_classes=10)
    gpu.load_state_dict(ref.state_dict())
    gpu = gpu.to(dev).to(memory_format=torch.channels_last)
    x = torch.randn(16, 3, 64, 64)
    y = torch.randint(0, 10, (16,))
    ref_logits = ref(x)
    ref_loss = F.cross_entropy(ref_logits, y)
    ref_loss.backward()
    xb = ops.to_device_layout(x.to(dev))
    logits = gpu(xb)
    loss = ops.softmax_cross_entropy(logits, y.to(dev))
    loss.backward()
    assert _rel_err(logits.detach().cpu(), ref_logits.detach()) < 0.08
    assert abs(float(loss) - float(ref_loss)) < 0.08
    gr = dict(ref.named_parameters())
    for name, p in gpu.named_parameters():
        assert p.grad is not None, name
        assert p.grad.shape == p.shape
    assert _rel_err(gpu.fc.weight.grad.cpu(), gr["fc.weight"].grad) < 0.1
    # the stem gradient has crossed every bf16 layer of a tiny-batch BatchNorm network: direction, not digits
    g_stem, r_stem = gpu.conv1.weight.grad.cpu().flatten(), gr["conv1.weight"].grad.flatten()
    assert float(F.cosine_similarity(g_stem, r_stem, dim=0)) > 0.85
    opt = FusedOptimizer(gpu.parameters(), mode="adam", lr=1e-3)
    losses = []
    for _ in range(12):
        opt.zero_grad()
        l = ops.softmax_cross_entropy(gpu(xb), y.to(dev))
        l.backward()
        opt.step()
        losses.append(float(l))
    assert losses[-1] < 0.6 * losses[0]


@pytest.mark.parametrize("projected", [False, True])
def test_residual_link_fuses_shortcut_gradient_without_changing_it(hip, monkeypatch, projected):
    """The shortcut gradient added in the dgrad epilogue (from the block tail of an identity block, from the
    projection convolution of a down-sampling block) equals the one autograd would add in a separate pass."""
    from pytorch_distributed_training_amd import ops
    from pytorch_distributed_training_amd.models.resnet import Bottleneck, Conv2d

    torch.manual_seed(7)
    dev = _dev()
    if projected:
        ds = nn.Sequential(Conv2d(256, 512, 1, 2, 0), nn.BatchNorm2d(512))
        blk = Bottleneck(256, 128, 2, ds).to(dev).to(memory_format=torch.channels_last)
        g = hip.to_device_layout(torch.randn(4, 512, 7, 7, device=dev))
    else:
        blk = Bottleneck(256, 64).to(dev).to(memory_format=torch.channels_last)
        g = hip.to_device_layout(torch.randn(4, 256, 14, 14, device=dev))
    x0 = hip.to_device_layout(torch.randn(4, 256, 14, 14, device=dev))
    links = []
    real_link = ops.residual_link

    def spy(x):
        l = real_link(x)
        links.append(l)
        return l

    monkeypatch.setattr(ops, "residual_link", spy)
    # the statistics pivots settle on the batch means in the first forward pass: after it, two passes over the same
    # batch are bit-identical in forward, and the comparison below sees the fused shortcut gradient alone
    with torch.no_grad():
        blk(x0)
    links.clear()
    for m in blk.modules():
        if isinstance(m, nn.BatchNorm2d):
            m.reset_running_stats()
    xa = x0.clone().requires_grad_(True)
    blk(xa).backward(g)
    assert len(links) == 1 and links[0].armed and links[0].consumed and links[0].grad is None  # produced and consumed
    wa = blk.conv1.weight.grad.clone()
    blk.zero_grad(set_to_none=True)
    monkeypatch.setattr(ops, "residual_link", lambda x: None)
    for m in blk.modules():
        if isinstance(m, nn.BatchNorm2d):
            m.reset_running_stats()
    xb = x0.clone().requires_grad_(True)
    blk(xb).backward(g)
    assert _rel_err(xa.grad, xb.grad) < 4e-3  # one bf16 rounding apart (sum rounded once vs twice)
    assert _rel_err(wa, blk.conv1.weight.grad) < 1e-6


def test_capturable_optimizer_matches_plain_steps():
    """Adam with device-resident hyper-parameters - eagerly and replayed from a hipGraph - takes the same steps
    as the plain launch: the step count, bias corrections and a changed learning rate all reach the kernel."""
    from pytorch_distributed_training_amd.optim import FusedOptimizer

    dev = _dev()
    torch.manual_seed(9)
    shapes = [(64, 3, 7, 7), (256,), (1000, 512), (37,)]
    grads = [[torch.randn(*sh, device=dev) for sh in shapes] for _ in range(6)]

    def run(mode: str):
        torch.manual_seed(10)
        params = [nn.Parameter(torch.randn(*sh, device=dev)) for sh in shapes]
        opt = FusedOptimizer(params, mode="adam", lr=1e-2, weight_decay=1e-2)
        for p in params:
            p.grad = torch.zeros_like(p)
        if mode != "plain":
            opt.set_capturable(True)
        graph = None
        for i, gs in enumerate(grads):
            if i == 3:
                opt.param_groups[0]["lr"] = 3e-3  # schedules change hyper-parameters between steps
            for p, g in zip(params, gs):
                p.grad.copy_(g)
            if mode == "graph" and i >= 2:
                if graph is None:
                    torch.cuda.synchronize()
                    graph = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(graph):
                        opt.step()
                opt.advance()
                graph.replay()
                opt.notify_replayed()
            else:
                opt.step()
        assert opt.param_groups[0]["step"] == len(grads)
        return [p.detach().clone() for p in params]

    ref = run("plain")
    for mode in ("capturable", "graph"):
        got = run(mode)
        for a, b in zip(got, ref):
            assert torch.allclose(a, b, rtol=1e-6, atol=1e-7), mode


def test_recorded_step_matches_eager_training():
    """A hipGraph replay of the whole step (zero-grad, forward, backward, optimizer update) does what the eager
    step does.  fp32 atomics make the summation order, hence the last bits of the gradients, vary between runs
    (eager runs differ from each other the same way).  The comparison is kept meaningful by SGD (Adam would turn
    last-bit noise on near-zero gradient elements into full-size steps of either sign) and by a batch and image
    size at which no BatchNorm normalises over a handful of values (a 16-sample, 1x1-pixel layer amplifies
    1e-8 weight differences to 1e-3 loss differences within three steps)."""
    from pytorch_distributed_training_amd import ops
    from pytorch_distributed_training_amd.models import build_model
    from pytorch_distributed_training_amd.optim import FusedOptimizer
    from pytorch_distributed_training_amd.train.graph import GraphedStep

    dev = _dev()
    torch.manual_seed(5)
    xs = [ops.to_device_layout(torch.randn(48, 3, 64, 64, device=dev)) for _ in range(6)]
    ys = [torch.randint(0, 10, (48,), device=dev) for _ in range(6)]

    def run(graphed: bool):
        torch.manual_seed(6)
        model = ops.prepare_module(build_model("resnet18", 10, 64).to(dev).to(memory_format=torch.channels_last))
        opt = FusedOptimizer(model.parameters(), mode="sgd", lr=0.002, momentum=0.9, weight_decay=1e-4)  # a stable trajectory
        opt.zero_grad()

        def body(x, y):
            loss = ops.softmax_cross_entropy(model(x), y)
            loss.backward()
            opt.step()
            opt.zero_grad()
            return loss.detach()

        losses = []
        step = None
        for i, (x, y) in enumerate(zip(xs, ys)):
            if graphed and i == 2:
                step = GraphedStep(body, opt, (x, y))
            losses.append(float(step(x, y) if step is not None else body(x, y)))
        assert opt.param_groups[0]["step"] == len(xs)
        if graphed:
            assert step.replays == len(xs) - 2
            # code that runs eagerly after replays (evaluation, a last smaller batch) must see current weights in
            # every derived copy: the bf16 copies the recorded optimizer kernel rewrites, and the lazily rebuilt
            # ones (packed stem rows, transposed data-gradient weights) whose host-side freshness marks a replay
            # does not touch by itself
            from pytorch_distributed_training_amd.ops import hip as h

            w = model.conv1.weight
            n, c, r, s_ = w.shape
            packed = h._packed_rows_weight(w, 4, 32, 256)
            assert torch.equal(packed[:, : r * 32].view(n, r, 8, 4)[:, :, :s_, :c], w.detach().permute(0, 2, 3, 1).to(torch.bfloat16))
            w2 = model.layer1[0].conv1.weight
            phys = w2.detach().permute(0, 2, 3, 1).reshape(w2.shape[0], -1).to(torch.bfloat16)
            assert torch.equal(h.SHADOWS.fwd(w2), phys)
            wt, npad = h.SHADOWS.dgrad(w2, 9)
            assert torch.equal(wt.view(w2.shape[1], 9, npad), phys.view(w2.shape[0], 9, w2.shape[1]).permute(2, 1, 0))
        return losses, [p.detach().clone() for p in model.parameters()]

    def rel(pa, pb):
        num = sum(float((a - b).norm() ** 2) for a, b in zip(pa, pb)) ** 0.5
        return num / sum(float(a.norm() ** 2) for a in pa) ** 0.5

    le, pe = run(False)
    lg, pg = run(True)
    assert all(math.isfinite(v) for v in lg)
    assert le == pytest.approx(lg, abs=2e-2), (le, lg)
    assert rel(pe, pg) < 2e-3, rel(pe, pg)


def test_no_kernel_reads_memory_it_did_not_write(monkeypatch):
    """Every ``torch.empty`` buffer handed to a kernel is poisoned with NaN: a read of memory that the kernels
    did not write themselves (a padding column, a pixel no launch covers) surfaces as a non-finite loss or
    gradient instead of depending on what the allocator recycled."""
    from pytorch_distributed_training_amd import ops
    from pytorch_distributed_training_amd.models import build_model

    real_empty, real_empty_like = torch.empty, torch.empty_like

    def poison(t):
        if t.is_cuda and t.dtype in (torch.float32, torch.bfloat16):
            t.fill_(float("nan"))
        elif t.is_cuda and t.dtype == torch.uint8:
            t.fill_(255)
        return t

    monkeypatch.setattr(torch, "empty", lambda *a, **k: poison(real_empty(*a, **k)))
    monkeypatch.setattr(torch, "empty_like", lambda *a, **k: poison(real_empty_like(*a, **k)))
    dev = _dev()
    torch.manual_seed(8)
    for name, shape, ncls in (("resnet18", (16, 3, 32, 32), 10), ("resnet50", (5, 3, 75, 75), 37)):
        model = ops.prepare_module(build_model(name, ncls, shape[-1]).to(dev).to(memory_format=torch.channels_last))
        x = ops.to_device_layout(torch.randn(*shape, device=dev))
        loss = ops.softmax_cross_entropy(model(x), torch.randint(0, ncls, (shape[0],), device=dev))
        loss.backward()
        assert math.isfinite(float(loss.detach()))
        assert all(torch.isfinite(p.grad).all() for p in model.parameters()), name
    for name in ("vit_tiny", "gpt2_tiny"):
        model = ops.prepare_module(build_model(name).to(dev))
        if name == "vit_tiny":
            x = ops.to_device_layout(torch.randn(6, 3, 32, 32, device=dev))
            loss = ops.softmax_cross_entropy(model(x), torch.randint(0, 10, (6,), device=dev))
        else:
            ids = torch.randint(0, 512, (3, 50), device=dev)
            loss = model(ids, ids)
        loss.backward()
        assert math.isfinite(float(loss.detach()))
        assert all(torch.isfinite(p.grad).all() for p in model.parameters()), name


def test_bn_backward_sums_from_the_dgrad_epilogue_match_the_separate_reduction(hip, monkeypatch):
    """conv(relu(bn(y))): the data-gradient kernels of the convolution accumulate BatchNorm's backward sums
    (ring kernels only).  Gradients must equal those of the path where BatchNorm reduces by itself."""
    torch.manual_seed(12)
    dev = _dev()
    cases = [(80, 128, 256, 1, 1, 0, 56), (300, 128, 128, 3, 2, 1, 56), (70, 64, 64, 3, 1, 1, 56),
             (140, 64, 64, 3, 1, 1, 56)]  # rows enough for ring kernels; the last one for the shared-rows 3x3 kernel
    real_reduce, real_can = hip._bn_bwd_reduce, hip._dgrad_can_fuse_stats
    for nb, cin, cout, k, stride, pad, hw in cases:
        y = hip.to_device_layout(torch.randn(nb, cin, hw, hw, device=dev))
        w = nn.Parameter((torch.randn(cout, cin, k, k, device=dev) / math.sqrt(cin * k * k)).contiguous(memory_format=torch.channels_last))
        gamma, beta = torch.empty(cin, device=dev).uniform_(0.5, 1.5), torch.empty(cin, device=dev).uniform_(-0.5, 0.5)
        yf = y.float()
        st = torch.zeros(hip.STAT_SLOTS + 1, 2, cin, device=dev)  # (+ the slot that carries the pivot: zero)
        st[0, 0] = yf.sum(dim=(0, 2, 3))
        st[0, 1] = (yf * yf).sum(dim=(0, 2, 3))
        g = None
        res = {}
        for fused in (True, False):
            bn = nn.BatchNorm2d(cin).to(dev)
            with torch.no_grad():
                bn.weight.copy_(gamma)
                bn.bias.copy_(beta)
            used = []
            monkeypatch.setattr(hip, "_dgrad_can_fuse_stats", real_can if fused else (lambda launches: False))
            monkeypatch.setattr(hip, "_bn_bwd_reduce", lambda *a, **kw: (used.append(1), real_reduce(*a, **kw))[1])
            yy = y.clone().requires_grad_(True)
            w.grad = None
            out, _ = hip.bn_relu_conv2d(yy, st.clone(), bn, w, stride, pad)
            if g is None:
                g = torch.randn(out.shape, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
            out.backward(g)
            assert (len(used) == 0) == fused, (fused, used)   # the separate reduction ran exactly when not fused
            res[fused] = (yy.grad.float().clone(), bn.weight.grad.clone(), bn.bias.grad.clone(), w.grad.clone())
        for a, b in zip(res[True], res[False]):
            assert _rel_err(a, b) < 2e-3


@pytest.mark.parametrize("tile", [13, 16])
@pytest.mark.parametrize("blocks", ["identity+identity", "identity+projected", "projected+identity"])
def test_block_tail_bn_sums_come_from_the_next_blocks_first_dgrad(hip, monkeypatch, tile, blocks):
    """Two residual blocks in a row: the data-gradient kernel of the second block's first convolution (which also
    adds the shortcut gradient: the tail's of an identity block, the projection's of a down-sampling block)
    accumulates the backward sums of the first block's tail BatchNorm, so that tail skips its own reduction.
    Every set of sums a BatchNorm backward receives is compared with a separate reduction over the same gradient
    (end-to-end gradients of two runs differ by more than that: statistics accumulated with atomics change the
    last bits of the normalisation constants from run to run)."""
    from pytorch_distributed_training_amd.models.resnet import Bottleneck, Conv2d

    torch.manual_seed(21)
    dev = _dev()
    monkeypatch.setattr(hip, "_TILE_WIDE", tile)
    monkeypatch.setattr(hip, "_TILE_NARROW", 14)  # ring kernels for the 64-channel layers too, whatever the row count
    if blocks == "projected+identity":  # the first tail has two BatchNorms (main branch and projected shortcut)
        b1 = Bottleneck(256, 128, 2, nn.Sequential(Conv2d(256, 512, 1, 2, 0), nn.BatchNorm2d(512)))
        b2 = Bottleneck(512, 128)
    else:
        b1 = Bottleneck(256, 64)
        b2 = (Bottleneck(256, 64) if blocks == "identity+identity"
              else Bottleneck(256, 128, 2, nn.Sequential(Conv2d(256, 512, 1, 2, 0), nn.BatchNorm2d(512))))
    net = nn.Sequential(b1, b2).to(dev).to(memory_format=torch.channels_last)
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.weight.uniform_(0.5, 1.5)
                m.bias.uniform_(-0.3, 0.3)
    real = hip._bn_backward
    seen = []

    def spy(dout, out, y, consts, gamma, beta, count, c, *args, **kw):
        slots, mask_mode = kw.get("slots"), kw.get("mask_mode")
        y2, consts2 = (args[0], args[1]) if len(args) >= 2 else (None, None)
        if slots is not None:
            ref = torch.zeros_like(slots)
            hip._check(hip._bn_bwd_reduce(dout.data_ptr(), hip._ptr(out), y.data_ptr(), consts.data_ptr(), hip._ptr(y2),
                                          hip._ptr(consts2), ref.data_ptr(), y.numel() // c, c, mask_mode, hip._stream()),
                       "bn_bwd_reduce")
            a, b = slots.sum(0), ref.sum(0)
            assert _rel_err(a[0], b[0]) < 1e-5 and _rel_err(a[1], b[1]) < 1e-5, (mask_mode, c)
            if y2 is not None:
                assert _rel_err(a[2], b[2]) < 1e-5, "sums of the shortcut BatchNorm"
        seen.append((mask_mode, c, slots is not None, y2 is not None))
        return real(dout, out, y, consts, gamma, beta, count, c, *args, **kw)

    monkeypatch.setattr(hip, "_bn_backward", spy)
    x = hip.to_device_layout(torch.randn(8 if blocks != "projected+identity" else 24, 256, 28, 28, device=dev)).requires_grad_(True)
    out = net(x)
    out.backward(torch.randn(out.shape, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last))
    assert torch.isfinite(x.grad.float()).all()
    # backward order: tail of block 2 (nothing behind it: reduces by itself), its two inner BatchNorms, the tail of
    # block 1 (mask bits, 256 channels, sums from block 2's first convolution), block 1's inner BatchNorms
    assert [s_[2] for s_ in seen] == [False, True, True, True, True, True], seen
    assert seen[3][:2] == (3, 512 if blocks == "projected+identity" else 256)
    assert seen[3][3] == (blocks == "projected+identity")


@pytest.mark.parametrize("nb,hw,fusable", [(70, 56, True), (240, 29, True), (56, 61, True), (400, 23, False), (4, 56, False)])
def test_bn_relu_fused_into_the_shared_rows_3x3_kernels(hip, monkeypatch, nb, hw, fusable):
    """conv3x3(relu(bn(y))), 64 -> 64 channels: the kernels that keep one copy of the input rows in LDS apply the
    normalisation to that copy (forward and weight gradient), so the normalised activation is never written.  Output,
    statistics and all gradients against the path that materialises relu(bn(y)), and against fp32 PyTorch.  Images too
    narrow for the kernel, or too few rows, take the materialising path."""
    torch.manual_seed(6)
    dev = _dev()
    cin = cout = 64
    y = hip.to_device_layout(torch.randn(nb, cin, hw, hw, device=dev) * 1.5 + 0.2)
    w = nn.Parameter((torch.randn(cout, cin, 3, 3, device=dev) / math.sqrt(cin * 9)).contiguous(memory_format=torch.channels_last))
    gamma, beta = torch.empty(cin, device=dev).uniform_(0.5, 1.5), torch.empty(cin, device=dev).uniform_(-0.5, 0.5)
    yf = y.float()
    st = torch.zeros(hip.STAT_SLOTS + 1, 2, cin, device=dev)
    st[0, 0] = yf.sum(dim=(0, 2, 3))
    st[0, 1] = (yf * yf).sum(dim=(0, 2, 3))
    g = torch.randn(nb, cout, hw, hw, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    res = {}
    launched = []
    real_launch = hip._launch_gather_gemm
    monkeypatch.setattr(hip, "_launch_gather_gemm", lambda *a, **kw: (launched.append(kw.get("xf") is not None), real_launch(*a, **kw))[1])
    for fused in (True, False):
        monkeypatch.setattr(hip, "FUSE_BN_INTO_3X3", 1 if fused else 0)
        bn = nn.BatchNorm2d(cin).to(dev)
        with torch.no_grad():
            bn.weight.copy_(gamma)
            bn.bias.copy_(beta)
        for p in list(bn.parameters()) + [w]:
            p.grad = None
        yy = y.clone().requires_grad_(True)
        launched.clear()
        out, stats = hip.bn_relu_conv2d(yy, st.clone(), bn, w, 1, 1)
        assert any(launched) == (fused and fusable), "the operand transform was (not) used"
        out.backward(g)
        torch.cuda.synchronize()
        res[fused] = (out.float().clone(), stats[:-1].sum(0).clone(), yy.grad.float().clone(), bn.weight.grad.clone(),
                      bn.bias.grad.clone(), w.grad.clone())
    for name, a, b in zip(("out", "stats", "dy", "dgamma", "dbeta", "dw"), res[True], res[False]):
        assert _rel_err(a, b) < 4e-3, name
    yr = yf.clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    wr = _bf16_round(w.detach()).requires_grad_(True)
    a = F.relu(F.batch_norm(yr, None, None, gr, br, training=True, eps=1e-5))
    o = F.conv2d(a, wr, padding=1)
    o.backward(g.float())
    out, _, dy, dgamma, dbeta, dw = res[True]
    assert _rel_err(out, o) < 1e-2
    assert _rel_err(dy, yr.grad) < 2e-2
    assert _rel_err(dgamma, gr.grad) < 2e-2 and _rel_err(dbeta, br.grad) < 2e-2
    assert _rel_err(dw, wr.grad) < 1e-2


def test_transposed_weight_copies_are_rebuilt_in_one_launch_after_the_optimizer_step(hip):
    """The data-gradient kernels read [Cin][taps][Cout] bf16 copies of the weights; once they exist, the optimizer step
    refreshes all of them with one batched launch, so the next backward pass finds them current."""
    from pytorch_distributed_training_amd import ops
    from pytorch_distributed_training_amd.optim import FusedOptimizer

    torch.manual_seed(9)
    dev = _dev()
    conv = nn.Conv2d(64, 128, 3, padding=1, bias=False).to(dev).to(memory_format=torch.channels_last)
    lin = nn.Linear(192, 320).to(dev)
    params = [conv.weight, lin.weight, lin.bias]
    opt = FusedOptimizer(params, mode="sgd", lr=0.1, momentum=0.0, weight_decay=0.0)
    x = ops.to_device_layout(torch.randn(4, 64, 16, 16, device=dev))
    z = torch.randn(50, 192, device=dev).to(torch.bfloat16)
    for _ in range(2):
        opt.zero_grad()
        xi = x.clone().requires_grad_(True)
        zi = z.clone().requires_grad_(True)
        y, _ = hip.conv2d_stats(xi, conv.weight, 1, 1, False, None, None)
        (y.float().mean() + hip.linear(zi, lin.weight, lin.bias).float().mean()).backward()
        opt.step()
        for w, taps in ((conv.weight, 9), (lin.weight, 1)):
            e = hip.SHADOWS._entries[id(w)]
            assert e.dgrad_epoch == hip.SHADOWS.epoch  # refreshed by the batched launch, not lazily
            wt, npad = hip.SHADOWS.dgrad(w, taps)
            mat = hip.SHADOWS.physical_matrix(w)  # [N][taps*C]
            n, c = mat.shape[0], mat.shape[1] // taps
            ref = mat.view(n, taps, c).permute(2, 1, 0).reshape(c, taps * n).to(torch.bfloat16)
            assert npad == n and torch.equal(wt, ref)


@pytest.mark.parametrize("shape", [(3, 5, 7), (16, 224, 224), (2, 32, 32)])
def test_u8_image_batch_to_bf16_nhwc(shape):
    """uint8 NHWC -> normalised bf16 NHWC in one kernel, against (x / 255 - mean) / std in fp32."""
    from pytorch_distributed_training_amd import ops

    n, h, w = shape
    g = torch.Generator().manual_seed(0)
    x = torch.randint(0, 256, (n, h, w, 3), dtype=torch.uint8, generator=g)
    mean, std = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
    for m, s in ((None, None), (mean, std)):
        out = ops.image_batch_to_device_layout(x.cuda(), torch.bfloat16, m, s)
        assert out.shape == (n, 3, h, w) and out.is_contiguous(memory_format=torch.channels_last) or n * h * w < 8
        ref = ops.image_batch_to_device_layout(x, torch.float32, m, s)
        assert torch.equal(out.float().cpu(), ref.to(torch.bfloat16).float())


@pytest.mark.parametrize("case", [(16, 64, 256, 56, True), (64, 128, 512, 28, True), (128, 512, 2048, 7, True),
                                  (53, 256, 1024, 14, True),  # 10 388 rows: the last 256-row tile is partial
                                  (8, 256, 1024, 14, False)])  # too few tiles for the kernel that carries the transform
def test_bn_relu_fused_into_the_1x1_convolution_kernels(hip, monkeypatch, case):
    """conv1x1(relu(bn(y))) with the normalisation applied inside the forward GEMM and the weight-gradient kernel
    (the normalised activation is never written): output, statistics and all four gradients against the path that
    materialises relu(bn(y)), and against fp32 PyTorch on the same operands."""
    torch.manual_seed(5)
    dev = _dev()
    nb, cin, cout, hw, fusable = case
    y = hip.to_device_layout(torch.randn(nb, cin, hw, hw, device=dev) * 1.5 + 0.2)
    w = nn.Parameter((torch.randn(cout, cin, 1, 1, device=dev) / math.sqrt(cin)).contiguous(memory_format=torch.channels_last))
    gamma, beta = torch.empty(cin, device=dev).uniform_(0.5, 1.5), torch.empty(cin, device=dev).uniform_(-0.5, 0.5)
    yf = y.float()
    st = torch.zeros(hip.STAT_SLOTS + 1, 2, cin, device=dev)  # (+ the slot that carries the pivot: zero)
    st[0, 0] = yf.sum(dim=(0, 2, 3))
    st[0, 1] = (yf * yf).sum(dim=(0, 2, 3))
    g = torch.randn(nb, cout, hw, hw, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    res = {}
    launched = []
    real_launch = hip._launch_gather_gemm
    monkeypatch.setattr(hip, "_launch_gather_gemm", lambda *a, **kw: (launched.append(kw.get("xf") is not None), real_launch(*a, **kw))[1])
    for fused in (True, False):
        monkeypatch.setattr(hip, "FUSE_BN_INTO_1X1", 2 if fused else 0)  # 2: wherever the kernels allow
        bn = nn.BatchNorm2d(cin).to(dev)
        with torch.no_grad():
            bn.weight.copy_(gamma)
            bn.bias.copy_(beta)
        for p in list(bn.parameters()) + [w]:
            p.grad = None
        yy = y.clone().requires_grad_(True)
        launched.clear()
        out, stats = hip.bn_relu_conv2d(yy, st.clone(), bn, w, 1, 0)
        assert any(launched) == (fused and fusable), "the operand transform was (not) used"
        out.backward(g)
        torch.cuda.synchronize()
        res[fused] = (out.float().clone(), stats[:-1].sum(0).clone(), yy.grad.float().clone(), bn.weight.grad.clone(),
                      bn.bias.grad.clone(), w.grad.clone(), bn.running_mean.clone())
    for name, a, b in zip(("out", "stats", "dy", "dgamma", "dbeta", "dw", "running_mean"), res[True], res[False]):
        assert _rel_err(a, b) < 4e-3, name
    # fp32 reference: batch statistics, affine, relu, 1x1 convolution on the bf16-rounded operands
    yr = yf.clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    wr = _bf16_round(w.detach()).requires_grad_(True)
    a = F.relu(F.batch_norm(yr, None, None, gr, br, training=True, eps=1e-5))
    o = F.conv2d(a, wr)
    o.backward(g.float())
    out, _, dy, dgamma, dbeta, dw, _ = res[True]
    assert _rel_err(out, o) < 1e-2
    assert _rel_err(dy, yr.grad) < 2e-2
    assert _rel_err(dgamma, gr.grad) < 2e-2 and _rel_err(dbeta, br.grad) < 2e-2
    assert _rel_err(dw, wr.grad) < 1e-2
