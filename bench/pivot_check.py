"""BatchNorm statistics through the pivoted sums, over the convolution shapes of a ResNet: mean and variance from
``bn_finalize`` against fp64 statistics of the stored output, for three consecutive steps (pivot 0, then the previous
batch mean).  Prints the worst relative variance error per shape and step."""
import math
import sys

import torch
import torch.nn as nn

sys.path.insert(0, ".")
from pytorch_distributed_training_amd.ops import hip  # noqa: E402

CASES = [(4, 256, 128, 1, 1, 0, 14), (4, 128, 128, 3, 2, 1, 14), (4, 128, 512, 1, 1, 0, 7), (4, 256, 512, 1, 2, 0, 14),
         (16, 64, 64, 1, 1, 0, 28), (16, 64, 256, 1, 1, 0, 14), (64, 64, 64, 3, 1, 1, 56), (64, 256, 64, 1, 1, 0, 56),
         (32, 512, 512, 3, 1, 1, 7), (8, 3, 64, 7, 2, 3, 64), (256, 1024, 256, 1, 1, 0, 14), (128, 512, 2048, 1, 1, 0, 7)]
dev = torch.device("cuda")
for nb, cin, cout, k, stride, pad, hw in CASES:
    torch.manual_seed(0)
    x = torch.randn(nb, cin, hw, hw, device=dev)
    w = nn.Parameter((torch.randn(cout, cin, k, k, device=dev) / math.sqrt(cin * k * k)).contiguous(memory_format=torch.channels_last))
    xb = hip.to_device_layout(x)
    bn = nn.BatchNorm2d(cout).to(dev)
    out = []
    for step in range(3):
        y, st = hip.conv2d_stats(xb, w, stride, pad)
        count = y.numel() // cout
        consts = hip._bn_consts(st, bn, count, y.device)
        yd = y.detach().double()
        mean, var = yd.mean(dim=(0, 2, 3)), yd.var(dim=(0, 2, 3), unbiased=False)
        got_var = 1.0 / consts[1].double() ** 2 - bn.eps
        out.append((float(((consts[0].double() - mean).abs() / var.sqrt()).max()), float(((got_var - var).abs() / var).max())))
    M = y.numel() // cout
    print(f"{(nb, cin, cout, k, stride, hw)} M={M} tile={hip._pick_tile(M, cout, k * k * max(cin // 64, 1))}: "
          + "  ".join(f"dmean/std {a:.1e} dvar/var {b:.1e}" for a, b in out), flush=True)
