"""conv1x1(relu(bn(y))) forward + backward per ResNet-50 stage shape (batch 2048): BatchNorm applied as its own pass
versus inside the consumer's GEMM / weight-gradient kernels.

    python bench/bn_fuse_bench.py [batch]
"""
from __future__ import annotations

import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402
import torch.nn as nn  # noqa: E402


def main():
    from pytorch_distributed_training_amd.ops import hip

    nb = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    dev = torch.device("cuda", 0)
    print(f"{'cin':>5}{'cout':>6}{'hw':>4}   {'separate pass':>26}   {'fused':>26}   (forward ms, backward ms, sum)")
    for cin, cout, hw in ((64, 256, 56), (128, 512, 28), (256, 1024, 14), (512, 2048, 7)):
        y = hip.to_device_layout(torch.randn(nb, cin, hw, hw, device=dev))
        w = nn.Parameter((torch.randn(cout, cin, 1, 1, device=dev) / math.sqrt(cin)).contiguous(memory_format=torch.channels_last))
        bn = nn.BatchNorm2d(cin).to(dev)
        yf = y.float()
        st = torch.zeros(hip.STAT_SLOTS + 1, 2, cin, device=dev)  # (+ the slot that carries the pivot: zero)
        st[0, 0] = yf.sum(dim=(0, 2, 3))
        st[0, 1] = (yf * yf).sum(dim=(0, 2, 3))
        del yf
        g = torch.randn(nb, cout, hw, hw, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        cols = []
        for fused in (0, 2):
            hip.FUSE_BN_INTO_1X1 = fused
            tf = tb = 0.0
            iters = 8
            for it in range(iters + 2):
                yy = y.detach().requires_grad_(True)
                w.grad = None
                e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
                e[0].record()
                out, _ = hip.bn_relu_conv2d(yy, st.clone(), bn, w, 1, 0)
                e[1].record()
                out.backward(g)
                e[2].record()
                torch.cuda.synchronize()
                if it >= 2:
                    tf += e[0].elapsed_time(e[1]) / iters
                    tb += e[1].elapsed_time(e[2]) / iters
                del out, yy
            cols.append(f"{tf:8.3f} {tb:8.3f} {tf + tb:8.3f}")
        print(f"{cin:5d}{cout:6d}{hw:4d}   {cols[0]}   {cols[1]}", flush=True)


if __name__ == "__main__":
    main()
