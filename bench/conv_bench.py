"""Per-layer micro-benchmark of the convolution kernels over the distinct ResNet-50 layer shapes.

For every shape it times forward, data gradient and weight gradient separately (HIP events, median of
``--iters`` runs) and prints achieved TFLOP/s and effective GB/s (minimum traffic: each operand read once,
result written once), so memory-bound and compute-bound layers can be told apart.

    python bench/conv_bench.py --batch 256 [--only 3x3] [--json out.json]
"""

from __future__ import annotations

import argparse
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402
import torch.nn as nn  # noqa: E402

# (name, cin, cout, k, stride, pad, input hw, count in ResNet-50)
RESNET50_LAYERS = [
    ("l1.c1_first", 64, 64, 1, 1, 0, 56, 1),
    ("l1.c1", 256, 64, 1, 1, 0, 56, 2),
    ("l1.c2", 64, 64, 3, 1, 1, 56, 3),
    ("l1.c3", 64, 256, 1, 1, 0, 56, 4),
    ("l2.c1_first", 256, 128, 1, 1, 0, 56, 1),
    ("l2.c2_s2", 128, 128, 3, 2, 1, 56, 1),
    ("l2.ds", 256, 512, 1, 2, 0, 56, 1),
    ("l2.c1", 512, 128, 1, 1, 0, 28, 3),
    ("l2.c2", 128, 128, 3, 1, 1, 28, 3),
    ("l2.c3", 128, 512, 1, 1, 0, 28, 4),
    ("l3.c1_first", 512, 256, 1, 1, 0, 28, 1),
    ("l3.c2_s2", 256, 256, 3, 2, 1, 28, 1),
    ("l3.ds", 512, 1024, 1, 2, 0, 28, 1),
    ("l3.c1", 1024, 256, 1, 1, 0, 14, 5),
    ("l3.c2", 256, 256, 3, 1, 1, 14, 5),
    ("l3.c3", 256, 1024, 1, 1, 0, 14, 6),
    ("l4.c1_first", 1024, 512, 1, 1, 0, 14, 1),
    ("l4.c2_s2", 512, 512, 3, 2, 1, 14, 1),
    ("l4.ds", 1024, 2048, 1, 2, 0, 14, 1),
    ("l4.c1", 2048, 512, 1, 1, 0, 7, 2),
    ("l4.c2", 512, 512, 3, 1, 1, 7, 2),
    ("l4.c3", 512, 2048, 1, 1, 0, 7, 3),
]


def time_fn(fn, iters: int, warmup: int = 3) -> float:
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    times = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        b.synchronize()
        times.append(a.elapsed_time(b) * 1e-3)
    return statistics.median(times)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--only", type=str, default="")
    ap.add_argument("--json", type=str, default="")
    args = ap.parse_args()
    from pytorch_distributed_training_amd.ops import hip

    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    rows = []
    tot = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
    print(f"{'layer':<14}{'shape':<28}{'pass':<7}{'us':>9}{'TF/s':>8}{'GB/s':>8}  x count")
    for name, cin, cout, k, stride, pad, hw, count in RESNET50_LAYERS:
        if args.only and args.only not in name:
            continue
        ho = (hw + 2 * pad - k) // stride + 1
        x = hip.to_device_layout(torch.randn(args.batch, cin, hw, hw, device=dev)).requires_grad_(True)
        w = nn.Parameter((torch.randn(cout, cin, k, k, device=dev) * 0.05).contiguous(memory_format=torch.channels_last))
        y, _ = hip.conv2d_stats(x, w, stride, pad)
        g = torch.randn_like(y)
        flops = 2.0 * args.batch * ho * ho * cout * cin * k * k
        bx, by, bw = x.numel() * 2, y.numel() * 2, w.numel() * 2

        def fwd():
            hip.conv2d_stats(x, w, stride, pad)

        class Ctx:  # drives the Function's backward directly so each gradient is timed alone
            saved_tensors = (x.detach(), w)
            geom = (args.batch, cin, hw, hw, cout, k, k, ho, ho, stride, pad)
            col = None
            link = None
            link_out = None
            stat_link = None
            needs_input_grad = (False, False, False, False, False)

        def dgrad():
            Ctx.needs_input_grad = (True, False, False, False, False)
            hip._ConvStats.backward(Ctx, g, None)

        def wgrad():
            Ctx.needs_input_grad = (False, True, False, False, False)
            hip._ConvStats.backward(Ctx, g, None)

        for pname, fn, traffic in (("fwd", fwd, bx + bw + by), ("dgrad", dgrad, by + bw + bx), ("wgrad", wgrad, by + bx + 2 * bw)):
            t = time_fn(fn, args.iters)
            tot[pname] += t * count
            shape = f"{cin}->{cout} k{k} s{stride} @{hw}"
            print(f"{name:<14}{shape:<28}{pname:<7}{t * 1e6:9.1f}{flops / t / 1e12:8.1f}{traffic / t / 1e9:8.0f}  x{count}")
            rows.append(dict(layer=name, cin=cin, cout=cout, k=k, stride=stride, hw=hw, count=count, op=pname,
                             us=t * 1e6, tflops=flops / t / 1e12, gbps=traffic / t / 1e9))
        del x, w, y, g
    print("weighted totals (ms/step):", {k: round(v * 1e3, 3) for k, v in tot.items()}, "sum", round(sum(tot.values()) * 1e3, 3))
    if args.json:
        os.makedirs(os.path.dirname(os.path.abspath(args.json)), exist_ok=True)
        with open(args.json, "w") as f:
            json.dump({"batch": args.batch, "rows": rows, "totals_ms": {k: v * 1e3 for k, v in tot.items()}}, f, indent=1)


if __name__ == "__main__":
    main()
