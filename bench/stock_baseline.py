"""Self-measured baseline: the same ResNet-50 training step on stock PyTorch-ROCm.

The model definition is this repository's ResNet with every fused operator replaced by its
plain PyTorch composition (``ops.reference``), so convolutions go to MIOpen, BatchNorm/ReLU to
ATen, the optimizer to ``torch.optim.SGD`` (foreach) - channels_last, bf16 autocast, synthetic
device-resident data.  This is what a user of the reference script would get on an MI355X after
fixing its device placement; it is the number the hand-written path has to beat.

    python bench/stock_baseline.py --steps 20 --warmup 5 --batch-size 256
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch-size", type=int, default=256)
    ap.add_argument("--model", type=str, default="resnet50")
    ap.add_argument("--image-size", type=int, default=224)
    ap.add_argument("--miopen-find", action="store_true", help="cudnn.benchmark=True (exhaustive MIOpen search; slow start)")
    args = ap.parse_args()

    from pytorch_distributed_training_amd import ops
    from pytorch_distributed_training_amd.models import build_model
    from pytorch_distributed_training_amd.ops import reference as ref

    # route every fused op to its PyTorch composition, regardless of device
    for name in ("conv2d_stats", "bn_act", "bn_relu_conv2d", "bn_add_relu", "bn_relu_maxpool", "global_avgpool", "linear"):
        setattr(ops, name, getattr(ref, name))

    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    torch.backends.cudnn.benchmark = bool(args.miopen_find)
    model = build_model(args.model, 1000, args.image_size).to(dev).to(memory_format=torch.channels_last)
    opt = torch.optim.SGD(model.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-4)
    xs = [torch.randn(args.batch_size, 3, args.image_size, args.image_size, device=dev).contiguous(
        memory_format=torch.channels_last) for _ in range(2)]
    ys = [torch.randint(0, 1000, (args.batch_size,), device=dev) for _ in range(2)]

    def step(i):
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = F.cross_entropy(model(xs[i % 2]).float(), ys[i % 2])
        loss.backward()
        opt.step()
        opt.zero_grad(set_to_none=True)
        return loss

    for i in range(args.warmup):
        step(i)
        torch.cuda.synchronize()
        print(f"warmup step {i} done at {time.perf_counter():.1f}", file=sys.stderr, flush=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = step(i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({
        "metric": f"stock PyTorch-ROCm {args.model} {args.image_size}^2 bf16 autocast channels_last training",
        "value": round(args.batch_size * args.steps / dt, 2), "unit": "images/s", "n_gpus": 1,
        "ms_per_step": round(1e3 * dt / args.steps, 3), "batch": args.batch_size, "final_loss": round(float(loss), 4),
        "torch": torch.__version__,
    }))


if __name__ == "__main__":
    main()
