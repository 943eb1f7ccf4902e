"""Self-measured baseline: the same training step on stock PyTorch-ROCm.

The model definition is this repository's with every fused operator replaced by its plain PyTorch
composition (``ops.reference``), so convolutions go to MIOpen, GEMMs to hipBLASLt, BatchNorm/ReLU/
LayerNorm/GELU to ATen, attention to ``F.scaled_dot_product_attention``, the optimizer to
``torch.optim`` (SGD foreach for the CNNs, fused AdamW for the transformers) - channels_last, bf16
autocast, synthetic device-resident data.  This is what a user of the reference script would get on an
MI355X after fixing its device placement; it is the number the hand-written path has to beat.

    python bench/stock_baseline.py --steps 20 --warmup 5 --batch-size 256
    python bench/stock_baseline.py --model vit_b_16 --batch-size 256
    python bench/stock_baseline.py --model gpt2_medium --batch-size 16 --seq-len 1024
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
    ap.add_argument("--seq-len", type=int, default=1024)
    ap.add_argument("--miopen-find", action="store_true", help="cudnn.benchmark=True (exhaustive MIOpen search; slow start)")
    args = ap.parse_args()

    from pytorch_distributed_training_amd import ops
    from pytorch_distributed_training_amd.models import build_model
    from pytorch_distributed_training_amd.ops import reference as ref

    # route every fused op to its PyTorch composition, regardless of device
    for name in ("conv2d_stats", "bn_act", "bn_relu_conv2d", "bn_add_relu", "bn_relu_maxpool", "global_avgpool", "linear",
                 "softmax_cross_entropy", "layer_norm", "gelu", "embedding", "add_bias_residual", "attention"):
        setattr(ops, name, getattr(ref, name))

    def sdpa_qkv(qkv, heads, causal=False):  # the library's fused attention instead of the fp32 definition
        b, s_, three_d = qkv.shape
        d = three_d // 3
        x = qkv.view(b, s_, 3, heads, d // heads)
        q, k, v = (x[:, :, i].transpose(1, 2) for i in range(3))
        return F.scaled_dot_product_attention(q, k, v, is_causal=causal).transpose(1, 2).reshape(b, s_, d)

    ops.attention_qkv = sdpa_qkv
    ops.residual_link = lambda x: None
    ops.skip_link = lambda x: None
    ops.gelu_mlp = lambda x, w1, b1, w2, b2, residual=None, link=None: ref.linear(
        ref.linear(x, w1, b1, activation="gelu"), w2, b2, residual=residual)

    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    torch.backends.cudnn.benchmark = bool(args.miopen_find)
    lm = args.model.startswith("gpt")
    if lm:
        model = build_model(args.model, n_ctx=args.seq_len).to(dev)
        vocab = model.vocab_size
        opt = torch.optim.AdamW(model.parameters(), lr=3e-4, weight_decay=0.01, fused=True)
        xs = [torch.randint(0, vocab, (args.batch_size, args.seq_len), device=dev) for _ in range(2)]
        ys = [torch.randint(0, vocab, (args.batch_size, args.seq_len), device=dev) for _ in range(2)]
    else:
        model = build_model(args.model, 1000, args.image_size).to(dev).to(memory_format=torch.channels_last)
        if args.model.startswith("vit"):
            opt = torch.optim.AdamW(model.parameters(), lr=3e-4, weight_decay=0.01, fused=True)
        else:
            opt = torch.optim.SGD(model.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-4)
        xs = [torch.randn(args.batch_size, 3, args.image_size, args.image_size, device=dev).contiguous(
            memory_format=torch.channels_last) for _ in range(2)]
        ys = [torch.randint(0, 1000, (args.batch_size,), device=dev) for _ in range(2)]

    def step(i):
        with torch.autocast("cuda", dtype=torch.bfloat16):
            if lm:
                loss = model(xs[i % 2], ys[i % 2])
            else:
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
        "metric": f"stock PyTorch-ROCm {args.model} bf16 autocast training",
        "value": round(args.batch_size * (args.seq_len if lm else 1) * args.steps / dt, 2),
        "unit": "tokens/s" if lm else "images/s", "n_gpus": 1,
        "ms_per_step": round(1e3 * dt / args.steps, 3), "batch": args.batch_size, "final_loss": round(float(loss), 4),
        "torch": torch.__version__,
    }))


if __name__ == "__main__":
    main()
