"""Fused attention kernels alone: forward and backward time on the GPT-2 medium and ViT-B/16 shapes.

    python bench/attn_bench.py [substring of a shape name]
"""
from __future__ import annotations

import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

SHAPES = [("gpt2-medium b32", 32, 1024, 16, True), ("vit-b/16 b256", 256, 197, 12, False), ("long 4k", 4, 4096, 16, True),
          ("long 4k full", 4, 4096, 16, False)]


def _time(run, iters=20, warm_s=0.25, repeats=3):
    # Warm up for a quarter of a second: the clocks of an idle GPU take ~0.2 s of load to settle (a backward pass
    # timed 30 ms after start-up reads 10-15 % slower than the same launches inside a running job); median of three.
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < warm_s:
        for _ in range(5):
            run()
        torch.cuda.synchronize()
    times = []
    for _ in range(repeats):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            run()
        e1.record()
        torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1) * 1e-3 / iters)
    return sorted(times)[len(times) // 2]


def main():
    from pytorch_distributed_training_amd.ops import hip_transformer as ht

    dev = torch.device("cuda", 0)
    print(f"{'shape':<18}{'fwd us':>9}{'TF/s':>7}{'bwd us':>9}{'TF/s':>7}   (FLOPs: 4 B H S^2 64 forward, 2.5x backward; halved when causal)")
    for name, b, s, h, causal in SHAPES:
        if len(sys.argv) > 1 and sys.argv[1] not in name:
            continue
        qkv = torch.randn(b, s, 3 * h * 64, device=dev).to(torch.bfloat16).requires_grad_(True)
        g = torch.randn(b, s, h * 64, device=dev).to(torch.bfloat16)
        flops = 4.0 * b * h * s * s * 64 * (0.5 if causal else 1.0)
        with torch.no_grad():
            tf = _time(lambda: ht.attention_qkv(qkv, h, causal))
        out = ht.attention_qkv(qkv, h, causal)

        def bwd():
            qkv.grad = None
            out.backward(g, retain_graph=True)

        tb = _time(bwd)
        print(f"{name:<18}{tf * 1e6:9.1f}{flops / tf / 1e12:7.0f}{tb * 1e6:9.1f}{2.5 * flops / tb / 1e12:7.0f}")


if __name__ == "__main__":
    main()
