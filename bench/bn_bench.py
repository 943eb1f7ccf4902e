"""Bandwidth of the BatchNorm element-wise / reduction kernels on ResNet-50 activation shapes.

    python bench/bn_bench.py [--batch 2048] [--iters 10]

For every shape it times the forward apply (plain, with identity, with mask bits), the backward reduction and
the backward apply, and prints microseconds and the achieved HBM traffic (bytes the kernel must move / time).
"""

from __future__ import annotations

import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402


def timed(fn, iters):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / iters  # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=2048)
    ap.add_argument("--iters", type=int, default=10)
    args = ap.parse_args()
    from pytorch_distributed_training_amd.ops import hip as H

    dev = torch.device("cuda", 0)
    shapes = [(56, 64), (56, 256), (28, 512), (14, 1024), (7, 2048)]
    print(f"{'shape':<16}{'kernel':<28}{'us':>10}{'GB/s':>9}")
    tot = 0.0
    for hw, C in shapes:
        M = args.batch * hw * hw
        n = M * C
        g = torch.Generator(device=dev).manual_seed(0)
        y = torch.randn(n, device=dev, generator=g).to(torch.bfloat16)
        ident = torch.randn(n, device=dev, generator=g).to(torch.bfloat16)
        dout = torch.randn(n, device=dev, generator=g).to(torch.bfloat16)
        out = torch.empty_like(y)
        dy = torch.empty_like(y)
        dz = torch.empty_like(y)
        bits = torch.empty(n // 8, dtype=torch.uint8, device=dev)
        consts = torch.randn(4 * C, device=dev)
        coef = torch.randn(3 * C, device=dev)
        slots = torch.zeros(H.STAT_SLOTS * 3 * C, device=dev)
        s = H._stream()
        P = H._ptr
        cases = [
            ("fwd relu", 2 * 2 * n, lambda: H._bn_apply_fwd(P(y), P(consts), None, None, P(out), None, n, C, 1, s)),
            ("fwd relu+ident+bits", 2 * 3 * n + n // 8,
             lambda: H._bn_apply_fwd(P(y), P(consts), P(ident), None, P(out), P(bits), n, C, 1, s)),
            ("bwd reduce mask=y", 2 * 2 * n,
             lambda: H._bn_bwd_reduce(P(dout), None, P(y), P(consts), None, None, P(slots), M, C, 2, s)),
            ("bwd reduce mask=bits", 2 * 2 * n + n // 8,
             lambda: H._bn_bwd_reduce(P(dout), P(bits), P(y), P(consts), None, None, P(slots), M, C, 3, s)),
            ("bwd apply mask=y", 2 * 3 * n,
             lambda: H._bn_bwd_apply(P(dout), None, P(y), P(coef), P(consts), P(dy), None, None, None, None, n, C, 2, s)),
            ("bwd apply mask=bits", 2 * 3 * n + n // 8,
             lambda: H._bn_bwd_apply(P(dout), P(bits), P(y), P(coef), P(consts), P(dy), None, None, None, None, n, C, 3, s)),
        ]
        for name, nbytes, fn in cases:
            us = timed(fn, args.iters)
            tot += us
            print(f"{hw}x{hw}x{C:<10}{name:<28}{us:>10.1f}{nbytes / us / 1e3:>9.0f}", flush=True)
        del y, ident, dout, out, dy, dz, bits
        torch.cuda.empty_cache()
    print(f"sum {tot / 1e3:.2f} ms")


if __name__ == "__main__":
    main()
