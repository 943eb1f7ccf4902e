"""Raw throughput of the MFMA GEMM core (the convolution kernel with a single tap) and of the weight-gradient
kernel (dW[N][K] = dY[M][N]^T X[M][K]) on square and skinny problems, random bf16 operands.

    python bench/gemm_bench.py [--wgrad-only | --tile-sweep]
"""
from __future__ import annotations

import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

SHAPES = [(4096, 4096, 4096), (8192, 8192, 8192), (50176, 256, 2304), (50176, 1024, 256), (50176, 256, 1024),
          (802816, 64, 576), (802816, 256, 64), (200704, 128, 1152), (12544, 512, 4608), (12544, 2048, 512),
          (8192, 3072, 768), (8192, 50304, 1024)]


WGRAD_SHAPES = [(50432, 768, 768), (50432, 2304, 768), (50432, 3072, 768), (50432, 768, 3072), (16384, 1024, 1024),
                (16384, 3072, 1024), (16384, 4096, 1024), (16384, 1024, 4096), (16384, 50304, 1024), (401408, 512, 128), (100352, 1024, 256)]


def _time(run):
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    ts = []
    for _ in range(10):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        run()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e-3)
    return statistics.median(ts)


def wgrad_section(hip, dev):
    print(f"weight gradient\n{'M':>8}{'N':>7}{'K':>6}{'us':>10}{'TF/s':>8}")
    for m, n, k in WGRAD_SHAPES:
        dy = torch.randn(m, n, device=dev).to(torch.bfloat16)
        x = torch.randn(m, k, device=dev).to(torch.bfloat16)
        dw = torch.zeros(n, k, device=dev)
        t = _time(lambda: hip._launch_wgrad(dy, x, dw, m, n, n, k, [(0, 0, 0)], k, 1, 1, 1, 1, 1))
        print(f"{m:8d}{n:7d}{k:6d}{t * 1e6:10.1f}{2.0 * m * n * k / t / 1e12:8.1f}")
        del dy, x, dw


TILE_SWEEP_SHAPES = [(50176, 256, 2304), (50176, 256, 1024), (50176, 1024, 256), (12544, 512, 4608), (12544, 2048, 512),
                     (50432, 768, 768), (50432, 2304, 768), (50432, 3072, 768), (50432, 768, 3072), (201728, 768, 768),
                     (201728, 3072, 768), (16384, 1024, 1024), (16384, 3072, 1024), (16384, 4096, 1024), (16384, 1024, 4096),
                     (32768, 1024, 1024), (32768, 3072, 1024), (32768, 4096, 1024), (32768, 1024, 4096), (8192, 3072, 768)]


def tile_sweep(hip, dev):
    """Forward GEMM under each wide ring tile (256x128 = 13, 256x256 = 16) next to the tile the shape rule picks."""
    print(f"{'M':>8}{'N':>7}{'K':>6}{'us@13':>9}{'us@16':>9}{'picked':>8}")
    for m, n, k in TILE_SWEEP_SHAPES:
        a = torch.randn(m, k, device=dev).to(torch.bfloat16)
        b = torch.randn(n, k, device=dev).to(torch.bfloat16)
        out = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
        picked = hip._pick_tile(m, n, k // 64)
        ts = []
        for tile in (13, 16):
            hip._TILE_WIDE = tile
            ts.append(_time(lambda: hip._launch_gather_gemm(a, b, out, None, None, m, n, k, n, k, [(0, 0, 0)], 1, 1, 1, 1, 1, 1,
                                                            1, 1, 0, 0)))
        hip._TILE_WIDE = -1
        print(f"{m:8d}{n:7d}{k:6d}{ts[0] * 1e6:9.1f}{ts[1] * 1e6:9.1f}{picked:8d}")
        del a, b, out


def main():
    from pytorch_distributed_training_amd.ops import hip

    dev = torch.device("cuda", 0)
    if "--wgrad-only" in sys.argv:
        return wgrad_section(hip, dev)
    if "--tile-sweep" in sys.argv:
        return tile_sweep(hip, dev)
    print(f"{'M':>8}{'N':>7}{'K':>6}{'us':>10}{'TF/s':>8}")
    for m, n, k in SHAPES:
        a = torch.randn(m, k, device=dev).to(torch.bfloat16)
        b = torch.randn(n, k, device=dev).to(torch.bfloat16)
        out = torch.empty(m, n, device=dev, dtype=torch.bfloat16)

        def run():
            hip._launch_gather_gemm(a, b, out, None, None, m, n, k, n, k, [(0, 0, 0)], 1, 1, 1, 1, 1, 1, 1, 1, 0, 0)

        for _ in range(3):
            run()
        torch.cuda.synchronize()
        ts = []
        for _ in range(10):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            run()
            e1.record()
            e1.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e-3)
        t = statistics.median(ts)
        print(f"{m:8d}{n:7d}{k:6d}{t * 1e6:10.1f}{2.0 * m * n * k / t / 1e12:8.1f}")
        del a, b, out
    wgrad_section(hip, dev)


if __name__ == "__main__":
    main()
