"""Weight-gradient kernel on the 256x256 tile: plain ring schedule (tile 16) against the ping-pong schedule (tile 17).

Checks both against an fp32 reference (dW = dY^T X and the bias gradient = column sums of dY, on shapes that
include ragged row counts and partial tiles), then times them on the transformer shapes.

    python bench/wgrad_pp_check.py
"""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pytorch_distributed_training_amd.ops import hip  # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(0)


def run(pp, dy, x, dw, db, m, n, k):
    hip.WGRAD_RING = 16
    hip.WGRAD_PP = pp
    hip._launch_wgrad(dy, x, dw, m, n, n, k, [(0, 0, 0)], k, 1, 1, 1, 1, 1, dbias=db)
    hip.WGRAD_RING = 1


def t(f):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(10):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        f()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e-3)
    return statistics.median(ts)


bad = 0
CHECK = [(256, 256, 256), (1000, 256, 256), (4133, 520, 328), (16384, 1024, 1024), (20000, 768, 3072)]
for m, n, k in CHECK:
    dy = torch.randn(m, n, device=dev).to(torch.bfloat16)
    x = torch.randn(m, k, device=dev).to(torch.bfloat16)
    ref = dy.float().t() @ x.float()
    refb = dy.float().sum(0)
    for pp in (0, 1):
        for rep in range(3):
            dw = torch.zeros(n, k, device=dev)
            db = torch.zeros(n, device=dev)
            run(pp, dy, x, dw, db, m, n, k)
            torch.cuda.synchronize()
            e = float((dw - ref).abs().max() / ref.abs().max())
            eb = float((db - refb).abs().max() / refb.abs().max())
            if not (e < 2e-3 and eb < 2e-3):
                print("MISMATCH", (m, n, k), "pp", pp, "rep", rep, "rel err", e, "bias", eb)
                bad += 1
                break
    print(f"checked {m} {n} {k}", flush=True)
print("bad", bad, flush=True)

SHAPES = [(131072, 1024, 1024), (131072, 3072, 1024), (131072, 4096, 1024), (131072, 1024, 4096), (32768, 50304, 1024),
          (201728, 768, 768), (201728, 2304, 768), (201728, 3072, 768), (201728, 768, 3072),
          (401408, 512, 256), (100352, 1024, 512), (100352, 512, 1024), (25088, 2048, 1024)]
for m, n, k in SHAPES:
    dy = torch.randn(m, n, device=dev).to(torch.bfloat16)
    x = torch.randn(m, k, device=dev).to(torch.bfloat16)
    dw = torch.zeros(n, k, device=dev)
    t16 = t(lambda: run(0, dy, x, dw, None, m, n, k))
    t17 = t(lambda: run(1, dy, x, dw, None, m, n, k))
    fl = 2.0 * m * n * k
    print(f"{m:7d}{n:7d}{k:6d}  ring {t16 * 1e6:9.1f} us {fl / t16 / 1e12:7.1f} TF/s   ping-pong {t17 * 1e6:9.1f} us {fl / t17 / 1e12:7.1f} TF/s",
          flush=True)
    del dy, x, dw
