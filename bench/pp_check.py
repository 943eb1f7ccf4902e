import sys, os, statistics
sys.path.insert(0, "/root/repo")
import torch
from pytorch_distributed_training_amd.ops import hip
dev = torch.device("cuda", 0)
torch.manual_seed(0)
def run(tile, a, b, out, m, n, k):
    hip._TILE_WIDE = tile
    hip._GEMM_PP = 0
    hip._launch_gather_gemm(a, b, out, None, None, m, n, k, n, k, [(0, 0, 0)], 1, 1, 1, 1, 1, 1, 1, 1, 0, 0)
    hip._TILE_WIDE = -1
def t(f):
    for _ in range(3): f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(10):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) * 1e-3)
    return statistics.median(ts)
bad = 0
for m, n, k in [(256, 256, 64), (300, 520, 192), (4096, 4096, 4096), (8192, 8192, 8192), (12544, 2048, 512), (50176, 1024, 256), (8192, 50304, 1024), (8192, 3072, 768), (32768, 4096, 1024), (32768, 1024, 4096), (16384, 1024, 1024), (65536, 3072, 768)]:
    a = torch.randn(m, k, device=dev).to(torch.bfloat16)
    b = torch.randn(n, k, device=dev).to(torch.bfloat16)
    o16 = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
    o17 = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
    run(16, a, b, o16, m, n, k)
    for rep in range(5):
        o17.fill_(float("nan"))
        run(17, a, b, o17, m, n, k)
        torch.cuda.synchronize()
        if not torch.equal(o16, o17):
            d = (o16.float() - o17.float()).abs()
            print("MISMATCH", m, n, k, "rep", rep, "max", float(d.max()), "count", int((d > 0).sum() + torch.isnan(o17.float()).sum()))
            bad += 1
            break
    t16 = t(lambda: run(16, a, b, o16, m, n, k))
    t17 = t(lambda: run(17, a, b, o17, m, n, k))
    fl = 2.0 * m * n * k
    print(f"{m:7d}{n:7d}{k:6d}  tile16 {t16*1e6:9.1f} us {fl/t16/1e12:7.1f} TF/s   pp {t17*1e6:9.1f} us {fl/t17/1e12:7.1f} TF/s", flush=True)
    del a, b, o16, o17
print("bad", bad)
