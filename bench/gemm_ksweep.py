import os, sys, statistics
sys.path.insert(0, "/root/repo")
import torch
from pytorch_distributed_training_amd.ops import hip
dev = torch.device("cuda", 0)
def t(f):
    for _ in range(5): f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(12):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) * 1e-3)
    return statistics.median(ts)
hip._GEMM_PP = 1
for m, n in [(32768, 4096), (65536, 3072)]:
    tiles_per_cu = (m // 256) * (n // 256) / 256
    for k in (512, 768, 1024, 2048, 4096, 8192):
        a = torch.randn(m, k, device=dev).to(torch.bfloat16)
        b = torch.randn(n, k, device=dev).to(torch.bfloat16)
        o = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
        row = []
        for nt in (64 << 20, -1, -2):
            hip._lib.pdt_set_out_nt_bytes(nt)
            f = lambda: hip._launch_gather_gemm(a, b, o, None, None, m, n, k, n, k, [(0, 0, 0)], 1, 1, 1, 1, 1, 1, 1, 1, 0, 0)
            row.append(t(f) * 1e6)
        hip._lib.pdt_set_out_nt_bytes(64 << 20)
        fl = 2.0 * m * n * k
        print(f"{m:7d}{n:6d}{k:6d}  store {row[0]:8.1f} us {fl/row[0]/1e6:7.0f} TF/s  per tile {row[0]/tiles_per_cu:6.2f} us | nostore {row[1]:8.1f} us per tile {row[1]/tiles_per_cu:6.2f} | no epilogue {row[2]:8.1f} us per tile {row[2]/tiles_per_cu:6.2f}", flush=True)
        del a, b, o
