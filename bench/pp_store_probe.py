"""What the output stores cost the 256x256 GEMM launches: time with and without them (PDT_OUT_NT_MB=-1 skips the store)."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pytorch_distributed_training_amd.ops import hip
dev = torch.device("cuda", 0)
def t(f):
    for _ in range(3): f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(10):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) * 1e-3)
    return statistics.median(ts)
for m, n, k in [(32768, 4096, 1024), (65536, 3072, 768), (16384, 1024, 1024), (8192, 8192, 8192), (50176, 1024, 256)]:
    a = torch.randn(m, k, device=dev).to(torch.bfloat16)
    b = torch.randn(n, k, device=dev).to(torch.bfloat16)
    o = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
    row = []
    for pp in (0, 1):
        for nt in (64 << 20, 1 << 60, -1):
            hip._GEMM_PP = pp
            hip._lib.pdt_set_out_nt_bytes(nt)
            f = lambda: hip._launch_gather_gemm(a, b, o, None, None, m, n, k, n, k, [(0, 0, 0)], 1, 1, 1, 1, 1, 1, 1, 1, 0, 0)
            row.append(t(f) * 1e6)
    hip._lib.pdt_set_out_nt_bytes(64 << 20)
    print(f"{m:7d}{n:6d}{k:6d}  tile16: nt {row[0]:8.1f} plain {row[1]:8.1f} nostore {row[2]:8.1f} | pp: nt {row[3]:8.1f} plain {row[4]:8.1f} nostore {row[5]:8.1f} us", flush=True)
