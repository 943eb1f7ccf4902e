"""Where a wide-output 1x1 forward launch spends its time: with / without statistics, with the output stores removed,
with the whole epilogue removed (measurement aids of pdt_set_out_nt_bytes).   python bench/c3_breakdown.py [batch]"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pytorch_distributed_training_amd.ops import hip
dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
def t(f):
    for _ in range(3): f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(8):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    return statistics.median(ts)
for cin, cout, hw in [(64, 256, 56), (128, 512, 28), (256, 1024, 14), (512, 2048, 7)]:
    m = B * hw * hw
    a = torch.randn(m, cin, device=dev).to(torch.bfloat16)
    w = (torch.randn(cout, cin, device=dev) * 0.05).to(torch.bfloat16)
    o = torch.empty(m, cout, device=dev, dtype=torch.bfloat16)
    stats = torch.zeros(hip.STAT_SLOTS + 1, 2, cout, device=dev)
    pivot = torch.zeros(cout, device=dev)
    row = f"{cin:5d}->{cout:5d} @{hw:2d} tiles/CU {(-(-m // 256)) * (cout // 256) / 256:6.1f} |"
    for label, st in (("stats", stats), ("plain", None)):
        for nt, nm in ((64 << 20, "full"), (-1, "nostore"), (-2, "noepi")):
            hip._lib.pdt_set_out_nt_bytes(nt)
            f = lambda: hip._launch_gather_gemm(a, w, o, st, None, m, cout, cin, cout, cin, [(0, 0, 0)], 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, pivot=pivot if st is not None else None)
            row += f" {label}/{nm} {t(f):7.1f}"
        row += " |"
    hip._lib.pdt_set_out_nt_bytes(64 << 20)
    print(row, flush=True)
    del a, w, o
