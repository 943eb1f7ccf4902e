"""One launch per configuration of the wide-output 1x1 forward GEMM, for a counter run:
    rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES \
              --kernel-trace --output-format csv -d OUT -o pmc -- python3 bench/c3_pmc.py
    python3 bench/c3_pmc.py --report OUT/pmc_counter_collection.csv
Dispatch order: for each of the four layers, statistics flavour then plain flavour."""
import os, sys, csv, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LAYERS = [(64, 256, 56), (128, 512, 28), (256, 1024, 14), (512, 2048, 7)]
B = 4000
if len(sys.argv) > 2 and sys.argv[1] == "--report":
    rows = collections.defaultdict(dict)
    for r in csv.DictReader(open(sys.argv[2])):
        if "gather_gemm_ring" in r["Kernel_Name"]:
            rows[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
    ids = sorted(rows)
    k = 0
    for cin, cout, hw in LAYERS:
        m = B * hw * hw
        tiles = -(-m // 256) * (cout // 256)
        for label in ("stats", "plain"):
            v = rows[ids[k]]; k += 1
            tw = tiles * 8.0  # tile-waves
            print(f"{cin:4d}->{cout:4d} @{hw:2d} {label}: per tile and wave: VALU {v['SQ_INSTS_VALU'] / tw:7.0f}  SALU {v['SQ_INSTS_SALU'] / tw:6.0f}  LDS {v['SQ_INSTS_LDS'] / tw:5.0f}"
                  f"  VMEM wr {v['SQ_INSTS_VMEM_WR'] / tw:5.1f} rd {v['SQ_INSTS_VMEM_RD'] / tw:5.1f}  MFMA busy {100 * v['SQ_VALU_MFMA_BUSY_CYCLES'] / (32 * v['SQ_BUSY_CYCLES']):5.1f} % of SIMD cycles")
    sys.exit(0)
import torch
from pytorch_distributed_training_amd.ops import hip
dev = torch.device("cuda", 0)
for cin, cout, hw in LAYERS:
    m = B * hw * hw
    a = torch.randn(m, cin, device=dev).to(torch.bfloat16)
    w = (torch.randn(cout, cin, device=dev) * 0.05).to(torch.bfloat16)
    o = torch.empty(m, cout, device=dev, dtype=torch.bfloat16)
    stats = torch.zeros(hip.STAT_SLOTS + 1, 2, cout, device=dev)
    pivot = torch.zeros(cout, device=dev)
    for st in (stats, None):
        hip._launch_gather_gemm(a, w, o, st, None, m, cout, cin, cout, cin, [(0, 0, 0)], 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, pivot=pivot if st is not None else None)
    torch.cuda.synchronize()
    del a, w, o
