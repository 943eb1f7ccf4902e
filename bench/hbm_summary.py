"""Per-kernel HBM traffic from two rocprofv3 counter passes (see bench/profile_hbm.sh).

    python bench/hbm_summary.py STEPS rd_counter_collection.csv wr_counter_collection.csv rd_kernel_trace.csv

Bytes that crossed the L2 <-> memory interface: read requests by size (TCC_EA0_RDREQ_{32B,64B,128B}; the remainder of
TCC_EA0_RDREQ is priced at 64 bytes), write requests of 64 bytes (TCC_EA0_WRREQ_64B) and the rest at 32 bytes.
Per kernel name: dispatches per step, time per step
(from the kernel trace of the read pass; counter collection serialises kernels, so the times are those of kernels
running alone), GB read / written per step, achieved TB/s = (read + written) / time, and the L2 hit rate.
"""
from __future__ import annotations

import collections
import csv
import re
import sys


def short(name: str) -> str:
    name = re.sub(r"\(.*", "", name)
    name = name.replace("void ", "")
    return name[:86]


def main(steps: int, rd_path: str, wr_path: str, trace_path: str) -> None:
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for path in (rd_path, wr_path):
        for r in csv.DictReader(open(path)):
            agg[short(r["Kernel_Name"])][r["Counter_Name"]] += float(r["Counter_Value"])
    t_ns = collections.defaultdict(float)
    n = collections.defaultdict(int)
    for r in csv.DictReader(open(trace_path)):
        k = short(r["Kernel_Name"])
        t_ns[k] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        n[k] += 1
    rows = []
    for k, v in agg.items():
        r32, r64, r128 = (v.get(f"TCC_EA0_RDREQ_{b}B_sum", 0.0) for b in (32, 64, 128))
        rd = (32 * r32 + 128 * r128 + 64 * (v.get("TCC_EA0_RDREQ_sum", 0.0) - r32 - r128)) / steps
        w64 = v.get("TCC_EA0_WRREQ_64B_sum", 0.0)
        wr = (64 * w64 + 32 * (v.get("TCC_EA0_WRREQ_sum", 0.0) - w64)) / steps
        ms = t_ns.get(k, 0.0) / 1e6 / steps
        hit, miss = v.get("TCC_HIT_sum", 0.0), v.get("TCC_MISS_sum", 0.0)
        rows.append((ms, k, n.get(k, 0) / steps, rd, wr, hit / (hit + miss) if hit + miss else 0.0))
    rows.sort(reverse=True)
    tot_ms = sum(r[0] for r in rows)
    tot_rd = sum(r[3] for r in rows)
    tot_wr = sum(r[4] for r in rows)
    print(f"# per step: {tot_ms:.2f} ms of kernels (run one at a time), {tot_rd / 1e9:.1f} GB read + {tot_wr / 1e9:.1f} GB written "
          f"= {(tot_rd + tot_wr) / 1e9:.1f} GB, {(tot_rd + tot_wr) / 1e9 / max(tot_ms, 1e-9):.2f} TB/s on average")
    print(f"{'ms/step':>8s} {'calls':>6s} {'GB rd':>7s} {'GB wr':>7s} {'TB/s':>6s} {'L2hit%':>6s}  kernel")
    for ms, k, calls, rd, wr, hr in rows:
        if ms < 0.05:
            continue
        tbs = (rd + wr) / 1e9 / ms if ms else 0.0
        print(f"{ms:8.3f} {calls:6.1f} {rd / 1e9:7.2f} {wr / 1e9:7.2f} {tbs:6.2f} {100 * hr:6.1f}  {k}")


if __name__ == "__main__":
    main(int(sys.argv[1]), sys.argv[2], sys.argv[3], sys.argv[4])
