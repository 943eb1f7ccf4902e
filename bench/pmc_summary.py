"""Per-kernel summary of a rocprofv3 `--pmc ... --output-format csv` run.

    rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS \
              SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 --kernel-trace \
              --output-format csv -d out -o pmc -- python bench.py --batch-size 256 --steps 2 --warmup 1
    python bench/pmc_summary.py out/pmc_counter_collection.csv > profiles/<name>.csv

Counters are summed over all dispatches of a kernel.  Derived columns: MFMA pipe busy cycles per SIMD-cycle
(SQ_BUSY_CYCLES is per shader engine: 32 of them hold 1024 SIMDs, so SIMD-cycles = 32 x SQ_BUSY_CYCLES), vector
ALU instructions per MFMA instruction (an MFMA of 16x16x32 bf16 is 16 busy cycles), LDS bank conflict share.
"""
from __future__ import annotations

import collections
import csv
import re
import sys


def main(path: str) -> None:
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"])
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k].add(r["Dispatch_Id"])
    names = sorted({c for v in agg.values() for c in v})
    w = csv.writer(sys.stdout)
    w.writerow(["kernel", "dispatches"] + names + ["mfma_busy_pct_of_simd_cycles", "valu_insts_per_mfma", "lds_conflict_pct_of_lds_active"])
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", 0.0)):
        busy = v.get("SQ_BUSY_CYCLES", 0.0) * 32.0
        mf = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        mfma_pct = 100.0 * mf / busy if busy else 0.0
        per = v.get("SQ_INSTS_VALU", 0.0) / (mf / 16.0) if mf else 0.0
        lds = 100.0 * v.get("SQ_LDS_BANK_CONFLICT", 0.0) / v["SQ_LDS_IDX_ACTIVE"] if v.get("SQ_LDS_IDX_ACTIVE") else 0.0
        w.writerow([k[:90], len(disp[k])] + [f"{v.get(c, 0.0):.0f}" for c in names] + [f"{mfma_pct:.1f}", f"{per:.1f}", f"{lds:.1f}"])


if __name__ == "__main__":
    main(sys.argv[1])
