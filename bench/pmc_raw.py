"""Sums of the counters of a rocprofv3 --pmc CSV per kernel, as shares of SQ_WAVE_CYCLES where that makes sense.

    python bench/pmc_raw.py out/pmc_counter_collection.csv [kernel-name filter]
"""
import collections
import csv
import re
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(sys.argv[1])):
    k = re.sub(r"\(.*", "", r["Kernel_Name"])
    if len(sys.argv) > 2 and sys.argv[2] not in k:
        continue
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in agg.items():
    wc = v.get("SQ_WAVE_CYCLES", 0.0)
    print(k[:60])
    for c, x in sorted(v.items()):
        print(f"    {c:28s} {x:16.0f}" + (f"  {100 * x / wc:6.1f} % of wave cycles" if wc and c != "SQ_WAVE_CYCLES" else ""))
