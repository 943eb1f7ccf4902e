"""Per-kernel time of a rocprofv3 run, normalised per training step.

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --stats -d OUT -o run -- python /root/repo/bench.py --steps 4 --warmup 1
    python bench/kernel_stats.py OUT/run_results.db 7        # 4 timed + 1 warmup + 2 setup steps

Reads the rocpd SQLite database rocprofv3 writes, sums dispatch durations by kernel name and prints the
kernels by share of GPU time (milliseconds divided by the step count given as the second argument).
"""

from __future__ import annotations

import re
import sqlite3
import sys


def main() -> None:
    if len(sys.argv) < 2:
        sys.exit(__doc__)
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    kd = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch"))
    ks = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
    rows = cur.execute(
        f"select s.kernel_name, count(*), sum(d.end - d.start) from {kd} d join {ks} s on d.kernel_id = s.id "
        "group by s.kernel_name order by 3 desc"
    ).fetchall()
    tot = sum(r[2] for r in rows)
    div = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    print(f"total kernel time {tot / 1e6:.2f} ms over all steps; /{div:g} = {tot / 1e6 / div:.2f} ms")
    for name, count, t in rows[:45]:
        name = re.sub(r"\(.*", "", name)[:110]
        print(f"{t / 1e6 / div:9.3f} ms {100 * t / tot:5.1f}% {count:6d}  {name}")


if __name__ == "__main__":
    main()
