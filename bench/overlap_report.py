"""Where the gradient collectives of a step run relative to its compute kernels, from a rocprofv3 kernel trace.

    cd /tmp && export TMPDIR=/tmp
    PDT_SELF_ALLREDUCE=1 rocprofv3 --kernel-trace -d OUT -o run -- python3 /root/repo/bench.py --steps 4 --warmup 1 --no-extras
    python bench/overlap_report.py OUT/run_results.db

A step ends with the optimizer kernel (`multi_tensor_step_kernel`).  Every kernel that is not one of this
repository's compute kernels and runs on another hardware queue than the step's is taken for communication work
(RCCL's device kernels, the bf16 wire pack / unpack kernels, the rehearsal occupants).  For each step the report
gives the span of the backward pass, the number of communication kernels, how much of their time lay under a
compute kernel of the step (hidden) and how much did not (exposed), and the gap between the end of the last
communication kernel and the start of the optimizer kernel.
"""

from __future__ import annotations

import bisect
import re
import sqlite3
import sys

COMM = re.compile(r"nccl|rccl|resident_|pack_bf16_kernel|count_kernel", re.I)


def main() -> None:
    if len(sys.argv) < 2:
        sys.exit(__doc__)
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    kd = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch"))
    ks = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
    cols = [r[1] for r in cur.execute(f"pragma table_info({kd})")]
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
    q = f"select s.kernel_name, d.start, d.end, {('d.' + qcol) if qcol else '0'} from {kd} d join {ks} s on d.kernel_id = s.id order by d.start"
    rows = cur.execute(q).fetchall()
    if not rows:
        sys.exit("no kernel dispatches in the trace")
    comm = [(s, e, n) for n, s, e, _ in rows if COMM.search(n)]
    comp = [(s, e, n) for n, s, e, _ in rows if not COMM.search(n)]
    queues = sorted({r[3] for r in rows})
    print(f"{len(rows)} dispatches on {len(queues)} queue(s) ({qcol}); {len(comm)} communication kernels, {len(comp)} compute kernels")
    names = {}
    for s, e, n in comm:
        k = re.sub(r"\(.*", "", n)[:70]
        c, t = names.get(k, (0, 0))
        names[k] = (c + 1, t + e - s)
    for k, (c, t) in sorted(names.items(), key=lambda kv: -kv[1][1]):
        print(f"   {c:5d} x {t / c / 1e3:9.1f} us  {k}")
    opt = [(s, e) for s, e, n in comp if "multi_tensor_step_kernel" in n]
    if not opt:
        sys.exit("no optimizer kernel in the trace: cannot delimit steps")
    # merged busy intervals of the compute kernels
    merged = []
    for s, e, _ in comp:
        if merged and s <= merged[-1][1]:
            merged[-1][1] = max(merged[-1][1], e)
        else:
            merged.append([s, e])
    starts = [m[0] for m in merged]

    def hidden(s: int, e: int) -> int:
        i = max(bisect.bisect_right(starts, s) - 1, 0)
        h = 0
        while i < len(merged) and merged[i][0] < e:
            h += max(0, min(e, merged[i][1]) - max(s, merged[i][0]))
            i += 1
        return h

    print(" step   step ms   comm kernels   comm ms   hidden ms   exposed ms   last comm end -> optimizer start (us)")
    prev_end = rows[0][1]
    for k, (os_, oe) in enumerate(opt):
        mine = [(s, e) for s, e, _ in comm if prev_end <= s < oe]
        tot = sum(e - s for s, e in mine)
        hid = sum(hidden(s, e) for s, e in mine)
        gap = (os_ - max(e for _, e in mine)) / 1e3 if mine else float("nan")
        print(f"{k:5d} {(oe - prev_end) / 1e6:9.2f} {len(mine):14d} {tot / 1e6:9.3f} {hid / 1e6:11.3f} {(tot - hid) / 1e6:12.3f} {gap:12.1f}")
        prev_end = oe


if __name__ == "__main__":
    main()
