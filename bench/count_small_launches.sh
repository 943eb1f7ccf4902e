#!/bin/bash
# How many copy / fill launches does a training step carry?  Two kernel traces of bench.py that differ only in the number
# of steps: the difference in launch counts divided by the difference in steps is the per-step number; the rest is setup.
set -e
R=$PWD
out=$R/gpurun_out/small_launches
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
for steps in 4 12; do
  timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $out/s$steps -o run -- python3 $R/bench.py --steps $steps --warmup 1 --no-extras "$@" > $out/s$steps.log 2>&1
done
cd $R
python3 - "$out" <<'PY'
import re, sqlite3, sys
def counts(path):
    db = sqlite3.connect(path); cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    kd = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch")); ks = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
    return {re.sub(r"\(.*", "", n)[:70]: (c, t) for n, c, t in cur.execute(f"select s.kernel_name, count(*), sum(d.end - d.start) from {kd} d join {ks} s on d.kernel_id = s.id group by s.kernel_name")}
a, b = counts(sys.argv[1] + "/s4/run_results.db"), counts(sys.argv[1] + "/s12/run_results.db")
print(f"{'kernel':70s} {'per step':>9} {'setup':>7} {'us per step':>12}")
tot = 0
for k in sorted(b, key=lambda k: -(b[k][0] - a.get(k, (0, 0))[0])):
    per = (b[k][0] - a.get(k, (0, 0))[0]) / 8.0
    setup = b[k][0] - per * 15  # 12 timed + 1 warm-up + 2 set-up steps
    us = (b[k][1] - a.get(k, (0, 0))[1]) / 8.0 / 1e3
    if re.search(r"copy|fill|memset|elementwise|Memcpy", k, re.I) or per != int(per):
        print(f"{k:70s} {per:9.2f} {setup:7.0f} {us:12.1f}")
    tot += per
print("launches per step, all kernels:", tot)
PY
