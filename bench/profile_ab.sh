#!/bin/bash
# Kernel-time profiles of the working tree and of ./ab_prev on one box: bench/profile_ab.sh [bench.py arguments]
set -e
R=$PWD
export TMPDIR=/tmp
for side in prev new; do
  if [ $side = prev ]; then dir=$R/ab_prev; else dir=$R; fi
  mkdir -p $R/gpurun_out/prof_ab_$side
  ( cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_ab_$side -o run -- python3 $dir/bench.py --steps 4 --warmup 1 --no-extras "$@" > $R/gpurun_out/prof_ab_$side/out.log 2>&1 )
  python3 bench/kernel_stats.py gpurun_out/prof_ab_$side/run_results.db 7 > gpurun_out/prof_ab_$side/summary.txt
  rm -f gpurun_out/prof_ab_$side/run_results.db
done
python3 - <<'PY'
import re
def load(f):
    d={}
    for ln in open(f).read().splitlines()[1:]:
        m=re.match(r"\s*([\d.]+) ms\s+[\d.]+%\s+(\d+)\s+(.*)", ln)
        if m: d[m.group(3)]=(float(m.group(1)), int(m.group(2)))
    return d
a=load("gpurun_out/prof_ab_prev/summary.txt"); b=load("gpurun_out/prof_ab_new/summary.txt")
print(open("gpurun_out/prof_ab_prev/summary.txt").readline().strip()); print(open("gpurun_out/prof_ab_new/summary.txt").readline().strip())
for k in sorted(set(a)|set(b), key=lambda k: -abs(b.get(k,(0,0))[0]-a.get(k,(0,0))[0]))[:25]:
    print(f"{a.get(k,(0,0))[0]:8.3f} -> {b.get(k,(0,0))[0]:8.3f}  {k[:120]}")
PY
