#!/bin/bash
# Same-box A/B of the working tree against a copy of an earlier build placed under ./ab_prev (bench.py + package with
# its own built libraries): alternating runs, ms/step of each.  Usage: bench/ab_prev.sh [rounds] [bench.py arguments]
rounds=${1:-3}; shift
for r in $(seq $rounds); do
  for side in prev new; do
    if [ $side = prev ]; then dir=ab_prev; else dir=.; fi
    ( cd $dir && timeout -k 10 300 python bench.py --no-extras "$@" 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$side', d['ms_per_step'], d['value'])" )
  done
done
