#!/bin/bash
# Throughput of bench.py over per-GPU batch sizes: bench/batch_sweep.sh MODEL B1 B2 ...   (one line per batch)
model=$1; shift
for b in "$@"; do
  timeout -k 10 300 python3 bench.py --model $model --batch-size $b --steps 8 --warmup 3 --no-extras 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$model', $b, d['ms_per_step'], d['value'])" || exit 1
done
