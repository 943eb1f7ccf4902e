#!/bin/bash
# 3x3 layers on the plain ring schedule (default) and on the ping-pong schedule (PDT_GEMM_PP=2): bench/pp_3x3.sh
for v in "PDT_GEMM_PP=1" "PDT_GEMM_PP=2"; do
  echo "== $v"
  env $v timeout -k 10 200 python bench/conv_bench.py --batch 4000 --iters 6 --only c2 2>/dev/null | grep -E "fwd|dgrad|totals" || exit 1
done
