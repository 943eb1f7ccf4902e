#!/bin/bash
# The wide-output 1x1 layers (conv3 of every bottleneck) under other tile choices: bench/c3_variants.sh
for v in "" "PDT_GEMM_PP=2" "PDT_GEMM_TILE_WIDE=13" "PDT_GEMM_TILE_WIDE=0" "PDT_GEMM_TILE_WIDE=6"; do
  echo "== ${v:-default}"
  env $v timeout -k 10 200 python bench/conv_bench.py --batch 4000 --iters 6 --only .c3 2>/dev/null | grep -E "fwd|dgrad" || exit 1
done
