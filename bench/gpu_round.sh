#!/bin/bash
# One GPU call: tests, the A/B of the fused BatchNorm->1x1 path and the three benchmark models.
# Every GPU step has its own time limit; the script stops at the first failure.
set -o pipefail
out=gpurun_out/round
mkdir -p $out
(while true; do date +%T >> $out/heartbeat.txt; sleep 45; done) & HB=$!
trap "kill $HB" EXIT
step() { name=$1; limit=$2; shift 2; echo "== $name"; timeout -k 10 $limit "$@" > $out/$name.log 2>&1; rc=$?; tail -n 4 $out/$name.log; if [ $rc -ne 0 ]; then echo "FAILED $name rc=$rc"; exit $rc; fi; }
step tests 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread
step r50_fused 200 python bench.py --steps 10 --warmup 3
PDT_FUSE_BN_1X1=0 step r50_unfused 200 python bench.py --steps 10 --warmup 3
step vit 200 python bench.py --model vit_b_16 --steps 8 --warmup 3
step gpt2 200 python bench.py --model gpt2_medium --steps 8 --warmup 3
step pp_check 200 python bench/pp_check.py
echo ALL_OK
