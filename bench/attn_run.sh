set -o pipefail
mkdir -p gpurun_out/attn
timeout -k 10 120 python bench/attn_check.py > gpurun_out/attn/check_new.txt 2>&1; rc=$?; grep -v Warn gpurun_out/attn/check_new.txt | tail -2
[ $rc -ne 0 ] && exit $rc
timeout -k 10 120 python bench/attn_bench.py > gpurun_out/attn/bench_new.txt 2>&1 && cat gpurun_out/attn/bench_new.txt && bash bench/profile_attn.sh gpt2c gpt2 | tail -6 | cut -c1-120
