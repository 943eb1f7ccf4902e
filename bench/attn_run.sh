set -o pipefail
mkdir -p gpurun_out/attn
timeout -k 10 120 python bench/attn_check.py > gpurun_out/attn/check_new.txt 2>&1; rc=$?; grep -v Warn gpurun_out/attn/check_new.txt | tail -2
[ $rc -ne 0 ] && exit $rc
timeout -k 10 120 python bench/attn_bench.py > gpurun_out/attn/bench_new.txt 2>&1 && cat gpurun_out/attn/bench_new.txt &&
timeout -k 10 300 python -u -m pytest tests/test_gpu_transformer.py tests/test_gpu_e2e.py -x -q --timeout 200 --timeout-method thread 2>&1 | tail -3 &&
timeout -k 10 200 python bench.py --model gpt2_medium --steps 8 --warmup 3 --no-extras 2>&1 | tail -1 | cut -c1-420 &&
timeout -k 10 200 python bench.py --model gpt2_medium --batch-size 128 --steps 4 --warmup 2 --no-extras 2>&1 | tail -1 | cut -c1-420 &&
timeout -k 10 200 python bench.py --model vit_b_16 --steps 8 --warmup 3 --no-extras 2>&1 | tail -1 | cut -c1-420
