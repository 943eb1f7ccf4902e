set -o pipefail
mkdir -p gpurun_out/attn
timeout -k 10 120 python bench/attn_check.py > gpurun_out/attn/check_new.txt 2>&1; rc=$?; grep -v Warn gpurun_out/attn/check_new.txt | tail -14
[ $rc -ne 0 ] && [ $rc -ne 1 ] && exit $rc
timeout -k 10 120 python bench/attn_bench.py > gpurun_out/attn/bench_new.txt 2>&1 && cat gpurun_out/attn/bench_new.txt &&
timeout -k 10 300 python -u -m pytest tests/test_gpu_transformer.py tests/test_gpu_e2e.py -x -q --timeout 200 --timeout-method thread 2>&1 | tail -5
