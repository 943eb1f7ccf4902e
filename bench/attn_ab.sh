# same-box A/B of the attention kernels: the library built from the tree against _lib/libpdt_kernels_prev.so
set -o pipefail
mkdir -p gpurun_out/attn
timeout -k 10 120 python bench/attn_check.py > gpurun_out/attn/check_new.txt 2>&1; rc=$?; grep -v Warn gpurun_out/attn/check_new.txt | tail -2
[ $rc -ne 0 ] && exit $rc
echo "== new" && timeout -k 10 120 python bench/attn_bench.py > gpurun_out/attn/bench_new.txt 2>&1 && cat gpurun_out/attn/bench_new.txt || exit 1
if [ -f pytorch_distributed_training_amd/_lib/libpdt_kernels_prev.so ]; then
  echo "== prev" && PDT_KERNELS_LIB=libpdt_kernels_prev.so timeout -k 10 120 python bench/attn_bench.py > gpurun_out/attn/bench_prev.txt 2>&1 && cat gpurun_out/attn/bench_prev.txt || exit 1
  echo "== new again" && timeout -k 10 120 python bench/attn_bench.py > gpurun_out/attn/bench_new2.txt 2>&1 && cat gpurun_out/attn/bench_new2.txt || exit 1
fi
timeout -k 10 300 python -u -m pytest tests/test_gpu_transformer.py -x -q --timeout 120 --timeout-method thread 2>&1 | tail -3
