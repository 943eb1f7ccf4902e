timeout -k 10 200 python -u -m pytest tests/test_gpu_ops.py -m gpu -q --timeout 120 --timeout-method thread -k "variant" 2>&1 | tail -2
for t in 13 16; do echo "tile $t"; PDT_GEMM_TILE_WIDE=$t timeout -k 10 120 python bench/gemm_bench.py 2>&1 | grep -v amdgpu.ids; done
