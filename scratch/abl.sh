for a in 0 1 2 3; do echo "ablate=$a"; PDT_ABLATE=$a PDT_GEMM_TILE_WIDE=15 timeout -k 10 120 python bench/gemm_bench.py 2>&1 | grep -E "^ +8192 +8192|^ +8192 +3072"; done
