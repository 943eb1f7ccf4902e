timeout -k 10 200 python -u -m pytest tests/test_gpu_transformer.py -m gpu -q --timeout 120 --timeout-method thread 2>&1 | tail -2
timeout -k 10 300 python bench.py --model vit_b_16 --batch-size 256 --steps 8 --warmup 3 2>&1 | tail -1 | cut -c1-200
timeout -k 10 300 python bench.py --model gpt2_medium --batch-size 16 --steps 8 --warmup 3 2>&1 | tail -1 | cut -c1-200
