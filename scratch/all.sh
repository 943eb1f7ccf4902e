timeout -k 10 300 python -u -m pytest tests -m gpu -q --timeout 120 --timeout-method thread > gpurun_out/t.log 2>&1; tail -2 gpurun_out/t.log | cut -c1-200; grep -n "^E" gpurun_out/t.log | head -5
timeout -k 10 300 python bench.py --steps 6 --warmup 3 2>&1 | tail -1 | cut -c1-180
timeout -k 10 300 python bench.py --model vit_b_16 --batch-size 256 --steps 8 --warmup 3 2>&1 | tail -1 | cut -c1-200
timeout -k 10 300 python bench.py --model gpt2_medium --batch-size 16 --steps 8 --warmup 3 2>&1 | tail -1 | cut -c1-200
