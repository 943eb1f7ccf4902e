cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_vit -o vit -- python bench.py --model vit_b_16 --batch-size 256 --steps 3 --warmup 2 > gpurun_out/prof_vit.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_gpt -o gpt -- python bench.py --model gpt2_medium --batch-size 16 --steps 3 --warmup 2 > gpurun_out/prof_gpt.log 2>&1
