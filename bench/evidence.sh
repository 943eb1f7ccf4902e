#!/bin/bash
# One call that collects the measurement summaries kept under profiles/: kernel-time profiles of the three benchmark models, where the collectives of a
# step sit relative to its compute kernels (one-rank collectives kept, 32 resident occupants per bucket at 150 GB/s),
# and the cost of deterministic mode.  Every GPU step under its own time limit; stops at the first failure.
set -e
R=$PWD
export TMPDIR=/tmp
mkdir -p gpurun_out/evidence
bash bench/profile_step.sh resnet50 > gpurun_out/evidence/prof_resnet50.log 2>&1
bash bench/profile_step.sh vit_b_16 --model vit_b_16 > gpurun_out/evidence/prof_vit.log 2>&1
bash bench/profile_step.sh gpt2_medium --model gpt2_medium > gpurun_out/evidence/prof_gpt2.log 2>&1
rm -f gpurun_out/prof_*/run_results.db
mkdir -p gpurun_out/evidence/overlap
( cd /tmp && PDT_SELF_ALLREDUCE=1 PDT_REHEARSE_CHANNELS=32 PDT_REHEARSE_GBPS=150 PDT_REHEARSE_HEAVY=1 timeout -k 10 400 rocprofv3 --kernel-trace -d $R/gpurun_out/evidence/overlap -o run -- python3 $R/bench.py --steps 4 --warmup 1 --no-extras --bucket-mb 25 > $R/gpurun_out/evidence/overlap/out.log 2>&1 )
python3 bench/overlap_report.py gpurun_out/evidence/overlap/run_results.db > gpurun_out/evidence/overlap_report.txt
rm -f gpurun_out/evidence/overlap/run_results.db
timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-extras > gpurun_out/evidence/bench_plain.json 2> gpurun_out/evidence/bench_plain.err
PDT_DETERMINISTIC=1 timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-extras > gpurun_out/evidence/bench_det.json 2> gpurun_out/evidence/bench_det.err
tail -1 gpurun_out/evidence/bench_plain.json | cut -c1-200; tail -1 gpurun_out/evidence/bench_det.json | cut -c1-200
cat gpurun_out/evidence/overlap_report.txt
