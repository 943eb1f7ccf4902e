#!/bin/bash
# Kernel-time profile of bench.py on one GPU:  bench/profile_step.sh NAME [bench.py arguments]
# Writes gpurun_out/prof_NAME/{run_results.db,out.log,summary.txt}; 4 timed + 1 warmup + 2 setup steps = 7 steps.
set -e
name=$1; shift
R=$PWD
mkdir -p $R/gpurun_out/prof_$name
export TMPDIR=/tmp
cd /tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$name -o run -- python3 $R/bench.py --steps 4 --warmup 1 --no-extras "$@" > $R/gpurun_out/prof_$name/out.log 2>&1
cd $R
python3 bench/kernel_stats.py gpurun_out/prof_$name/run_results.db 7 > gpurun_out/prof_$name/summary.txt
tail -1 gpurun_out/prof_$name/out.log | cut -c1-220
cat gpurun_out/prof_$name/summary.txt
