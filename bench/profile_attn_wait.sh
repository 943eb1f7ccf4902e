#!/bin/bash
# Where the waves of the attention kernels spend their time (wait / active counters):  bench/profile_attn_wait.sh NAME SHAPE
set -e
name=$1; shape=$2
R=$PWD
out=$R/gpurun_out/attn_$name
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $out/pmc2 -o pmc -- python3 $R/bench/attn_bench.py "$shape" > $out/pmc2.log 2>&1
cd $R
python3 bench/pmc_raw.py $(ls $out/pmc2/*counter_collection.csv | head -1) attn > $out/wait.txt
cat $out/wait.txt
