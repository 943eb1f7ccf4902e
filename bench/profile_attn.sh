#!/bin/bash
# Per-kernel time and MFMA / LDS counters of the attention kernels on one shape:  bench/profile_attn.sh NAME SHAPE
# (SHAPE: substring of a name in bench/attn_bench.py).  Two runs: kernel trace, then one counter pass.
set -e
name=$1; shape=$2
R=$PWD
out=$R/gpurun_out/attn_$name
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $out -o run -- python3 $R/bench/attn_bench.py "$shape" > $out/out.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 --kernel-trace --output-format csv -d $out/pmc -o pmc -- python3 $R/bench/attn_bench.py "$shape" > $out/pmc.log 2>&1
cd $R
python3 bench/kernel_stats.py $out/run_results.db 1 | grep -i "attn\|total" > $out/summary.txt
python3 bench/pmc_summary.py $(ls $out/pmc/*counter_collection.csv | head -1) | grep -i "attn\|kernel" >> $out/summary.txt
cat $out/out.log | tail -2
cat $out/summary.txt | cut -c1-250
