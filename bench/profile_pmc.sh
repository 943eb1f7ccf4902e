#!/bin/bash
# MFMA / LDS counters of bench.py on one GPU:  bench/profile_pmc.sh NAME [bench.py arguments]
# One counter pass (8 SQ counters is what one run can hold); writes gpurun_out/pmc_NAME/ and a per-kernel summary.
set -e
name=$1; shift
R=$PWD
mkdir -p $R/gpurun_out/pmc_$name
export TMPDIR=/tmp
cd /tmp
timeout -s KILL 300 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$name -o pmc -- python3 $R/bench.py --steps 2 --warmup 1 --no-extras "$@" > $R/gpurun_out/pmc_$name/out.log 2>&1
cd $R
python3 bench/pmc_summary.py $(ls gpurun_out/pmc_$name/*counter_collection.csv | head -1) > gpurun_out/pmc_$name/summary.csv
head -12 gpurun_out/pmc_$name/summary.csv | cut -c1-200
