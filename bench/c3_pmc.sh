#!/bin/bash
set -e
R=$PWD
mkdir -p $R/gpurun_out/c3_pmc
export TMPDIR=/tmp
cd /tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/c3_pmc -o pmc -- python3 $R/bench/c3_pmc.py > $R/gpurun_out/c3_pmc/out.log 2>&1
cd $R
python3 bench/c3_pmc.py --report $(ls gpurun_out/c3_pmc/*counter_collection.csv | head -1) > gpurun_out/c3_pmc.txt
cat gpurun_out/c3_pmc.txt
rm -f gpurun_out/c3_pmc/*.csv
