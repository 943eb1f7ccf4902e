#!/bin/bash
# MFMA / vector-ALU counters of the convolution kernels of some ResNet-50 layers: bench/layer_pmc.sh LAYER-FILTER [batch]
set -e
R=$PWD
D=$R/gpurun_out/layer_pmc
mkdir -p $D
export TMPDIR=/tmp
cd /tmp
timeout -s KILL 200 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 --kernel-trace --output-format csv -d $D -o pmc -- python3 $R/bench/conv_bench.py --batch ${2:-4000} --iters 1 --only $1 > $D/out.log 2>&1
cd $R
python3 bench/pmc_summary.py $(ls $D/*counter_collection.csv | head -1) > $D/summary.csv
rm -f $D/*counter_collection.csv $D/*kernel_trace.csv
cut -d, -f1,2,10,11,12 $D/summary.csv | head -12
