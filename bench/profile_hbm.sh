#!/bin/bash
# HBM traffic and L2 hit rate per kernel of bench.py on one GPU:  bench/profile_hbm.sh NAME [bench.py arguments]
# Two counter passes (a run holds 4 TCC counters), each a run of its own with the kernel trace only.  Reads are counted
# by request size from the raw counters: rocprofv3's derived FETCH_SIZE prices every request that is not 32 bytes at 64
# bytes, but on gfx950 practically all of them are 128-byte requests (TCC_EA0_RDREQ_128B), so it reports half the bytes.  Writes gpurun_out/hbm_NAME/ and a per-kernel summary (bytes per step, achieved TB/s).
# 1 timed + 2 setup steps = 3 steps per pass.
set -e
name=$1; shift
R=$PWD
D=$R/gpurun_out/hbm_$name
mkdir -p $D
export TMPDIR=/tmp
cd /tmp
timeout -s KILL 400 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --kernel-trace --output-format csv -d $D -o rd -- python3 $R/bench.py --steps 1 --warmup 0 --no-extras "$@" > $D/rd.log 2>&1
timeout -s KILL 400 rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $D -o wr -- python3 $R/bench.py --steps 1 --warmup 0 --no-extras "$@" > $D/wr.log 2>&1
cd $R
python3 bench/hbm_summary.py 3 $(ls $D/*rd_counter_collection.csv | head -1) $(ls $D/*wr_counter_collection.csv | head -1) $(ls $D/*rd_kernel_trace.csv | head -1) > $D/summary.txt
head -40 $D/summary.txt | cut -c1-200
