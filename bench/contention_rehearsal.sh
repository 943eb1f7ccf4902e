#!/bin/bash
# One-GPU rehearsal of what resident collective kernels cost the persistent GEMM grids: the bucket collectives run at world
# size 1 (PDT_SELF_ALLREDUCE=1) and each is followed by C resident workgroups that stay for the time the bucket would
# take at a given link rate.  Usage: bench/contention_rehearsal.sh [bench.py arguments]
out=gpurun_out/contention
mkdir -p $out
run() { name=$1; shift; env "$@" timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-extras $ARGS 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(f'$name: {d[\"ms_per_step\"]:8.3f} ms/step  {d[\"value\"]:10.1f}')" | tee -a $out/table.txt; }
ARGS="$@"
: > $out/table.txt
run "baseline (one-rank collective skipped)      " PDT_X=0
run "self all-reduce, no resident workgroups    " PDT_SELF_ALLREDUCE=1
for gbps in 150 50; do for c in 16 32 64; do
run "resident $c workgroups at $gbps GB/s          " PDT_SELF_ALLREDUCE=1 PDT_REHEARSE_CHANNELS=$c PDT_REHEARSE_GBPS=$gbps
done; done
# the footprint of a real collective's workgroup: 512 threads x 128 registers + 48 KB of LDS (cannot share a CU with a
# 256-register ring workgroup)
for gbps in 150 50; do for c in 16 32 64; do
run "heavy resident $c workgroups at $gbps GB/s    " PDT_SELF_ALLREDUCE=1 PDT_REHEARSE_HEAVY=1 PDT_REHEARSE_CHANNELS=$c PDT_REHEARSE_GBPS=$gbps
done; done
run "resident 32 at 50 GB/s, 32 CUs reserved      " PDT_SELF_ALLREDUCE=1 PDT_REHEARSE_CHANNELS=32 PDT_REHEARSE_GBPS=50 PDT_RESERVE_CUS=32
run "no collective, 32 CUs reserved               " PDT_RESERVE_CUS=32
# finer hand-out by the hardware's dispatcher: two workgroups per CU slot, each with half the tiles
run "two rounds of workgroups, no collective       " PDT_GRID_ROUNDS=2 PDT_WGRAD_RING_TARGET_BLOCKS=512
for gbps in 150 50; do for c in 16 32 64; do
run "two rounds, heavy resident $c at $gbps GB/s      " PDT_GRID_ROUNDS=2 PDT_WGRAD_RING_TARGET_BLOCKS=512 PDT_SELF_ALLREDUCE=1 PDT_REHEARSE_HEAVY=1 PDT_REHEARSE_CHANNELS=$c PDT_REHEARSE_GBPS=$gbps
done; done
run "four rounds of workgroups, no collective      " PDT_GRID_ROUNDS=4 PDT_WGRAD_RING_TARGET_BLOCKS=1024
run "four rounds, heavy resident 32 at 50 GB/s      " PDT_GRID_ROUNDS=4 PDT_WGRAD_RING_TARGET_BLOCKS=1024 PDT_SELF_ALLREDUCE=1 PDT_REHEARSE_HEAVY=1 PDT_REHEARSE_CHANNELS=32 PDT_REHEARSE_GBPS=50
