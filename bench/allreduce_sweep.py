"""All-reduce bandwidth vs. message size through the framework's own RCCL communicator.

Run on a multi-GPU node to pick `--bucket-mb`: xGMI is point-to-point (7 links per GPU), so a bucket must
be large enough that each per-peer shard runs a link at bandwidth rather than latency, yet small enough
that several buckets are in flight during backward.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        --master-port 29500 bench/allreduce_sweep.py [--dtype fp32|bf16] [--max-mb 512]

Prints, per size: time per all-reduce, algorithm bandwidth (bytes / time) and bus bandwidth
(algbw x 2(N-1)/N, the figure comparable across N).  Works with a single rank too (N = 1).
"""

from __future__ import annotations

import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", choices=["fp32", "bf16"], default="fp32")
    ap.add_argument("--min-kb", type=int, default=64)
    ap.add_argument("--max-mb", type=int, default=256)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--json", type=str, default="")
    args = ap.parse_args()
    from pytorch_distributed_training_amd.parallel import RcclComm, init_distributed, shutdown

    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    os.environ.setdefault("LOCAL_RANK", "0")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29512")
    env = init_distributed(True)
    comm = RcclComm(env.rank, env.world_size, env.device)
    dtype = torch.float32 if args.dtype == "fp32" else torch.bfloat16
    esize = 4 if args.dtype == "fp32" else 2
    n = env.world_size
    rows = []
    size = args.min_kb * 1024
    if env.is_main:
        print(f"world={n} dtype={args.dtype}")
        print(f"{'bytes':>12}{'us':>10}{'algbw GB/s':>12}{'busbw GB/s':>12}")
    while size <= args.max_mb * 1024 * 1024:
        t = torch.ones(size // esize, dtype=dtype, device=env.device)
        us = comm.time_all_reduce(t, warmup=3, iters=args.iters)
        alg = size / (us * 1e-6) / 1e9
        bus = alg * (2.0 * (n - 1) / n) if n > 1 else alg
        if env.is_main:
            print(f"{size:12d}{us:10.1f}{alg:12.1f}{bus:12.1f}")
        rows.append(dict(bytes=size, us=us, algbw=alg, busbw=bus))
        size *= 2
        del t
    if args.json and env.is_main:
        with open(args.json, "w") as f:
            json.dump(dict(world=n, dtype=args.dtype, rows=rows), f, indent=1)
    comm.close()
    shutdown(env)


if __name__ == "__main__":
    main()
