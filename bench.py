"""Headline benchmark: ResNet-50, 224x224, bf16, synthetic data, images/sec for the whole job.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python bench.py --gpus 8                      # starts 8 ranks itself (children of this process)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One process per GPU (RANK / LOCAL_RANK / WORLD_SIZE from the environment), weak scaling: the
per-GPU batch is fixed, so the global batch is ``batch x N``.  Every timed step is a complete
training step: forward, loss, backward with the bucketed gradient all-reduce overlapped, and the
fused optimizer step.  Rank 0 prints one JSON line.

The run either finishes or fails loudly: a step that does not finish within ``--step-timeout-s``
(set-up, warm-up and timed steps alike) makes every rank dump its stacks, abort its RCCL
communicator and exit non-zero, and rank 0 prints a JSON line with an ``error`` field instead of a
number.  ``--gpus`` that disagrees with the launcher's ``WORLD_SIZE`` is an error, not a warning.

Role parity: the launch contract and timing of /root/reference/src/main.py:38-42,65-84.
"""

from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC for RCCL between ranks; must precede HIP start-up


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch-size", type=int, default=0,
                    help="per-GPU batch; 0 = the model's default, sized for 288 GB of HBM (ResNet-50 4000: ~150 GiB, "
                         "ViT-B/16 2216: ~130 GiB, GPT-2 medium 128: 107 GiB)")
    ap.add_argument("--model", type=str, default="resnet50")
    ap.add_argument("--image-size", type=int, default=224)
    ap.add_argument("--seq-len", type=int, default=1024)
    ap.add_argument("--optimizer", type=str, default="")
    ap.add_argument("--bucket-mb", type=str, default="auto",
                    help="gradient bucket capacity in MiB, or 'auto': chosen at start-up from the measured all-reduce "
                         "latency / bandwidth of this node and the measured backward time (a no-op on one GPU)")
    ap.add_argument("--last-bucket-mb", type=float, default=1.0)
    ap.add_argument("--grad-dtype", type=str, default="fp32", choices=["fp32", "bf16"],
                    help="element type of gradients on the wire (accumulation inside a rank stays fp32)")
    ap.add_argument("--reduce-algorithm", type=str, default="allreduce", choices=["allreduce", "rs_ag"])
    ap.add_argument("--step-timeout-s", type=float, default=300.0,
                    help="no finished step for this long (set-up, warm-up or timed): stacks are dumped, the communicator "
                         "is aborted, rank 0 prints a JSON line with an error and every rank exits non-zero")
    ap.add_argument("--extras-timeout-s", type=float, default=120.0,
                    help="the untimed diagnostics after the measurement are abandoned after this long")
    ap.add_argument("--cpu", action="store_true", help="plumbing check on CPU/gloo (not a valid benchmark)")
    ap.add_argument("--no-dp", action="store_true", help="single process: skip the one-rank data-parallel wrap")
    ap.add_argument("--no-extras", action="store_true", help="skip the untimed diagnostics after the timed steps")
    return ap.parse_args(argv)


def _free_port() -> int:
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch_ranks(args) -> int:
    """``python bench.py --gpus N`` outside a launcher: start the N ranks as children (``torch.distributed.run``) and
    return their exit code.  Runs before this process has touched the GPU; it never replaces itself."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    print(f"bench.py: --gpus {args.gpus} without a launcher: starting {args.gpus} ranks: {' '.join(cmd)}", file=sys.stderr)
    return subprocess.call(cmd, env=env)


def _claim_stdout():
    """Keep stdout for the one JSON line: the process's file descriptor 1 is pointed at stderr (native libraries -
    RCCL prints a version banner when a communicator is created - write there directly), and the returned file
    object writes to the original stdout."""
    sys.stdout.flush()
    real = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    return real


# Per-GPU batch of each model.  Large, because the persistent kernels deal whole rounds of tiles to 256 CUs, and chosen so
# that the layers with the fewest tiles fill their last round: ResNet-50's 7x7 and 14x14 stages have batch x 49 (x 196)
# output rows, and 256x256 tiles of a 512- (256-) channel output make batch / 668.7 rounds - 4000 is just under six (2048
# was 3.06: a fourth round for 6 % of a round's work); ViT-B/16's 768-wide layers make batch / 110.9 rounds (2216: just
# under twenty).  Measured on one box: ResNet-50 2048 -> 4000 +2-3 %, ViT-B/16 1024 -> 2216 +3.7 % (bench/batch_sweep.sh).
DEFAULT_BATCH = {"resnet50": 4000, "vit_b_16": 2216, "gpt2_medium": 128}

# Context only, NOT measured in this run: stock PyTorch-ROCm 2.10 on one MI355X (bench/stock_baseline.py, MIOpen find
# mode for ResNet-50), each at the largest batch measured for it.
STOCK_PER_GPU = {"resnet50": (6888.2, 1024), "vit_b_16": (4073.8, 1024), "gpt2_medium": (217173.9, 128)}


def metric_name(args, is_lm: bool, on_gpu: bool) -> str:
    return (("tokens/sec (whole job) " if is_lm else "images/sec (whole node) ")
            + f"{args.model} {'seq ' + str(args.seq_len) if is_lm else str(args.image_size) + '^2'} "
            + f"{'bf16' if on_gpu else 'fp32'} training")


def main():
    args = parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be at least 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(launch_ranks(args))
    if args.batch_size <= 0:
        args.batch_size = DEFAULT_BATCH.get(args.model, 256)
    result_out = _claim_stdout()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank_env = int(os.environ.get("RANK", "0"))
    if world != args.gpus:
        # the number asked for and the number of ranks the launcher started differ: whatever would be measured is
        # not what the caller expects to read
        msg = f"--gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks"
        if rank_env == 0:
            print(json.dumps({"metric": None, "value": None, "n_gpus": world, "error": msg}), file=result_out, flush=True)
        print(f"bench.py: {msg}", file=sys.stderr)
        raise SystemExit(2)

    import torch

    from pytorch_distributed_training_amd import ops
    from pytorch_distributed_training_amd.data import synthetic_image_batches, synthetic_token_batches
    from pytorch_distributed_training_amd.models import MODELS, build_model
    from pytorch_distributed_training_amd.optim import build_optimizer
    from pytorch_distributed_training_amd.parallel import (all_gather_values, all_reduce_values, barrier, init_distributed,
                                                           shutdown, wrap_data_parallel)
    from pytorch_distributed_training_amd.parallel.autotune import BucketAutoTuner
    from pytorch_distributed_training_amd.utils.streams import install_stream_census, stream_census
    from pytorch_distributed_training_amd.utils.watchdog import WATCHDOG_EXIT_CODE, StepWatchdog, maybe_inject_fault

    install_stream_census()
    if world == 1 and torch.cuda.is_available() and not args.cpu and not args.no_dp:
        # a plain `python bench.py` on one GPU still runs the data-parallel path (own RCCL communicator, bucket
        # launches from the gradient hooks, event-ordered optimizer step) with a one-rank group, so the 1-GPU
        # number carries the same per-step bookkeeping as the N-GPU numbers it is compared with
        if "RANK" not in os.environ or "MASTER_PORT" not in os.environ:  # not launched by torchrun
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1",
                              LOCAL_RANK="0")
        env = init_distributed(True, use_cpu=False)
    else:
        env = init_distributed(world > 1, use_cpu=args.cpu)
    dev = env.device
    on_gpu = dev.type == "cuda"
    spec = MODELS[args.model]
    is_lm = spec.kind == "lm"
    auto_buckets = str(args.bucket_mb).lower() == "auto"
    bucket_mb = 25.0 if auto_buckets else float(args.bucket_mb)

    # ---- from here on a hang must not be silent
    state = {"phase": "build", "step": 0}
    printed = threading.Lock()

    def emit(rec):
        """Rank 0 prints the one JSON line (exactly once, whoever gets here first)."""
        if not printed.acquire(blocking=False):
            return
        if env.is_main:
            print(json.dumps(rec), file=result_out, flush=True)

    comm = env.comm

    def on_hang():
        # (the watchdog has already written this rank's stacks to stderr)
        if comm is not None:
            try:
                comm.abort()  # kernels blocked inside a collective return; peers see the failure
            except Exception:
                pass
        emit({"metric": metric_name(args, is_lm, on_gpu), "value": None, "n_gpus": env.world_size, "steps": args.steps,
              "warmup": args.warmup, "error": f"rank {env.rank}: {dog.failure} during {state['phase']}"})
        sys.stderr.flush()
        os._exit(WATCHDOG_EXIT_CODE)

    dog = StepWatchdog(args.step_timeout_s, env.rank, on_timeout=on_hang,
                       probe=(comm.async_error if comm is not None else None), probe_every_s=2.0).start()

    torch.manual_seed(1234)
    model = build_model(args.model, 1000, args.image_size).to(dev)
    if on_gpu:
        model = model.to(memory_format=torch.channels_last)
    raw = ops.prepare_module(model)
    if env.distributed:
        model = wrap_data_parallel(model, env, bucket_mb=bucket_mb, grad_dtype=args.grad_dtype,
                                   algorithm=args.reduce_algorithm, last_bucket_mb=args.last_bucket_mb)
    opt_name = args.optimizer or ("adamw" if (is_lm or args.model.startswith("vit")) else "sgd")
    lr = 3e-4 if opt_name != "sgd" else 0.1
    wd = 0.1 if opt_name == "adamw" else 1e-4
    optimizer = build_optimizer([p for p in raw.parameters() if p.requires_grad], opt_name, lr, wd, 0.9)

    # two device-resident synthetic batches, alternated
    if is_lm:
        make = synthetic_token_batches(args.batch_size, args.seq_len, raw.vocab_size, dev, seed=env.rank)
        batches = [make(i) for i in range(2)]
    else:
        make = synthetic_image_batches(args.batch_size, args.image_size, 1000, "cpu", seed=env.rank)
        batches = []
        for i in range(2):
            if on_gpu:
                # generated on the device (a 4000-image fp32 batch is 2.4 GB: eight ranks would draw 40 GB on the host)
                g = torch.Generator(device=dev).manual_seed(2 * env.rank + i)
                x = torch.rand((args.batch_size, 3, args.image_size, args.image_size), device=dev, generator=g)
                y = torch.randint(0, 1000, (args.batch_size,), device=dev, generator=g)
                batches.append((ops.to_device_layout(x, torch.bfloat16), y))
                del x
            else:
                x, y = make(i)
                batches.append((x.to(dev), y.to(dev)))

    def zero_grad():
        if env.distributed:
            model.reducer.zero_grad()
        else:
            optimizer.zero_grad(set_to_none=False)

    def step(i):
        a, b = batches[i % 2]
        if is_lm:
            loss = model(a, b)
        else:
            loss = ops.softmax_cross_entropy(model(a), b)
        loss.backward()
        optimizer.step()
        zero_grad()
        state["step"] += 1
        dog.beat(state["step"])
        maybe_inject_fault(env.rank, state["step"])
        return loss

    def sync():
        if on_gpu:
            torch.cuda.synchronize(dev)
        barrier(env)
        if on_gpu:
            torch.cuda.synchronize(dev)
        dog.beat()

    model.train()
    zero_grad()
    # one-time lazy work happens in the first two steps (bf16 weight copies, optimizer launch plan, kernel
    # attributes, the reducer's bucket rebuild from the observed gradient order): keep it out of the timed
    # region even when the caller asks for no warmup
    SETUP_STEPS = 2
    state["phase"] = "set-up steps"
    loss = step(0)  # the reducer re-plans its buckets from the observed gradient order after this one
    tuner = BucketAutoTuner(model, env, args.grad_dtype) if (auto_buckets and env.distributed) else None
    if tuner is not None:
        tuner.arm()  # timing marks behind every gradient of the next, ordinary step
    loss = step(1)
    tuned = None
    if tuner is not None:
        state["phase"] = "bucket auto-tune"
        sync()
        tuned = tuner.finish()  # None on a single rank
    state["phase"] = "warm-up steps"
    for i in range(args.warmup):
        loss = step(i)
    sync()
    state["phase"] = "timed steps"
    marks = []
    if on_gpu:
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
        marks[0].record()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = step(i)
        if marks:
            marks[i + 1].record()
    sync()
    elapsed_local = time.perf_counter() - t0
    state["phase"] = "collecting results"
    final_loss = float(loss.detach())
    elapsed = all_reduce_values(env, [elapsed_local], "max")[0]

    per_step_samples = args.batch_size * env.world_size
    unit_scale = args.seq_len if is_lm else 1
    value = per_step_samples * unit_scale * args.steps / elapsed
    out = {
        "metric": metric_name(args, is_lm, on_gpu),
        "value": round(value, 2),
        "unit": "tokens/s" if is_lm else "images/s",
        "n_gpus": env.world_size,
        "steps": args.steps,
        "warmup": args.warmup,
        "setup_steps": SETUP_STEPS,
        "ms_per_step": round(1e3 * elapsed / args.steps, 3),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "bf16" if on_gpu else "fp32",
        "data": "synthetic (device-resident random batches, random-init weights)",
        "config": {
            "model": args.model,
            "global_batch": per_step_samples,
            "per_gpu_batch": args.batch_size,
            "seq_len": args.seq_len if is_lm else None,
            "image_size": None if is_lm else args.image_size,
            "optimizer": opt_name,
            "parallelism": f"dp{env.world_size}",
            "bucket_mb": (tuned or {}).get("bucket_mb", bucket_mb) if auto_buckets else bucket_mb,
            "bucket_policy": "auto" if auto_buckets else "fixed",
            "grad_dtype": args.grad_dtype,
            "reduce_algorithm": args.reduce_algorithm,
        },
        "final_loss": round(final_loss, 4),
        "peak_mem_gib": round(torch.cuda.max_memory_allocated(dev) / 2**30, 1) if on_gpu else None,
    }
    # the reference publishes no number (vs_baseline stays null).  For orientation only: the same model on stock
    # PyTorch-ROCm, measured in another session at another batch size - a constant of this file, not a same-run ratio
    stock = STOCK_PER_GPU.get(args.model)
    if stock and on_gpu:
        out["context"] = {"stock_pytorch_rocm_per_gpu": stock[0], "stock_per_gpu_batch": stock[1],
                          "stock_measured": "separately (bench/stock_baseline.py), not in this run",
                          "ratio_per_gpu": round(value / (stock[0] * env.world_size), 3)}

    # Untimed checks and diagnostics (every rank takes part).  They prove that the timed steps were a real N-rank
    # job - RCCL's own rank count, a rank-sum through the reducer's communicator, bit-identical parameters on every
    # rank afterwards - and tell how much of the collective is hidden behind backward, what the links deliver and
    # how the bucket size matters.  The headline number above is not affected, and a diagnostic that hangs cannot
    # take it away: after --extras-timeout-s the line is printed with what has been collected and the process ends.
    extras = {}
    if tuned is not None:
        extras["bucket_autotune"] = tuned

    def give_up():
        extras["error"] = f"diagnostics abandoned after {args.extras_timeout_s:.0f} s"
        emit(dict(out, extras=dict(extras)))
        os._exit(0)

    dog.stop()  # the measurement is over; from here the extras guard applies
    guard = threading.Timer(args.extras_timeout_s, give_up)
    guard.daemon = True
    if not args.no_extras:
        guard.start()
        try:
            # a straggler shows here: wall time of the timed loop on every rank, and each rank's fastest / slowest step
            extras["elapsed_s_per_rank"] = [round(v, 4) for v in all_gather_values(env, elapsed_local)]
            if marks:
                ms = [marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)]
                extras["step_ms_min_per_rank"] = [round(v, 3) for v in all_gather_values(env, min(ms))]
                extras["step_ms_max_per_rank"] = [round(v, 3) for v in all_gather_values(env, max(ms))]
            if on_gpu:
                extras["streams"] = stream_census()
            if env.distributed:
                collect_extras(extras, args, env, model, raw, step, sync, zero_grad, elapsed, on_gpu, dev, bucket_mb)
        except Exception as e:  # diagnostics must never cost the measurement
            extras["error"] = repr(e)
    guard.cancel()
    emit(dict(out, extras=extras) if extras else out)
    if env.distributed:
        model.reducer.close()
    shutdown(env)


def collect_extras(extras, args, env, model, raw, step, sync, zero_grad, elapsed, on_gpu, dev, bucket_mb):
    import torch
    import torch.distributed as dist

    reducer = model.reducer
    extras["buckets"] = reducer.bucket_summary()
    extras["transport"] = type(model.transport).__name__
    extras["control_backend"] = env.control_backend
    extras["wire_mib_per_step"] = round(reducer.wire_bytes_per_step / 2**20, 1)
    comm = getattr(model.transport, "comm", None)
    # --- proofs of an N-rank job
    if comm is not None:
        extras["rccl_ranks"] = comm.rccl_ranks  # ncclCommCount of the process's one communicator
        t = torch.full((4,), float(env.rank + 1), device=dev)
        comm.ordered_all_reduce_(t, "sum")
        got = float(t[0])
        extras["rank_sum"] = {"got": got, "expected": env.world_size * (env.world_size + 1) / 2}
        sums = torch.stack([p.detach().sum() for p in raw.parameters()]
                           + [p.detach().abs().sum() for p in raw.parameters()]).float()
        spread = torch.cat([sums, -sums])
        comm.ordered_all_reduce_(spread, "max")  # max(v) + max(-v) = max(v) - min(v) over ranks
        n = sums.numel()
        delta = float((spread[:n] + spread[n:]).abs().max())
        extras["params_max_abs_delta_across_ranks"] = delta
        extras["params_in_sync"] = bool(delta == 0.0 and got == extras["rank_sum"]["expected"]
                                        and extras["rccl_ranks"] == env.world_size)
    else:
        flat = torch.cat([p.detach().reshape(-1) for p in raw.parameters()])
        ref = flat.clone()
        dist.broadcast(ref, src=0)
        same = torch.tensor([1.0 if torch.equal(ref, flat) else 0.0])
        dist.all_reduce(same, op=dist.ReduceOp.MIN)
        extras["params_in_sync"] = bool(float(same) == 1.0)
    # --- the same step with the gradient collectives switched off
    n_local = max(2, min(5, args.steps))

    def time_steps(n):
        step(0)
        sync()
        t1 = time.perf_counter()
        for i in range(n):
            step(i)
        sync()
        return 1e3 * (time.perf_counter() - t1) / n

    with model.no_sync():
        local_ms = time_steps(n_local)
    zero_grad()
    extras["ms_per_step_without_allreduce"] = round(local_ms, 3)
    extras["allreduce_exposed_ms"] = round(1e3 * elapsed / args.steps - local_ms, 3)
    if comm is not None and env.world_size > 1:
        # --- what the links deliver: all-reduce rate of the communicator over message sizes
        for name, dtype, per_mib in (("fp32", torch.float32, 262144), ("bf16", torch.bfloat16, 524288)):
            sweep = {}
            for mib in (1, 4, 16, 64, 256):
                buf = torch.zeros(mib * per_mib, dtype=dtype, device=dev)
                us = comm.time_all_reduce(buf, warmup=3, iters=10)
                alg = mib * 1048576 / us / 1e3  # GB/s
                sweep[f"{mib}MiB"] = {"us": round(us, 1), "algbw_GBps": round(alg, 1),
                                      "busbw_GBps": round(alg * 2 * (env.world_size - 1) / env.world_size, 1)}
                del buf
            extras[f"allreduce_sweep_{name}"] = sweep
        # --- how the bucket capacity matters for this model on this node (3 timed steps each, same step function)
        keep = (reducer.bucket_bytes / 2**20, reducer.first_bucket_bytes / 2**20, reducer.last_bucket_bytes / 2**20)
        tuning = {}
        for mb in (8, 16, 25, 50, 100):
            reducer.rebucket(mb)
            zero_grad()
            tuning[f"{mb}MiB"] = {"ms_per_step": round(time_steps(3), 3), "buckets": len(reducer.buckets)}
        extras["bucket_mb_sweep"] = tuning
        reducer.rebucket(*keep)
        zero_grad()


if __name__ == "__main__":
    main()
