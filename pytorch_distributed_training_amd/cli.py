"""Command line.

The first seven options are the reference's, with identical names, types, defaults and
``show_default`` settings (/root/reference/src/main.py:18-25); running with no other option
reproduces the reference workload (ResNet-18, CIFAR-10-shaped data, Adam with L2 decay,
cross-entropy, one pass).  Everything after them is additive.

One deliberate divergence: ``--distributed --use-cpu`` is accepted and runs over gloo
(the reference asserts against it at main.py:36-37).
"""

from __future__ import annotations

import click

from .models import MODELS
from .train import TrainConfig, run_training


@click.command()
# ---- reference flags (main.py:19-25)
@click.option("--data-dir", type=str, default="~/Downloads/data", show_default=True)
@click.option("--distributed", is_flag=True, default=False)
@click.option("--use-cpu", is_flag=True, default=False)
@click.option("--batch-size", type=int, default=32, show_default=True)
@click.option("--num-workers", type=int, default=2, show_default=True)
@click.option("--learning-rate", type=float, default=0.1, show_default=True)
@click.option("--weight-decay", type=float, default=0.001, show_default=True)
# ---- additive flags
@click.option("--model", type=click.Choice(sorted(MODELS)), default="resnet18", show_default=True)
@click.option("--dataset", type=click.Choice(["auto", "cifar10", "synthetic"]), default="auto", show_default=True,
              help="auto: CIFAR-10 from --data-dir when present, else synthetic data of the same shape")
@click.option("--num-classes", type=int, default=0, help="0: taken from the dataset / model default")
@click.option("--image-size", type=int, default=0, help="0: model default (32 for resnet18, 224 otherwise)")
@click.option("--seq-len", type=int, default=1024, show_default=True)
@click.option("--epochs", type=int, default=1, show_default=True)
@click.option("--steps", type=int, default=0,
              help="stop once this many optimisation steps have been taken in total, resumed steps included (0: full epochs)")
@click.option("--warmup-steps", type=int, default=0, help="steps excluded from the throughput measurement")
@click.option("--dtype", type=click.Choice(["auto", "fp32", "bf16"]), default="auto", show_default=True)
@click.option("--optimizer", type=click.Choice(["adam", "adamw", "sgd"]), default="adam", show_default=True)
@click.option("--momentum", type=float, default=0.9, show_default=True)
@click.option("--grad-accum", type=int, default=1, show_default=True)
@click.option("--shuffle/--no-shuffle", default=False)
@click.option("--drop-last", is_flag=True, default=False)
@click.option("--train-split", is_flag=True, default=False, help="use the CIFAR-10 training split (reference uses the test split)")
@click.option("--eval", "evaluate", is_flag=True, default=False, help="run the evaluation loop after training")
@click.option("--checkpoint-path", type=str, default="", help="rank 0 writes model.state_dict() here")
@click.option("--resume", type=str, default="",
              help="checkpoint to restore before training; 'auto': the newest intact checkpoint at --checkpoint-path, "
                   "if any (restartable job lines)")
@click.option("--save-every", type=int, default=0, help="also checkpoint every N steps")
@click.option("--keep-checkpoints", type=int, default=2, show_default=True,
              help="checkpoint generations (<path>.step<N>) kept next to --checkpoint-path; when the newest one is torn "
                   "(the job died while writing it) a resume falls back to the one before; 0: overwrite in place")
@click.option("--bucket-mb", type=str, default="auto", show_default=True,
              help="gradient bucket capacity in MiB, or 'auto': chosen after the first step from the all-reduce latency and "
                   "bandwidth measured on this node and the measured gradient-ready times (nothing to tune on one rank)")
@click.option("--first-bucket-mb", type=float, default=1.0, show_default=True)
@click.option("--sync-buffers/--no-sync-buffers", default=True, help="broadcast BatchNorm buffers from rank 0 every step")
@click.option("--last-bucket-mb", type=float, default=1.0, show_default=True,
              help="capacity of the final bucket (the collective nothing overlaps); 0: no separate final bucket")
@click.option("--grad-dtype", type=click.Choice(["fp32", "bf16"]), default="fp32", show_default=True,
              help="element type of gradients on the wire (accumulation inside a rank stays fp32)")
@click.option("--reduce-algorithm", type=click.Choice(["allreduce", "rs_ag"]), default="allreduce", show_default=True,
              help="per-bucket collective: all-reduce, or reduce-scatter + all-gather (GPU ranks)")
@click.option("--device-data", is_flag=True, default=False, help="keep synthetic batches resident on the device")
@click.option("--dataset-length", type=int, default=0)
@click.option("--seed", type=int, default=0, show_default=True)
@click.option("--log-every", type=int, default=0, help="record/print the loss every N steps")
@click.option("--progress/--no-progress", default=True)
@click.option("--watchdog-s", type=float, default=0.0, help="abort when no step finishes for this many seconds")
@click.option("--dist-timeout-s", type=float, default=600.0, show_default=True)
@click.option("--comm-poll-s", type=float, default=2.0, show_default=True,
              help="seconds between queries of RCCL's asynchronous error state (0: never); an error aborts the communicator "
                   "and ends the process so a supervisor can restart the job")
@click.option("--stop-check-every", type=int, default=0,
              help="steps between the ranks' agreements on stop requests (0: every step on CPU, every 16 on GPU)")
@click.option("--profile", type=str, default="", help="write a chrome trace of host-side phases to this path")
@click.option("--check-finite", is_flag=True, default=False,
              help="stop with an error on a non-finite loss (checked on --log-every steps; distributed GPU jobs act on it "
                   "at the next agreement step, at most --stop-check-every steps later, and once more when the loop ends)")
@click.option("--deterministic", is_flag=True, default=False,
              help="bit-identical GPU results from run to run (weight gradients, BatchNorm sums and embedding gradients "
                   "avoid order-dependent floating-point atomics); slower, for debugging divergence")
@click.option("--hip-graph/--no-hip-graph", default=None,
              help="record the whole training step into a hipGraph after --graph-after eager steps (launch-bound configs); "
                   "default: automatic - on for single-GPU image runs of at most 64 x 224 x 224 input pixels per batch")
@click.option("--graph-after", type=int, default=3, show_default=True)
def main(**kwargs):
    cfg = TrainConfig(**kwargs)
    run_training(cfg)


if __name__ == "__main__":
    main()
