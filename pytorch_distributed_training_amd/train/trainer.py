"""Training / evaluation loop.

Covers the reference's loop (/root/reference/src/main.py:61-84: DataLoader, CrossEntropy,
Adam, one pass of ``zero_grad -> forward -> loss -> backward -> step`` under a tqdm bar,
elapsed-time report) and the capabilities the reference lacks: sharded sampling, epochs,
evaluation, throughput metering, rank-0 checkpoint/resume, gradient accumulation, failure
detection.  The five printed contract lines are preserved verbatim.
"""

from __future__ import annotations

import math
import os
import sys
import time
from dataclasses import dataclass, field
from typing import Any, Dict, Optional

import torch
import torch.nn as nn
from tqdm import tqdm

from .. import ops
from ..data import (
    DevicePrefetcher,
    DeviceSyntheticLoader,
    build_dataset,
    build_loader,
    synthetic_image_batches,
    synthetic_token_batches,
)
from ..models import MODELS, build_model
from ..optim import build_optimizer
from ..parallel.autotune import BucketAutoTuner
from ..parallel import DistEnv, all_reduce_values, barrier, init_distributed, shutdown, wrap_data_parallel
from ..utils.logging import get_logger, rank_zero_print
from ..utils.meter import AverageMeter, ThroughputMeter
from ..utils.streams import check_stream_budget, install_stream_census
from ..utils.timers import ChromeTrace, StepTimer
from ..utils.watchdog import GracefulStop, StepWatchdog, maybe_inject_fault
from . import checkpoint as ckpt

log = get_logger("pdt.train")


@dataclass
class TrainConfig:
    # --- the reference's seven flags (main.py:19-25), same defaults
    data_dir: str = "~/Downloads/data"
    distributed: bool = False
    use_cpu: bool = False
    batch_size: int = 32  # per process
    num_workers: int = 2
    learning_rate: float = 0.1
    weight_decay: float = 0.001
    # --- additive options
    model: str = "resnet18"
    dataset: str = "auto"  # auto | cifar10 | synthetic
    num_classes: int = 0  # 0 => dataset / model default
    image_size: int = 0
    seq_len: int = 1024
    epochs: int = 1
    steps: int = 0  # >0: stop once this many optimisation steps have been taken in total (resumed steps count)
    warmup_steps: int = 0  # steps excluded from the throughput window
    dtype: str = "auto"  # auto: bf16 on GPU, fp32 on CPU
    optimizer: str = "adam"
    momentum: float = 0.9
    grad_accum: int = 1
    shuffle: bool = False
    drop_last: bool = False
    train_split: bool = False  # reference trains on the CIFAR-10 *test* split (main.py:47)
    evaluate: bool = False
    checkpoint_path: str = ""
    resume: str = ""  # checkpoint to restore; "auto": --checkpoint-path when it exists (restartable job lines)
    save_every: int = 0  # steps; 0 => only at the end
    keep_checkpoints: int = 2  # checkpoint generations kept on disk; a torn newest one falls back to the one before
    # gradient bucket capacity in MiB, or "auto": chosen after the first step from the measured all-reduce cost of
    # this node and the measured gradient-ready times of this model (single rank: nothing to tune, 25 MiB)
    bucket_mb: Any = "auto"
    first_bucket_mb: float = 1.0
    last_bucket_mb: float = 1.0
    grad_dtype: str = "fp32"  # element type of gradients on the wire: fp32 | bf16
    reduce_algorithm: str = "allreduce"  # allreduce | rs_ag
    sync_buffers: bool = True
    device_data: bool = False  # keep synthetic batches resident on the device
    dataset_length: int = 0
    seed: int = 0
    log_every: int = 0
    progress: bool = True
    watchdog_s: float = 0.0
    comm_poll_s: float = 2.0  # how often the watchdog thread asks RCCL for asynchronous errors (0: never)
    stop_check_every: int = 0  # steps between the ranks' agreements on stop requests; 0: every step on CPU, 16 on GPU
    dist_timeout_s: float = 600.0
    profile: str = ""  # path of a chrome trace with per-phase host spans
    check_finite: bool = False
    # run-to-run bit-identical GPU arithmetic (no order-dependent floating-point atomics; slower): for telling a real
    # divergence between runs or replicas from noise
    deterministic: bool = False
    # record the whole step into a hipGraph after a few eager steps (GPU only).  None = automatic: on for single-GPU
    # image runs small enough to be bound by kernel launches (the reference's own configuration, ResNet-18 on 32x32
    # images with a batch of 32, runs 1.7x faster recorded), off otherwise
    hip_graph: Optional[bool] = None
    graph_after: int = 3  # eager steps before the capture


@dataclass
class TrainResult:
    steps: int = 0
    epochs: int = 0
    elapsed_s: float = 0.0
    samples_per_sec: float = 0.0
    ms_per_step: float = 0.0
    final_loss: float = float("nan")
    eval_accuracy: Optional[float] = None
    eval_loss: Optional[float] = None
    losses: list = field(default_factory=list)
    interrupted: bool = False


def resolve_dtype(cfg: TrainConfig, device: torch.device) -> torch.dtype:
    if cfg.dtype == "auto":
        return torch.bfloat16 if device.type == "cuda" else torch.float32
    return {"fp32": torch.float32, "bf16": torch.bfloat16}[cfg.dtype]


def prepare_image_batch(x: torch.Tensor, device: torch.device, dtype: torch.dtype) -> torch.Tensor:
    """Device placement, compute dtype and (on GPU) NHWC layout for an image batch: raw uint8 ``[N, H, W, 3]``
    (scaled to [0, 1], the effect of the reference's ``ToTensor``, main.py:44-47) or NCHW float."""
    x = x.to(device=device, non_blocking=True)
    if x.dtype == torch.uint8:
        return ops.image_batch_to_device_layout(x, dtype)
    if device.type == "cuda":
        return ops.to_device_layout(x, dtype)
    return x.to(dtype)


class Trainer:
    def __init__(self, cfg: TrainConfig, env: Optional[DistEnv] = None):
        self.cfg = cfg
        self._owns_env = env is None
        install_stream_census()
        self.env = env if env is not None else init_distributed(cfg.distributed, cfg.use_cpu, cfg.dist_timeout_s)
        if self.env.distributed:
            # contract line 1 (main.py:42)
            print(f"Process group initialized - WORLD_SIZE: {self.env.world_size}, RANK: {self.env.rank}", flush=True)
        torch.manual_seed(cfg.seed)
        self.device = self.env.device
        if cfg.deterministic:
            ops.set_deterministic(True)
        self.dtype = resolve_dtype(cfg, self.device)
        spec = MODELS[cfg.model]
        self.kind = spec.kind
        self.image_size = cfg.image_size or spec.default_image_size
        self.num_classes = cfg.num_classes or spec.default_num_classes

        lm_model = build_model(cfg.model) if self.kind == "lm" else None
        vocab = lm_model.vocab_size if lm_model is not None else 0
        if lm_model is not None and cfg.seq_len > lm_model.n_ctx:
            raise ValueError(f"--seq-len {cfg.seq_len} exceeds the model context {lm_model.n_ctx}")
        self.dataset, self.data_desc = build_dataset(
            cfg.dataset, cfg.data_dir, self.kind, self.image_size, self.num_classes,
            cfg.dataset_length or None, cfg.seq_len, vocab or 50257, train=cfg.train_split, seed=cfg.seed,
            wait_for_main=(self.env.is_main, lambda: barrier(self.env)) if self.env.distributed else None,
        )
        if self.kind == "image" and hasattr(self.dataset, "classes") and not cfg.num_classes:
            self.num_classes = len(self.dataset.classes)  # main.py:49

        model = lm_model if lm_model is not None else build_model(cfg.model, self.num_classes, self.image_size)
        model = model.to(self.device)
        if self.device.type == "cuda":
            model = model.to(memory_format=torch.channels_last)
        self.raw_model = ops.prepare_module(model)
        self._use_graph = self._decide_graph()
        self._auto_buckets = False
        if self.env.distributed:
            self._auto_buckets = str(cfg.bucket_mb).lower() == "auto"
            model = wrap_data_parallel(
                model, self.env, 25.0 if self._auto_buckets else float(cfg.bucket_mb), cfg.first_bucket_mb,
                cfg.sync_buffers and not self._use_graph,  # the per-step buffer broadcast is not part of a recorded step
                cfg.grad_dtype, cfg.reduce_algorithm, cfg.last_bucket_mb,
            )
        self.model = model
        self.bucket_decision = None  # what --bucket-mb auto chose (None: fixed capacity, or a single rank)
        # contract line 2 (main.py:59)
        print(f"Device - {self.env.device_label}", flush=True)

        self.optimizer = build_optimizer(
            [p for p in self.raw_model.parameters() if p.requires_grad],
            cfg.optimizer, cfg.learning_rate, cfg.weight_decay, cfg.momentum,
        )
        self.start_epoch = 0
        self.global_step = 0
        self.start_batch = 0  # micro-batches of ``start_epoch`` already consumed before the checkpoint was written
        self._saved_batches_per_epoch = 0  # length of an epoch in the run that wrote the checkpoint (0: unknown)
        resume = asked = cfg.resume
        if resume == "auto":
            asked = cfg.checkpoint_path
            # the newest checkpoint that is whole: the main pair, or - when the job died while replacing it - the
            # generation before it
            resume = (ckpt.find_resumable(cfg.checkpoint_path) or "") if cfg.checkpoint_path else ""
        if resume:
            info = ckpt.load_checkpoint(resume, self.model, self.optimizer)
            self.start_epoch, self.global_step = info["epoch"], info["step"]
            self.start_batch = int(info.get("extra", {}).get("batch_in_epoch", 0))
            self._saved_batches_per_epoch = int(info.get("extra", {}).get("batches_per_epoch", 0))
            # the saved generator state is rank 0's; giving it to every rank would make their random streams equal
            if info.get("torch_rng") is not None and self.env.rank == 0:
                torch.set_rng_state(info["torch_rng"])
            elif self.env.rank != 0:
                torch.manual_seed(cfg.seed + 1_000_003 * self.env.rank + self.global_step)
            if info["source"] != asked:
                rank_zero_print(f"Checkpoint {asked} is torn or missing; using the intact generation {info['source']}")
            rank_zero_print(f"Resumed from {info['source']} at epoch {self.start_epoch}, step {self.global_step}"
                            + (f", batch {self.start_batch} of the epoch" if self.start_batch else ""))
        self.timer = StepTimer(self.device, enabled=bool(cfg.profile))
        self.trace = ChromeTrace(self.env.rank) if cfg.profile else None
        self.stopper = GracefulStop()
        self._graph = None
        self._eager_steps = 0
        if self._use_graph:
            self.optimizer.set_capturable(True)

    # steps with at most this many input pixels per batch are bound by kernel launches, not by the GPU (64 images of 224x224)
    AUTO_GRAPH_MAX_PIXELS = 64 * 224 * 224

    def _decide_graph(self) -> bool:
        cfg = self.cfg
        on_gpu = self.device.type == "cuda"
        if cfg.hip_graph is None:
            return (on_gpu and self.env.world_size == 1 and self.kind == "image" and cfg.grad_accum == 1 and not cfg.profile
                    and cfg.batch_size * self.image_size * self.image_size <= self.AUTO_GRAPH_MAX_PIXELS)
        if cfg.hip_graph and not on_gpu:
            rank_zero_print("--hip-graph ignored: not running on a GPU")
            return False
        if cfg.hip_graph and (cfg.grad_accum > 1 or cfg.profile):
            raise ValueError("--hip-graph records whole steps: it cannot be combined with --grad-accum > 1 or --profile")
        return bool(cfg.hip_graph)

    # ------------------------------------------------------------------ data
    def _make_loader(self):
        cfg = self.cfg
        if cfg.device_data:
            steps = cfg.steps or math.ceil(len(self.dataset) / (cfg.batch_size * self.env.world_size))
            if self.kind == "lm":
                make = synthetic_token_batches(cfg.batch_size, cfg.seq_len, self.raw_model.vocab_size, self.device,
                                               cfg.seed + self.env.rank)
            else:
                base = synthetic_image_batches(cfg.batch_size, self.image_size, self.num_classes, "cpu",
                                               seed=cfg.seed + self.env.rank)

                def make(i):
                    x, y = base(i)
                    return prepare_image_batch(x, self.device, self.dtype), y.to(self.device)

            return DeviceSyntheticLoader(make, steps), None
        loader, sampler = build_loader(
            self.dataset, cfg.batch_size, cfg.num_workers, self.env.rank, self.env.world_size,
            shuffle=cfg.shuffle, drop_last=cfg.drop_last, pin_memory=self.device.type == "cuda", seed=cfg.seed,
        )
        if self.device.type == "cuda":
            loader = DevicePrefetcher(loader, self.device, convert=lambda u8: ops.image_batch_to_device_layout(u8, self.dtype))
        return loader, sampler

    # ------------------------------------------------------------------ one step
    def _forward_loss(self, batch) -> torch.Tensor:
        if self.kind == "lm":
            ids, targets = batch
            return self.model(ids.to(self.device), targets.to(self.device))
        imgs, lbls = batch
        if imgs.device != self.device or imgs.dtype != self.dtype:
            imgs = prepare_image_batch(imgs, self.device, self.dtype)
        preds = self.model(imgs)
        return ops.softmax_cross_entropy(preds, lbls.to(self.device))

    def _zero_grad(self) -> None:
        if hasattr(self.model, "reducer"):
            self.model.reducer.zero_grad()
        else:
            self.optimizer.zero_grad(set_to_none=False)

    def _device_batch(self, batch):
        """The batch as device tensors in the form the model consumes (inputs of a recorded step)."""
        if self.kind == "lm":
            ids, targets = batch
            return ids.to(self.device), targets.to(self.device)
        imgs, lbls = batch
        if imgs.device != self.device or imgs.dtype != self.dtype:
            imgs = prepare_image_batch(imgs, self.device, self.dtype)
        return imgs, lbls.to(self.device)

    def _graph_body(self, a, b) -> torch.Tensor:
        loss = self._forward_loss((a, b))
        loss.backward()
        self.optimizer.step()
        self._zero_grad()
        return loss.detach()

    def _graphed_train_step(self, batch) -> torch.Tensor:
        """Eager for the first ``graph_after`` steps, then one hipGraph replay per batch of the recorded shape."""
        from .graph import GraphedStep

        inputs = self._device_batch(batch)
        if self._graph is None:
            self._eager_steps += 1
            if self._eager_steps <= self.cfg.graph_after:
                return self._graph_body(*inputs)
            self._graph = GraphedStep(self._graph_body, self.optimizer, inputs)
            rank_zero_print(f"hipGraph: recorded the training step after {self.cfg.graph_after} eager steps")
        if self._graph.matches(inputs):
            return self._graph(*inputs)
        return self._graph_body(*inputs)

    def train_step(self, batch, accumulate: bool = False, group_size: Optional[int] = None) -> torch.Tensor:
        """forward + backward (+ optimizer step unless ``accumulate``); returns the detached loss.
        ``group_size``: number of micro-batches whose gradients are summed into this optimizer step (default
        ``--grad-accum``; smaller for the incomplete group at the end of an epoch): the loss is divided by it."""
        cfg = self.cfg
        group_size = group_size or cfg.grad_accum
        if self._use_graph:
            return self._graphed_train_step(batch)
        span = self.trace.span if self.trace else None
        sync_ctx = self.model.no_sync() if (accumulate and hasattr(self.model, "no_sync")) else _null()
        with sync_ctx:
            with self.timer.phase("forward"), (span("forward") if span else _null()):
                loss = self._forward_loss(batch)
                if group_size > 1:
                    loss = loss / group_size
            with self.timer.phase("backward"), (span("backward") if span else _null()):
                loss.backward()
        if not accumulate:
            with self.timer.phase("optimizer"), (span("optimizer") if span else _null()):
                self.optimizer.step()
                self._zero_grad()
        return loss.detach()

    # ------------------------------------------------------------------ loops
    # ------------------------------------------------------------------ agreement between ranks
    def _agree(self, flags: int) -> int:
        """Bitwise OR over ranks of a small set of flags, so every rank takes the same decision at the same step.

        A stop request (SIGTERM reaches ranks at different moments, or only one of them) or a non-finite loss acted
        on by one rank alone leaves the others blocked in the next bucket collective."""
        env = self.env
        if not env.distributed:
            return flags
        t = torch.tensor([float(flags & 1), float(flags >> 1 & 1)], dtype=torch.float32, device=self.device)
        comm = getattr(getattr(self.model, "transport", None), "comm", None)
        if comm is not None:
            comm.ordered_all_reduce_(t, "max")
        else:
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        a, b = t.tolist()
        return int(a > 0) | (int(b > 0) << 1)

    def _comm_probe(self):
        comm = getattr(getattr(self.model, "transport", None), "comm", None)
        return comm.async_error if (comm is not None and self.cfg.comm_poll_s > 0) else None

    def _abort_comm(self) -> None:
        comm = getattr(getattr(self.model, "transport", None), "comm", None)
        if comm is not None:
            comm.abort()

    # ------------------------------------------------------------------ loops
    def fit(self) -> TrainResult:
        cfg, env = self.cfg, self.env
        res = TrainResult()
        loader, sampler = self._make_loader()
        sync = (lambda: torch.cuda.synchronize(self.device)) if self.device.type == "cuda" else None
        meter = ThroughputMeter(env.world_size, cfg.warmup_steps, sync)
        loss_meter = AverageMeter()
        self.model.train()
        self._zero_grad()
        self.stopper.install()
        if self.device.type == "cuda":
            self.stream_census = check_stream_budget(f"rank {env.rank}")
        dog = StepWatchdog(cfg.watchdog_s, env.rank, probe=self._comm_probe(), probe_every_s=cfg.comm_poll_s,
                           before_exit=self._abort_comm).start()
        check_every = cfg.stop_check_every or (1 if self.device.type == "cpu" else 16)
        ga = max(1, cfg.grad_accum)
        tuner = BucketAutoTuner(self.model, env, cfg.grad_dtype) if self._auto_buckets else None
        if tuner is not None and not tuner.active:
            tuner = None
        tuner_armed = False

        if os.environ.get("PDT_TEST_PIDFILE_RANK1") and env.rank == 1:  # lets a test signal exactly this rank
            with open(os.environ["PDT_TEST_PIDFILE_RANK1"], "w") as f:
                f.write(str(os.getpid()))
        t0 = time.perf_counter()
        print("Training starting...", flush=True)  # contract line 3 (main.py:66)
        meter.start()
        done = False
        stopped = False
        last_loss = None
        steps_this_run = 0
        pending = 0  # bit 0: stop requested here, bit 1: non-finite loss seen here; not yet agreed with the other ranks
        epoch = self.start_epoch
        batch_in_epoch = 0
        try:
            if cfg.steps and self.global_step >= cfg.steps:
                done = True
            if self.start_batch and self.start_batch == len(loader):
                # a checkpoint of an older format taken on the last batch of an epoch: that epoch is finished
                self.start_epoch, self.start_batch = self.start_epoch + 1, 0
            for epoch in range(self.start_epoch, cfg.epochs):
                if done:
                    break
                if sampler is not None:
                    sampler.set_epoch(epoch)
                # a checkpoint written in the middle of an epoch: continue behind the batches already trained on
                skip = self.start_batch if epoch == self.start_epoch else 0
                n_batches = len(loader)
                if skip:
                    if skip >= n_batches:
                        # the position does not exist in this run's epochs: batch size, world size or dataset length
                        # changed since the checkpoint was written (a checkpoint taken on the last batch of an epoch
                        # is recorded as the start of the next one)
                        rank_zero_print(f"Resume position batch {skip} is beyond this run's {n_batches} batches per "
                                        f"epoch (the checkpoint's run had {self._saved_batches_per_epoch or 'an unknown number'}); "
                                        f"epoch {epoch} starts from its first batch")
                        skip = 0
                    elif sampler is not None:
                        sampler.set_start(skip * cfg.batch_size)
                    else:
                        loader.skip_next(skip)
                bar = tqdm(loader, disable=not (cfg.progress and env.is_main), desc=f"epoch {epoch}" if cfg.epochs > 1 else None,
                           total=n_batches - skip)
                micro = skip
                for batch in bar:
                    # micro-batches [group_start, group_start + group_size) feed one optimizer step; the last group of
                    # an epoch may be short (its loss is scaled by its own size and it still steps: nothing leaks into
                    # the next epoch, nothing is dropped)
                    group_start = (micro // ga) * ga
                    group_size = min(ga, n_batches - group_start)
                    micro += 1
                    accumulate = (micro - group_start) < group_size
                    if tuner is not None and steps_this_run == 1 and not accumulate:
                        # the second optimizer step of this run is an ordinary step with timing marks behind its
                        # gradients (the first one re-plans the buckets from the observed gradient order)
                        tuner.arm()
                        tuner_armed = True
                    last_loss = self.train_step(batch, accumulate, group_size)
                    if accumulate:
                        continue
                    if tuner_armed:
                        self.bucket_decision = tuner.finish()  # every rank, same step; re-plans the buckets
                        tuner, tuner_armed = None, False
                        self._zero_grad()
                    batch_in_epoch = micro
                    self.global_step += 1
                    steps_this_run += 1
                    meter.step(_batch_len(batch) * group_size)
                    dog.beat(self.global_step)
                    maybe_inject_fault(env.rank, self.global_step)
                    if cfg.log_every and self.global_step % cfg.log_every == 0:
                        lv = float(last_loss) * group_size
                        loss_meter.update(lv)
                        res.losses.append(lv)
                        # (the kernels' flag catches an Inf / NaN that BatchNorm + ReLU would otherwise turn into zeros
                        # before it reaches the loss)
                        if cfg.check_finite and (not math.isfinite(lv) or ops.nonfinite_seen(self.device)):
                            pending |= 2
                        if env.is_main:
                            bar.set_postfix(loss=f"{lv:.4f}", refresh=False)
                    if self.stopper.requested:
                        pending |= 1
                    if not env.distributed or self.global_step % check_every == 0:
                        # every rank evaluates this condition at the same step, so the collective inside _agree is
                        # entered by all of them; flags raised between two agreement steps wait for the next one
                        agreed = self._agree(pending)
                        if agreed & 2:
                            raise FloatingPointError(f"non-finite loss at or before step {self.global_step} on some rank")
                        if agreed & 1:
                            stopped = done = True
                    if cfg.save_every and cfg.checkpoint_path and self.global_step % cfg.save_every == 0:
                        self.save(epoch, batch_in_epoch, n_batches)
                    if cfg.steps and self.global_step >= cfg.steps:
                        done = True
                    if done:
                        break
                res.epochs = epoch + 1 - self.start_epoch
                if not done:
                    batch_in_epoch = 0
                if done:
                    break
            if env.distributed and check_every > 1:
                # flags raised since the last agreement step (a NaN in the final steps, a stop request that arrived
                # late) must not be lost when the loop ends: one more agreement, entered by every rank
                if self.stopper.requested:
                    pending |= 1
                agreed = self._agree(pending)
                if agreed & 2:
                    raise FloatingPointError(f"non-finite loss at or before step {self.global_step} on some rank")
                if agreed & 1:
                    stopped = True
        finally:
            dog.stop()
            self.stopper.uninstall()
        meter.stop()
        t1 = time.perf_counter()
        print("Training complete", flush=True)  # contract line 4 (main.py:82)
        print(f"Elapsed time: {t1 - t0}", flush=True)  # contract line 5 (main.py:84)

        res.steps = steps_this_run
        res.elapsed_s = t1 - t0
        res.samples_per_sec = meter.samples_per_sec
        res.ms_per_step = meter.ms_per_step
        res.interrupted = stopped
        if last_loss is not None:
            res.final_loss = float(last_loss) * min(ga, max(1, len(loader)))
        unit = "tokens/sec" if self.kind == "lm" else "images/sec"
        scale = cfg.seq_len if self.kind == "lm" else 1
        if meter.timed_steps:
            rank_zero_print(
                f"Throughput: {meter.samples_per_sec * scale:.1f} {unit} over {meter.timed_steps} steps "
                f"({meter.ms_per_step:.2f} ms/step, world size {env.world_size}, {self.data_desc})"
            )
        if cfg.evaluate:
            res.eval_loss, res.eval_accuracy = self.evaluate()
            if res.eval_accuracy is not None:
                rank_zero_print(f"Eval: loss {res.eval_loss:.4f}, accuracy {100 * res.eval_accuracy:.2f}%")
            else:
                rank_zero_print(f"Eval: loss {res.eval_loss:.4f}")
        if cfg.checkpoint_path:
            if done and batch_in_epoch:
                # stopped inside an epoch: a resume continues behind this batch (or with the next epoch when it was
                # the epoch's last one)
                self.save(epoch, batch_in_epoch, len(loader))
            else:
                self.save(epoch + 1, 0, len(loader))
            rank_zero_print(f"Checkpoint written to {cfg.checkpoint_path}")
        if cfg.profile:
            if self.trace and env.is_main:
                self.trace.dump(cfg.profile)
            rank_zero_print(self.timer.format())
        return res

    def evaluate(self):
        """Mean loss and top-1 accuracy over the dataset, reduced across ranks."""
        loss_sum, correct, count = all_reduce_values(self.env, self.evaluate_totals(), "sum")
        count = max(count, 1.0)
        return loss_sum / count, (correct / count if self.kind == "image" else None)

    @torch.no_grad()
    def evaluate_totals(self):
        """``[loss_sum, correct, count]`` over this rank's shard, every sample of the dataset counted by exactly one rank.

        The sharded sampler pads the last positions of some shards by wrapping around so that all ranks see the same
        number of samples; those repeats are cut off here (``valid_samples``), so the result does not depend on the
        world size."""
        env = self.env
        was_training = self.model.training
        self.model.eval()
        loader, sampler = build_loader(self.dataset, self.cfg.batch_size, self.cfg.num_workers, env.rank, env.world_size,
                                       shuffle=False, pin_memory=self.device.type == "cuda")
        valid = sampler.valid_samples
        if self.device.type == "cuda":  # uploads (and uint8 -> bf16 conversion) of batch k+1 overlap batch k
            loader = DevicePrefetcher(loader, self.device, convert=lambda u8: ops.image_batch_to_device_layout(u8, self.dtype))
        tot = torch.zeros(3, dtype=torch.float64, device=self.device)  # loss_sum, correct, count
        seen = 0
        for batch in loader:
            b = _batch_len(batch)
            n = max(0, min(b, valid - seen))
            seen += b
            if n == 0:
                continue
            if self.kind == "lm":
                ids, targets = batch
                loss = self.raw_model(ids[:n].to(self.device), targets[:n].to(self.device))
                tot[0] += loss.double() * n
                tot[2] += n
            else:
                imgs, lbls = batch
                lbls = lbls.to(self.device)[:n]
                if imgs.device != self.device or imgs.dtype != self.dtype:
                    imgs = prepare_image_batch(imgs, self.device, self.dtype)
                preds = self.raw_model(imgs[:n] if n < b else imgs)
                loss = ops.softmax_cross_entropy(preds, lbls)
                tot[0] += loss.double() * n
                tot[1] += (preds.argmax(dim=1) == lbls).sum().double()
                tot[2] += n
        self.model.train(was_training)
        return tot.tolist()

    def save(self, epoch: int, batch_in_epoch: int = 0, batches_per_epoch: int = 0) -> None:
        """``(epoch, batch_in_epoch)`` is where a resumed run continues.  A position behind the last batch of an epoch
        is the first batch of the next one: recorded that way, a resume neither replays the finished epoch nor
        trains one epoch too many."""
        if batches_per_epoch and batch_in_epoch >= batches_per_epoch:
            epoch, batch_in_epoch = epoch + 1, 0
        ckpt.save_checkpoint(self.cfg.checkpoint_path, self.model, self.env, self.optimizer, epoch, self.global_step,
                             extra={"batch_in_epoch": int(batch_in_epoch), "batches_per_epoch": int(batches_per_epoch)},
                             keep=self.cfg.keep_checkpoints)

    def close(self) -> None:
        if hasattr(self.model, "reducer"):
            self.model.reducer.close()
        if self._owns_env:
            shutdown(self.env)


class _null:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


def _batch_len(batch) -> int:
    return int(batch[0].shape[0])


def run_training(cfg: TrainConfig) -> TrainResult:
    trainer = Trainer(cfg)
    try:
        return trainer.fit()
    finally:
        trainer.close()
