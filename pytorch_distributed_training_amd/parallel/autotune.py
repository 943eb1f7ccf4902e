"""Bucket capacities chosen from measurements of the node the job runs on (``--bucket-mb auto``).

Stock DDP ships one default (``bucket_cap_mb=25``, the value behind
/root/reference/src/main.py:53) that was tuned for NVLink/NVSwitch rings.  On an MI355X node every
GPU has seven point-to-point xGMI links; what a bucket costs there - a fixed launch/latency part and
a per-byte part that depends on how many links RCCL's channels keep busy for that message size - is
measured, not assumed:

1. the all-reduce of this process's own communicator is timed at a few message sizes (every rank takes
   part; the result is a piecewise-linear cost ``t(bytes)``);
2. one training step is run with a timing event behind every parameter's gradient, which gives the
   moment each gradient becomes ready relative to the start of backward and the length of backward;
3. for every candidate (main capacity, final-bucket capacity) the bucket plan is laid out exactly as the
   reducer would, and the communication stream is simulated: a bucket starts when its last gradient is
   ready and the previous bucket has finished; the *exposed* time is what is left of the last bucket
   after backward has ended;
4. rank 0's best candidate (least exposed time; ties go to fewer, larger buckets) is sent to every rank,
   and the reducer re-plans its buckets with it.

The decision function is pure (``choose_capacities``) so it is unit-tested on synthetic timings.
"""

from __future__ import annotations

import os
import time
from dataclasses import dataclass
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

from ..utils.logging import get_logger
from .reducer import MiB, plan_buckets

log = get_logger("pdt.autotune")

MAIN_CANDIDATES_MB = (4.0, 8.0, 16.0, 25.0, 32.0, 50.0, 64.0, 100.0, 200.0)
LAST_CANDIDATES_MB = (0.0, 0.25, 1.0, 4.0)
PROBE_SIZES_MB = (0.25, 1.0, 4.0, 16.0, 64.0)


@dataclass
class CommCost:
    """Piecewise-linear time of one collective as a function of its size, from measured ``(bytes, microseconds)``
    points; linear extrapolation beyond the ends (never below the smallest measured time's latency part)."""

    sizes: Tuple[float, ...]
    us: Tuple[float, ...]

    @classmethod
    def fit(cls, sizes: Sequence[float], us: Sequence[float]) -> "CommCost":
        pts = sorted(zip(sizes, us))
        if len(pts) < 2:
            raise ValueError("need at least two measured sizes")
        s = [float(p[0]) for p in pts]
        t = [float(p[1]) for p in pts]
        for i in range(1, len(t)):  # a bigger message never costs less: smooth measurement noise away
            t[i] = max(t[i], t[i - 1])
        return cls(tuple(s), tuple(t))

    def __call__(self, nbytes: float) -> float:
        s, t = self.sizes, self.us
        if nbytes <= s[0]:
            # below the smallest probe: latency-bound, scale only the (small) per-byte part
            slope = (t[1] - t[0]) / (s[1] - s[0])
            return max(t[0] - slope * (s[0] - nbytes), 0.5 * t[0])
        for i in range(1, len(s)):
            if nbytes <= s[i]:
                w = (nbytes - s[i - 1]) / (s[i] - s[i - 1])
                return t[i - 1] + w * (t[i] - t[i - 1])
        slope = (t[-1] - t[-2]) / (s[-1] - s[-2])
        return t[-1] + slope * (nbytes - s[-1])

    @property
    def latency_us(self) -> float:
        return self.us[0]

    @property
    def peak_gbps(self) -> float:
        return max(s / (u * 1e3) for s, u in zip(self.sizes, self.us) if u > 0)


def simulate_exposed_us(bucket_bytes: Sequence[float], bucket_ready_us: Sequence[float], backward_us: float,
                        cost: Callable[[float], float]) -> float:
    """Time the communication stream is still busy after backward has ended, for buckets launched in order."""
    end = 0.0
    for nbytes, ready in zip(bucket_bytes, bucket_ready_us):
        end = max(end, ready) + cost(nbytes)
    return max(0.0, end - backward_us)


def choose_capacities(
    numels: Sequence[int],
    elem_sizes: Sequence[int],
    order: Sequence[int],
    ready_us: Sequence[float],
    backward_us: float,
    cost: Callable[[float], float],
    wire_scale: float = 1.0,
    first_bucket_mb: float = 1.0,
    main_candidates: Sequence[float] = MAIN_CANDIDATES_MB,
    last_candidates: Sequence[float] = LAST_CANDIDATES_MB,
    tolerance_us: float = 20.0,
) -> Dict:
    """Pick ``(bucket_mb, last_bucket_mb)``.

    ``order``: parameter indices in the order their gradients become ready; ``ready_us[i]``: when parameter ``i``'s
    gradient is ready, measured from the start of backward; ``wire_scale``: bytes on the wire per gradient byte (0.5
    for fp32 gradients sent as bf16).  Among candidates whose modelled exposed time is within ``tolerance_us`` of the
    best one, the one with the fewest buckets wins (every collective occupies compute units while it runs)."""
    table = []
    for main in main_candidates:
        for last in last_candidates:
            plan = plan_buckets(numels, elem_sizes, int(first_bucket_mb * MiB), int(main * MiB), list(order),
                                last_bucket_bytes=int(last * MiB))
            nb = plan.n_buckets
            ready = [0.0] * nb
            for i, b in enumerate(plan.bucket_of):
                ready[b] = max(ready[b], ready_us[i])
            sizes = [bb * wire_scale for bb in plan.bucket_bytes]
            exposed = simulate_exposed_us(sizes, ready, backward_us, cost)
            table.append({"bucket_mb": main, "last_bucket_mb": last, "buckets": nb, "exposed_us": round(exposed, 1)})
    best = min(r["exposed_us"] for r in table)
    ok = [r for r in table if r["exposed_us"] <= best + tolerance_us]
    pick = min(ok, key=lambda r: (r["buckets"], r["exposed_us"], -r["bucket_mb"]))
    return {"bucket_mb": pick["bucket_mb"], "last_bucket_mb": pick["last_bucket_mb"], "buckets": pick["buckets"],
            "exposed_us": pick["exposed_us"], "best_exposed_us": best, "table": table}


# ---------------------------------------------------------------------------------------------- measurements
def measure_comm_cost(env, wire_dtype: torch.dtype = torch.float32, sizes_mb: Sequence[float] = PROBE_SIZES_MB,
                      warmup: int = 3, iters: int = 8) -> CommCost:
    """Time the all-reduce of this process's communicator at ``sizes_mb`` (a collective: every rank calls it)."""
    es = torch.empty((), dtype=wire_dtype).element_size()
    sizes, us = [], []
    for mb in sizes_mb:
        n = max(8, int(mb * MiB) // es // 8 * 8)
        if env.comm is not None:
            buf = torch.zeros(n, dtype=wire_dtype, device=env.device)
            t = env.comm.time_all_reduce(buf, warmup=warmup, iters=iters)
        else:
            buf = torch.zeros(n, dtype=torch.float32)
            for _ in range(warmup):
                dist.all_reduce(buf)
            t0 = time.perf_counter()
            for _ in range(iters):
                dist.all_reduce(buf)
            t = 1e6 * (time.perf_counter() - t0) / iters
        sizes.append(float(n * es))
        us.append(float(t))
        del buf
    return CommCost.fit(sizes, us)


class ReadyProfile:
    """Timing marks behind every parameter's gradient for ONE backward pass of an ordinary training step.

    ``arm()`` before the step, ``collect()`` after it: ``(ready_us per parameter, backward_us)`` with the clock
    starting at the first gradient (forward and loss are shifted away).  GPU: HIP events on the step's stream,
    recorded from a gradient hook that runs after the reducer's own; CPU: the host clock.  Nothing about the step
    changes - the reducer stays switched on, so the ranks' parameters stay identical."""

    def __init__(self, reducer):
        self.reducer = reducer
        self._handles = []
        self._marks: List[Optional[object]] = []
        self._on_gpu = reducer.params[0].is_cuda

    def arm(self) -> "ReadyProfile":
        params = self.reducer.params
        self._marks = [None] * len(params)

        def make(i):
            def hook(_p):
                if self._marks[i] is not None:
                    return  # an accumulation micro-batch: keep the first pass
                if self._on_gpu:
                    ev = torch.cuda.Event(enable_timing=True)
                    ev.record()
                    self._marks[i] = ev
                else:
                    self._marks[i] = time.perf_counter()
            return hook

        self._handles = [p.register_post_accumulate_grad_hook(make(i)) for i, p in enumerate(params)]
        return self

    def disarm(self) -> None:
        for h in self._handles:
            h.remove()
        self._handles = []

    def collect(self):
        self.disarm()
        marks = self._marks
        seen = [i for i, m in enumerate(marks) if m is not None]
        if not seen:
            return [0.0] * len(marks), 0.0
        if self._on_gpu:
            torch.cuda.synchronize()
            # events of one stream complete in record order: the first recorded is the earliest
            base = marks[seen[0]]
            rel = {i: 1e3 * base.elapsed_time(marks[i]) for i in seen}
        else:
            base = marks[seen[0]]
            rel = {i: 1e6 * (marks[i] - base) for i in seen}
        lo = min(rel.values())
        ready = [max(0.0, rel[i] - lo) if i in rel else 0.0 for i in range(len(marks))]
        return ready, max(ready)


def measure_ready_profile(reducer, step: Callable[[], object], sync: Callable[[], None]):
    """Convenience for callers that can afford an extra step: arm, run ``step()``, collect."""
    prof = ReadyProfile(reducer).arm()
    try:
        step()
        sync()
    finally:
        prof.disarm()
    return prof.collect()


class BucketAutoTuner:
    """``--bucket-mb auto``: ``arm()`` before one ordinary (synchronising) training step, ``finish()`` after it, on
    every rank at the same step.  ``finish`` times the communicator, decides on rank 0, distributes the decision over
    the host control plane and re-plans every rank's buckets.  A single rank has nothing to tune: both calls do
    nothing and ``finish`` returns ``None``."""

    def __init__(self, model, env, grad_dtype: str = "fp32", force: Optional[bool] = None):
        """``force`` (default: the ``PDT_FORCE_AUTOTUNE`` environment variable): run the whole procedure in a one-rank
        group too - nothing can be gained there, but it executes every step of it (probe collectives on the RCCL
        communicator, gradient timing marks, decision, re-planning under a live optimizer) on a single GPU."""
        self.model, self.env, self.grad_dtype = model, env, grad_dtype
        if force is None:
            force = os.environ.get("PDT_FORCE_AUTOTUNE", "0") not in ("", "0")
        self.active = bool(env.distributed and (env.world_size > 1 or force) and hasattr(model, "reducer"))
        self._profile: Optional[ReadyProfile] = None
        self.decision: Optional[Dict] = None

    def arm(self) -> None:
        if self.active:
            self._profile = ReadyProfile(self.model.reducer).arm()

    def finish(self) -> Optional[Dict]:
        if not self.active or self._profile is None:
            return None
        env, reducer = self.env, self.model.reducer
        ready, backward_us = self._profile.collect()
        self._profile = None
        params = reducer.params
        fp32_on_bf16_wire = self.grad_dtype == "bf16" and params[0].dtype == torch.float32
        wire = torch.bfloat16 if fp32_on_bf16_wire else params[0].dtype
        sizes = PROBE_SIZES_MB if env.comm is not None else (0.0625, 0.25, 1.0, 4.0)
        cost = measure_comm_cost(env, wire, sizes)
        first = reducer.first_bucket_bytes / MiB
        decision = None
        if env.rank == 0:
            try:
                order = sorted(range(len(params)), key=lambda i: ready[i])
                decision = choose_capacities(
                    [p.numel() for p in params], [p.element_size() for p in params], order, ready, backward_us, cost,
                    wire_scale=0.5 if fp32_on_bf16_wire else 1.0, first_bucket_mb=first,
                )
                decision.pop("table", None)
                decision["probe_us"] = {f"{s / MiB:g}MiB": round(u, 1) for s, u in zip(cost.sizes, cost.us)}
                decision["latency_us"] = round(cost.latency_us, 1)
                decision["peak_algbw_GBps"] = round(cost.peak_gbps, 1)
                decision["backward_ms"] = round(backward_us / 1e3, 3)
            except Exception as e:  # keep the default plan rather than lose the job over a tuning step
                log.warning("bucket auto-tune failed on rank 0 (%r); keeping %s MiB", e, reducer.bucket_bytes / MiB)
                decision = None
        box = [decision]
        dist.broadcast_object_list(box, src=0)  # host control plane (gloo): one plan for everybody
        decision = box[0]
        if decision is None:
            return None
        decision = dict(decision)
        before = (reducer.bucket_bytes, reducer.last_bucket_bytes)
        after = (int(decision["bucket_mb"] * MiB), int(decision["last_bucket_mb"] * MiB))
        decision["changed"] = before != after
        if decision["changed"]:
            reducer.rebucket(decision["bucket_mb"], first, decision["last_bucket_mb"])
        if env.is_main:
            log.info("bucket auto-tune: %s MiB buckets (final bucket %s MiB, %d buckets), modelled exposed time %.0f us; "
                     "all-reduce latency %.0f us, peak %.1f GB/s; backward %.1f ms", decision["bucket_mb"],
                     decision["last_bucket_mb"], decision["buckets"], decision["exposed_us"], decision["latency_us"],
                     decision["peak_algbw_GBps"], decision["backward_ms"])
        self.decision = decision
        return decision
