"""Throughput and running-average meters.

The reference's only metric is total elapsed seconds
(/root/reference/src/main.py:65,81,84).  The headline metric of this framework is
samples/second for the whole job, so the meter counts *global* samples
(per-rank batch x world size) and excludes warm-up steps.
"""

from __future__ import annotations

import time
from dataclasses import dataclass, field
from typing import Callable, Optional


@dataclass
class AverageMeter:
    """Sum/count average; ``update(v, n)`` weights ``v`` by ``n``."""

    total: float = 0.0
    count: float = 0.0
    last: float = 0.0

    def update(self, value: float, n: float = 1.0) -> None:
        self.last = float(value)
        self.total += float(value) * n
        self.count += n

    @property
    def avg(self) -> float:
        return self.total / self.count if self.count else 0.0

    def reset(self) -> None:
        self.total = self.count = self.last = 0.0


@dataclass
class ThroughputMeter:
    """Samples/sec over the steps after ``warmup_steps``.

    ``sync`` is called before every clock read that bounds the timed window
    (device synchronisation on GPU, no-op on CPU) so queued kernels are charged
    to the window they belong to.
    """

    world_size: int = 1
    warmup_steps: int = 0
    sync: Optional[Callable[[], None]] = None
    clock: Callable[[], float] = time.perf_counter
    steps: int = 0
    timed_steps: int = 0
    timed_samples: int = 0
    _t_start: Optional[float] = field(default=None, repr=False)
    _t_last: Optional[float] = field(default=None, repr=False)

    def _now(self) -> float:
        if self.sync is not None:
            self.sync()
        return self.clock()

    def start(self) -> None:
        """Mark the beginning of the run; with zero warm-up the window opens here."""
        if self.warmup_steps == 0:
            self._t_start = self._now()

    def step(self, local_batch: int) -> None:
        """Record one finished optimisation step of ``local_batch`` samples per rank."""
        self.steps += 1
        if self.steps == self.warmup_steps:
            self._t_start = self._now()
            return
        if self.steps > self.warmup_steps:
            if self._t_start is None:  # start() was not called
                self._t_start = self.clock()
                return
            self.timed_steps += 1
            self.timed_samples += local_batch * self.world_size

    def stop(self) -> None:
        self._t_last = self._now()

    @property
    def elapsed(self) -> float:
        if self._t_start is None:
            return 0.0
        end = self._t_last if self._t_last is not None else self.clock()
        return max(end - self._t_start, 0.0)

    @property
    def samples_per_sec(self) -> float:
        e = self.elapsed
        return self.timed_samples / e if e > 0 and self.timed_samples else 0.0

    @property
    def ms_per_step(self) -> float:
        return 1e3 * self.elapsed / self.timed_steps if self.timed_steps else 0.0

    @staticmethod
    def scaling_efficiency(throughput_n: float, n: int, throughput_1: float) -> float:
        """throughput(N) / (N * throughput(1))."""
        return throughput_n / (n * throughput_1) if throughput_1 > 0 and n > 0 else 0.0
