"""Phase timers and a lightweight tracer.

The reference has no profiling beyond one ``perf_counter`` pair
(/root/reference/src/main.py:65,81).  ``StepTimer`` measures named phases of a
training step with HIP events (GPU) or wall clock (CPU); ``ChromeTrace`` dumps
the phases as a chrome://tracing JSON so a run can be inspected without
rocprofv3.
"""

from __future__ import annotations

import json
import os
import time
from collections import defaultdict
from contextlib import contextmanager
from typing import Dict, List, Optional

import torch


class StepTimer:
    """Accumulates per-phase durations.  Cheap enough to leave on in training."""

    def __init__(self, device: Optional[torch.device] = None, enabled: bool = True):
        self.enabled = enabled
        self.use_events = bool(device is not None and torch.device(device).type == "cuda")
        self.totals: Dict[str, float] = defaultdict(float)
        self.counts: Dict[str, int] = defaultdict(int)
        self._pending: List[tuple] = []

    @contextmanager
    def phase(self, name: str):
        if not self.enabled:
            yield
            return
        if self.use_events:
            start = torch.cuda.Event(enable_timing=True)
            end = torch.cuda.Event(enable_timing=True)
            start.record()
            try:
                yield
            finally:
                end.record()
                self._pending.append((name, start, end))
        else:
            t0 = time.perf_counter()
            try:
                yield
            finally:
                self.totals[name] += time.perf_counter() - t0
                self.counts[name] += 1

    def flush(self) -> None:
        """Resolve pending GPU events (synchronises once)."""
        if self._pending:
            torch.cuda.synchronize()
            for name, start, end in self._pending:
                self.totals[name] += start.elapsed_time(end) * 1e-3
                self.counts[name] += 1
            self._pending.clear()

    def summary(self) -> Dict[str, Dict[str, float]]:
        self.flush()
        return {
            k: {"total_s": v, "count": self.counts[k], "mean_ms": 1e3 * v / max(self.counts[k], 1)}
            for k, v in self.totals.items()
        }

    def format(self) -> str:
        rows = self.summary()
        width = max((len(k) for k in rows), default=4)
        lines = [f"{'phase':<{width}}  count  mean_ms   total_s"]
        for k, r in sorted(rows.items(), key=lambda kv: -kv[1]["total_s"]):
            lines.append(f"{k:<{width}}  {r['count']:5d}  {r['mean_ms']:7.3f}  {r['total_s']:8.3f}")
        return "\n".join(lines)


class ChromeTrace:
    """Host-side span recorder writing the chrome://tracing event format."""

    def __init__(self, rank: int = 0):
        self.rank = rank
        self.events: List[dict] = []
        self._t0 = time.perf_counter()

    @contextmanager
    def span(self, name: str, **args):
        ts = (time.perf_counter() - self._t0) * 1e6
        try:
            yield
        finally:
            dur = (time.perf_counter() - self._t0) * 1e6 - ts
            self.events.append(
                {"name": name, "ph": "X", "ts": ts, "dur": dur, "pid": self.rank, "tid": 0, "args": args}
            )

    def instant(self, name: str, **args) -> None:
        ts = (time.perf_counter() - self._t0) * 1e6
        self.events.append({"name": name, "ph": "i", "ts": ts, "pid": self.rank, "tid": 0, "s": "p", "args": args})

    def dump(self, path: str) -> str:
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        with open(path, "w") as f:
            json.dump({"traceEvents": self.events, "displayTimeUnit": "ms"}, f)
        return path
