"""Failure detection and fault injection.

The reference has none: a dead or hung rank blocks its peers until the collective backend
times out (/root/reference/src/main.py:39-41 passes no timeout; there is no signal handling
anywhere in main.py:26-84).  This module adds

* :class:`StepWatchdog` - a daemon thread that notices when the training loop stops making
  progress, dumps every thread's stack and terminates the process with a distinctive exit
  code so ``torchrun --max-restarts=N`` can restart the job from the last checkpoint;
* :class:`GracefulStop` - turns SIGTERM/SIGINT/SIGUSR1 into a flag the trainer polls, so a
  pre-empted job writes a checkpoint before exiting;
* :func:`maybe_inject_fault` - deterministic fault injection for tests
  (``PDT_FAULT_INJECT=<rank>:<step>:<exit|hang|raise>``).
"""

from __future__ import annotations

import faulthandler
import os
import signal
import sys
import threading
import time
from typing import Optional

WATCHDOG_EXIT_CODE = 75  # EX_TEMPFAIL: "try again" - a restart is expected to succeed


class StepWatchdog:
    def __init__(self, timeout_s: float, rank: int = 0, on_timeout=None, poll_s: Optional[float] = None):
        self.timeout_s = float(timeout_s)
        self.rank = rank
        self.on_timeout = on_timeout
        self.poll_s = poll_s if poll_s is not None else max(min(self.timeout_s / 4.0, 5.0), 0.01)
        self._last = time.monotonic()
        self._step = 0
        self._stop = threading.Event()
        self._thread: Optional[threading.Thread] = None
        self.fired = False

    def start(self) -> "StepWatchdog":
        if self.timeout_s > 0 and self._thread is None:
            self._last = time.monotonic()
            self._thread = threading.Thread(target=self._run, name="pdt-watchdog", daemon=True)
            self._thread.start()
        return self

    def beat(self, step: Optional[int] = None) -> None:
        """Call once per finished step."""
        self._last = time.monotonic()
        if step is not None:
            self._step = step

    def stop(self) -> None:
        self._stop.set()
        if self._thread is not None:
            self._thread.join(timeout=2 * self.poll_s + 1)
            self._thread = None

    def _run(self) -> None:
        while not self._stop.wait(self.poll_s):
            stalled = time.monotonic() - self._last
            if stalled > self.timeout_s:
                self.fired = True
                sys.stderr.write(
                    f"[pdt watchdog] rank {self.rank}: no step finished for {stalled:.1f}s "
                    f"(last step {self._step}); dumping stacks\n"
                )
                try:
                    faulthandler.dump_traceback(file=sys.stderr, all_threads=True)
                except Exception:
                    pass
                if self.on_timeout is not None:
                    self.on_timeout()
                else:
                    os._exit(WATCHDOG_EXIT_CODE)
                return

    def __enter__(self):
        return self.start()

    def __exit__(self, *exc):
        self.stop()


class GracefulStop:
    """Records termination signals; the trainer checks ``requested`` between steps."""

    def __init__(self, signals=(signal.SIGTERM, signal.SIGINT, signal.SIGUSR1)):
        self.requested = False
        self.signum: Optional[int] = None
        self._old = {}
        self._signals = signals

    def _handler(self, signum, _frame):
        self.requested = True
        self.signum = signum

    def install(self) -> "GracefulStop":
        if threading.current_thread() is threading.main_thread():
            for s in self._signals:
                try:
                    self._old[s] = signal.signal(s, self._handler)
                except (ValueError, OSError):
                    pass
        return self

    def uninstall(self) -> None:
        for s, h in self._old.items():
            try:
                signal.signal(s, h)
            except (ValueError, OSError):
                pass
        self._old.clear()


def maybe_inject_fault(rank: int, step: int) -> None:
    """``PDT_FAULT_INJECT=<rank>:<step>:<kind>`` with kind in exit | hang | raise."""
    spec = os.environ.get("PDT_FAULT_INJECT")
    if not spec:
        return
    try:
        r, s, kind = spec.split(":")
        r, s = int(r), int(s)
    except ValueError:
        raise ValueError(f"PDT_FAULT_INJECT must look like '1:10:exit', got {spec!r}")
    if r != rank or s != step:
        return
    if os.environ.get("PDT_FAULT_INJECT_ONCE_FILE"):
        marker = os.environ["PDT_FAULT_INJECT_ONCE_FILE"]
        if os.path.exists(marker):
            return  # already fired in a previous incarnation of this job
        open(marker, "w").close()
    if kind == "exit":
        sys.stderr.write(f"[pdt fault] rank {rank} exiting at step {step}\n")
        os._exit(13)
    if kind == "hang":
        sys.stderr.write(f"[pdt fault] rank {rank} hanging at step {step}\n")
        while True:
            time.sleep(3600)
    if kind == "raise":
        raise RuntimeError(f"injected fault on rank {rank} at step {step}")
    raise ValueError(f"unknown fault kind {kind!r}")
