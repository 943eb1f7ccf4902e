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
    """``timeout_s > 0``: no finished step for that long is a failure.  ``probe``: a callable polled every
    ``probe_every_s`` that returns ``None`` while the communication backend is healthy and a description once it
    has reported an asynchronous error (``RcclComm.async_error``: a dead peer is then noticed within seconds
    instead of after the step timeout).  On either failure: stacks are dumped, ``before_exit`` runs (abort the
    communicator, so kernels blocked in a collective return), and the process ends with ``WATCHDOG_EXIT_CODE`` -
    it is never re-executed in place; a supervisor (``torchrun --max-restarts``) starts a fresh one."""

    def __init__(self, timeout_s: float, rank: int = 0, on_timeout=None, poll_s: Optional[float] = None,
                 probe=None, probe_every_s: float = 2.0, before_exit=None):
        self.timeout_s = float(timeout_s)
        self.rank = rank
        self.on_timeout = on_timeout
        self.probe = probe
        self.before_exit = before_exit
        self.failure: Optional[str] = None
        if poll_s is not None:
            self.poll_s = poll_s
        elif self.timeout_s > 0:
            self.poll_s = max(min(self.timeout_s / 4.0, 5.0), 0.01)
            if probe is not None:
                self.poll_s = min(self.poll_s, max(probe_every_s, 0.01))
        else:
            self.poll_s = max(probe_every_s, 0.01)
        self._last = time.monotonic()
        self._step = 0
        self._stop = threading.Event()
        self._thread: Optional[threading.Thread] = None
        self.fired = False

    def start(self) -> "StepWatchdog":
        if (self.timeout_s > 0 or self.probe is not None) and self._thread is None:
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
            why = None
            if self.timeout_s > 0 and stalled > self.timeout_s:
                why = f"no step finished for {stalled:.1f}s (last step {self._step})"
            elif self.probe is not None:
                try:
                    err = self.probe()
                except Exception as e:  # the probe itself failing is a failure of the backend
                    err = repr(e)
                if err:
                    why = f"communication backend reports: {err} (last step {self._step})"
            if why is not None:
                self.fired = True
                self.failure = why
                sys.stderr.write(f"[pdt watchdog] rank {self.rank}: {why}; dumping stacks\n")
                try:
                    faulthandler.dump_traceback(file=sys.stderr, all_threads=True)
                except Exception:
                    pass
                if self.before_exit is not None:
                    try:
                        self.before_exit()
                    except Exception:
                        pass
                if self.on_timeout is not None:
                    self.on_timeout()
                else:
                    sys.stderr.flush()
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
