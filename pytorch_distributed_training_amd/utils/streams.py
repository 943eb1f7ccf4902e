"""Census of the HIP streams a training process owns.

A process is served ``GPU_MAX_HW_QUEUES`` hardware queues at a time (4 unless the environment says otherwise).
With more streams than queues, streams share a queue, and a stream that spends its life waiting on events of
another one (the communication stream waits for "bucket ready" events of the step's stream) can be scheduled
ahead of the stream that would produce the event: the device stops.  The framework therefore keeps to a fixed
budget - the step's stream, the input pipeline's copy stream, the one communication stream, and the capture
stream of a recorded step - and this module makes that budget checkable at run time and in tests.

What is counted: the default stream (1), every live ``torch.cuda.Stream`` object created since
:func:`install_stream_census` (the prefetcher's copy stream, the side stream ``torch.cuda.graph`` captures on,
anything user code adds), the streams owned by this package's native libraries, and the streams of
``torch.distributed`` NCCL process groups (two per group: collectives run on an internal stream of the
communicator) - of which a process built by ``init_distributed`` has none.

Role parity: the stream set stock DDP + ProcessGroupNCCL would bring up implicitly at
/root/reference/src/main.py:39-41,53.
"""

from __future__ import annotations

import os
import threading
import weakref
from typing import Dict

import torch

_lock = threading.Lock()
_live: "weakref.WeakSet" = weakref.WeakSet()
_installed = False
_created = 0


def hw_queue_budget() -> int:
    """Hardware queues the runtime gives this process (``GPU_MAX_HW_QUEUES``, default 4)."""
    try:
        return max(1, int(os.environ.get("GPU_MAX_HW_QUEUES", "4")))
    except ValueError:
        return 4


def install_stream_census() -> None:
    """Start counting ``torch.cuda.Stream`` objects (idempotent).  Streams created earlier are not seen, so call
    this before the trainer or the benchmark builds anything."""
    global _installed
    with _lock:
        if _installed:
            return
        _installed = True
        cls = torch.cuda.Stream
        orig_new = cls.__new__

        def counted_new(klass, *args, **kwargs):
            global _created
            obj = orig_new(klass, *args, **kwargs)
            # wrappers around an existing handle (ExternalStream, stream_ptr=...) and the objects that merely describe
            # the current / default stream are views, not new streams
            if klass is cls and "stream_ptr" not in kwargs and "stream_id" not in kwargs:
                try:
                    _live.add(obj)
                    _created += 1
                except TypeError:
                    pass
            return obj

        cls.__new__ = staticmethod(counted_new)


def _native_streams() -> int:
    try:
        from .. import native

        if not (native.LIB_DIR / "libpdt_rccl.so").exists():
            return 0
        return int(native.rccl().pdt_rccl_live_streams())
    except Exception:
        return 0


def _torch_nccl_groups() -> int:
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return 0
    n = 0
    try:
        groups = list(dist.distributed_c10d._world.pg_map.keys())
    except Exception:
        groups = [dist.group.WORLD]
    for g in groups:
        try:
            if "nccl" in str(dist.get_backend(g)):
                n += 1
        except Exception:
            pass
    return n


def stream_census() -> Dict[str, int]:
    """``{"default", "torch", "native", "torch_nccl", "total", "budget"}`` for the calling process."""
    torch_streams = len(_live)
    native_streams = _native_streams()
    nccl_groups = _torch_nccl_groups()
    total = 1 + torch_streams + native_streams + 2 * nccl_groups
    return {"default": 1, "torch": torch_streams, "native": native_streams, "torch_nccl": 2 * nccl_groups,
            "total": total, "budget": hw_queue_budget()}


def check_stream_budget(where: str = "", strict: bool = False) -> Dict[str, int]:
    """Warn (or raise, ``strict``) when the process owns more streams than hardware queues."""
    c = stream_census()
    if c["total"] > c["budget"]:
        msg = (f"{where or 'process'} owns {c['total']} HIP streams ({c}) but is served {c['budget']} hardware queues: "
               "streams will share queues, and a stream that waits on another one's events can starve it")
        if strict:
            raise RuntimeError(msg)
        from .logging import get_logger

        get_logger("pdt.streams").warning(msg)
    return c
