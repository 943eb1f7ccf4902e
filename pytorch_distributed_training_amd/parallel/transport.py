"""Bucket all-reduce transports for :class:`GradReducer`.

A transport turns "bucket k is complete" into an asynchronous all-reduce and later makes
the consumer of the gradients wait for it.  Three exist:

* :class:`RcclTransport`    - own RCCL communicator on a dedicated HIP stream
  (``csrc/reducer/reducer_rccl.cpp``); the bucket launch and wait run entirely in C++.
* :class:`TorchDistTransport` - ``torch.distributed`` async all-reduce (gloo on CPU); the C++
  core calls back into Python.
* :class:`LoopbackTransport` - ranks are threads of one process (``csrc/reducer/loopback.cpp``);
  used by CPU tests.

Role parity: the NCCL process group that stock DDP drives at
/root/reference/src/main.py:39-41,53.
"""

from __future__ import annotations

import ctypes
import os
from typing import List, Optional, Sequence

import torch
import torch.distributed as dist

from .. import native

DTYPE_CODE = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2, torch.float64: 3}


class Transport:
    """Interface: ``bind(buckets)`` returns ``(launch_ptr, wait_ptr, user_ptr)`` for the C++ core."""

    world_size: int = 1
    averages_in_flight: bool = True  # False => reducer divides by world size after the wait
    bucket_multiple: int = 1  # bucket lengths (in elements) must be multiples of this
    wire_dtype: Optional[torch.dtype] = None  # element type on the wire when it differs from the bucket's

    def wire_bytes(self, bucket: torch.Tensor) -> int:
        """Bytes one collective over ``bucket`` hands to the backend."""
        es = bucket.element_size()
        if self.wire_dtype is not None and bucket.dtype == torch.float32:
            es = torch.empty((), dtype=self.wire_dtype).element_size()
        return bucket.numel() * es

    def bind(self, buckets: Sequence[torch.Tensor]):
        raise NotImplementedError

    def post_finalize(self) -> None:
        """Called once after every bucket has been waited for."""

    def broadcast_(self, tensor: torch.Tensor, src: int = 0) -> None:
        raise NotImplementedError

    def close(self) -> None:
        pass


class TorchDistTransport(Transport):
    """``dist.all_reduce(async_op=True)`` per bucket; SUM on the wire, mean applied after the wait.

    ``wire_dtype=torch.bfloat16``: an fp32 bucket is divided by the world size and cast to bf16 before the
    collective and cast back after it (same arithmetic as the RCCL transport's pack / unpack kernels)."""

    def __init__(self, group: Optional[dist.ProcessGroup] = None, wire_dtype: Optional[torch.dtype] = None):
        self.group = group
        self.world_size = dist.get_world_size(group)
        self.averages_in_flight = False
        self.wire_dtype = wire_dtype if wire_dtype == torch.bfloat16 else None
        self._buckets: List[torch.Tensor] = []
        self._wires: List[Optional[torch.Tensor]] = []
        self._works: List[Optional[dist.Work]] = []
        self._cb_launch = None
        self._cb_wait = None

    def bind(self, buckets):
        self._buckets = list(buckets)
        self._works = [None] * len(self._buckets)
        self._wires = [None] * len(self._buckets)

        def launch(_user, b, _stream):
            buf = self._buckets[b]
            if self.wire_dtype is not None and buf.dtype == torch.float32:
                buf = self._wires[b] = (buf / self.world_size).to(self.wire_dtype)
            self._works[b] = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

        def wait(_user, b, _stream):
            w = self._works[b]
            if w is not None:
                w.wait()
                self._works[b] = None
                if self._wires[b] is not None:
                    self._buckets[b].copy_(self._wires[b])
                    self._wires[b] = None
                else:
                    self._buckets[b].div_(self.world_size)

        # keep references: ctypes callbacks must outlive the reducer
        self._cb_launch = native.BUCKET_FN(launch)
        self._cb_wait = native.BUCKET_FN(wait)
        return (
            ctypes.cast(self._cb_launch, ctypes.c_void_p),
            ctypes.cast(self._cb_wait, ctypes.c_void_p),
            ctypes.c_void_p(None),
        )

    def broadcast_(self, tensor, src=0):
        dist.broadcast(tensor, src=src, group=self.group)


class LoopbackFabric:
    """Shared state of an in-process "cluster" whose ranks are threads."""

    def __init__(self, world_size: int):
        self.world_size = world_size
        self._lib = native.core()
        self._h = self._lib.pdt_loopback_create(world_size)
        if not self._h:
            raise ValueError("world_size must be positive")

    def allreduce_(self, rank: int, tensor: torch.Tensor, average: bool = True, wire_bf16: bool = False) -> None:
        assert tensor.dtype == torch.float32 and tensor.is_contiguous() and tensor.device.type == "cpu"
        self._lib.pdt_loopback_allreduce(self._h, rank, tensor.data_ptr(), tensor.numel(), int(average) | (2 if wire_bf16 else 0))

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.pdt_loopback_destroy(self._h)
            self._h = None


class LoopbackTransport(Transport):
    def __init__(self, fabric: LoopbackFabric, rank: int, wire_dtype: Optional[torch.dtype] = None,
                 algorithm: str = "allreduce"):
        self.fabric = fabric
        self.rank = rank
        self.world_size = fabric.world_size
        self.wire_dtype = wire_dtype if wire_dtype == torch.bfloat16 else None
        # the fabric's collective is always the two-phase direct form (reduce a shard, gather all shards); "rs_ag"
        # only adds the RCCL transport's constraint on bucket lengths so the padding logic is exercised
        self.bucket_multiple = 8 * self.world_size if algorithm == "rs_ag" else 1
        self._lib = native.core()
        self._ctx = None
        self._keep = None

    def bind(self, buckets):
        for b in buckets:
            if b.dtype != torch.float32 or b.device.type != "cpu":
                raise TypeError("loopback transport handles fp32 host buckets only")
        n = len(buckets)
        ptrs = (ctypes.c_void_p * n)(*[b.data_ptr() for b in buckets])
        numel = (ctypes.c_int64 * n)(*[b.numel() for b in buckets])
        self._keep = (list(buckets), ptrs, numel)
        if self._ctx:
            self._lib.pdt_loopback_rank_destroy(self._ctx)
        for b in buckets:
            if b.numel() % self.bucket_multiple:
                raise ValueError(f"bucket length {b.numel()} is not a multiple of {self.bucket_multiple}")
        flags = 1 | (2 if self.wire_dtype is not None else 0)
        self._ctx = self._lib.pdt_loopback_rank_create(self.fabric._h, self.rank, n, ptrs, numel, flags)
        return (
            ctypes.c_void_p(self._lib.pdt_loopback_launch_fn()),
            ctypes.c_void_p(self._lib.pdt_loopback_wait_fn()),
            ctypes.c_void_p(self._ctx),
        )

    @property
    def collectives(self) -> int:
        return int(self._lib.pdt_loopback_rank_collectives(self._ctx)) if self._ctx else 0

    def broadcast_(self, tensor, src=0):
        # all-reduce of (tensor if rank == src else 0) is a broadcast
        flat = tensor.detach().reshape(-1).to(torch.float32).contiguous()
        if self.rank != src:
            flat.zero_()
        self.fabric.allreduce_(self.rank, flat, average=False)
        tensor.detach().copy_(flat.view_as(tensor))

    def close(self):
        if self._ctx:
            self._lib.pdt_loopback_rank_destroy(self._ctx)
            self._ctx = None


class RcclError(RuntimeError):
    pass


class RcclComm:
    """Own RCCL communicator + dedicated HIP stream, bootstrapped through torch's store."""

    _counter = 0

    PRIORITIES = {"normal": 0, "high": 1, "low": -1}

    def __init__(self, rank: int, world_size: int, device: torch.device, store=None, priority: Optional[str] = None):
        """``priority`` of the communication stream: ``normal`` (default; ``PDT_COMM_PRIORITY`` overrides), ``high`` or
        ``low``.  A raised priority looks attractive for collectives but is a priority inversion on this hardware: the
        stream mostly *waits* for bucket-ready events, and once the process has more streams than hardware queues
        served at a time (a prefetcher's copy stream is enough) the waiting high-priority queue starves the queues
        that would produce the event and the device hangs."""
        priority = priority or os.environ.get("PDT_COMM_PRIORITY", "normal")
        if priority not in self.PRIORITIES:
            raise ValueError(f"communication stream priority must be one of {sorted(self.PRIORITIES)}")
        self._lib = native.rccl()
        self.rank, self.world_size, self.device = rank, world_size, torch.device(device)
        nbytes = self._lib.pdt_rccl_unique_id_bytes()
        key = f"pdt/rccl_uid/{RcclComm._counter}"
        RcclComm._counter += 1
        if world_size > 1:
            if store is None:
                raise ValueError("a multi-rank RCCL communicator needs a key-value store (set/get) that every rank "
                                 "can reach, to hand out rank 0's unique id")
            if rank == 0:
                buf = ctypes.create_string_buffer(nbytes)
                self._check(self._lib.pdt_rccl_unique_id(buf), "ncclGetUniqueId")
                store.set(key, buf.raw)
                uid = buf.raw
            else:
                uid = bytes(store.get(key))
        else:
            buf = ctypes.create_string_buffer(nbytes)
            self._check(self._lib.pdt_rccl_unique_id(buf), "ncclGetUniqueId")
            uid = buf.raw
        self.priority = priority
        self._h = self._lib.pdt_rccl_comm_create(uid, rank, world_size, self.device.index or 0, self.PRIORITIES[priority])
        if not self._h:
            raise RcclError(f"RCCL communicator creation failed: {self._lib.pdt_rccl_last_error().decode()}")

    def _check(self, rc: int, what: str) -> None:
        if rc != 0:
            raise RcclError(f"{what} failed: {self._lib.pdt_rccl_last_error().decode()}")

    @property
    def stream_handle(self) -> int:
        return int(self._lib.pdt_rccl_comm_stream(self._h) or 0)

    @property
    def rccl_ranks(self) -> int:
        """Number of ranks RCCL itself reports for this communicator (``ncclCommCount``)."""
        return int(self._lib.pdt_rccl_comm_count(self._h))

    def all_reduce_(self, tensor: torch.Tensor, average: bool = False, stream: Optional[int] = None) -> None:
        """In-place all-reduce; ``stream=None`` uses the communicator's own stream (caller must order it)."""
        assert tensor.is_cuda and tensor.is_contiguous()
        self._check(
            self._lib.pdt_rccl_allreduce(
                self._h, tensor.data_ptr(), tensor.numel(), DTYPE_CODE[tensor.dtype], int(average), stream
            ),
            "ncclAllReduce",
        )

    def broadcast_(self, tensor: torch.Tensor, src: int = 0, stream: Optional[int] = None) -> None:
        assert tensor.is_cuda and tensor.is_contiguous()
        self._check(
            self._lib.pdt_rccl_broadcast(self._h, tensor.data_ptr(), tensor.numel() * tensor.element_size(), src, stream),
            "ncclBroadcast",
        )

    # -- collectives on the communication stream, event-ordered against the caller's current stream: everything
    #    already submitted to that stream happens before, everything submitted afterwards happens after; the host
    #    never waits, and inside a hipGraph capture the collective is recorded as a node of the graph
    def ordered_broadcast_(self, tensor: torch.Tensor, src: int = 0) -> None:
        assert tensor.is_cuda and tensor.is_contiguous()
        self._check(
            self._lib.pdt_rccl_ordered(self._h, 0, tensor.data_ptr(), tensor.numel() * tensor.element_size(), 0, src,
                                       current_stream_handle(self.device)),
            "ncclBroadcast",
        )

    def ordered_all_reduce_(self, tensor: torch.Tensor, op: str = "sum") -> None:
        assert tensor.is_cuda and tensor.is_contiguous()
        kind = {"sum": 1, "mean": 2, "max": 3, "min": 4}[op]
        self._check(
            self._lib.pdt_rccl_ordered(self._h, kind, tensor.data_ptr(), tensor.numel(), DTYPE_CODE[tensor.dtype], 0,
                                       current_stream_handle(self.device)),
            "ncclAllReduce",
        )

    def time_all_reduce(self, tensor: torch.Tensor, warmup: int = 5, iters: int = 20) -> float:
        """Mean microseconds per in-place all-reduce of ``tensor`` on the communication stream."""
        torch.cuda.synchronize(self.device)
        us = self._lib.pdt_rccl_time_allreduce(
            self._h, tensor.data_ptr(), tensor.numel(), DTYPE_CODE[tensor.dtype], warmup, iters
        )
        if us < 0:
            raise RcclError(self._lib.pdt_rccl_last_error().decode())
        return float(us)

    def synchronize(self) -> None:
        self._check(self._lib.pdt_rccl_stream_sync(self._h), "hipStreamSynchronize")

    # -- failure detection
    def async_error(self) -> Optional[str]:
        """``None`` while the communicator is healthy, else RCCL's description of the asynchronous error (a peer
        died, a link failed).  Safe to call from a monitoring thread."""
        if not getattr(self, "_h", None):
            return None
        rc = self._lib.pdt_rccl_comm_async_error(self._h)
        return None if rc == 0 else self._lib.pdt_rccl_last_error().decode()

    def abort(self) -> None:
        """``ncclCommAbort``: outstanding collectives are abandoned so blocked kernels return and the process can exit."""
        if getattr(self, "_h", None):
            self._lib.pdt_rccl_comm_abort(self._h)

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._lib.pdt_rccl_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def current_stream_handle(device: Optional[torch.device] = None) -> int:
    """Raw handle of the HIP stream that is current on the calling thread (0 on a box without GPUs)."""
    if not torch.cuda.is_available():
        return 0
    index = device.index if (device is not None and device.index is not None) else torch.cuda.current_device()
    if _raw_stream is not None:
        return int(_raw_stream(index))
    return int(torch.cuda.current_stream(index).cuda_stream)


FLAG_AVERAGE, FLAG_SCATTER_GATHER, FLAG_KEEP_ONE_RANK, FLAG_COUNT_ON_DEVICE = 1, 2, 4, 8


class RcclTransport(Transport):
    """Bucket collectives on the communicator's dedicated stream, event-ordered in C++ against the stream that is
    current when a bucket becomes ready.

    ``wire_dtype=torch.bfloat16``: fp32 buckets cross the links as bf16 (pack / unpack kernels on the communication
    stream, pre-divided by the world size; the gradients a rank accumulates stay fp32).
    ``algorithm="rs_ag"``: reduce-scatter + all-gather instead of all-reduce.
    ``keep_one_rank``: issue the collective even in a one-rank group (default: the ``PDT_SELF_ALLREDUCE``
    environment variable) - on one GPU this shows what RCCL's kernels on the priority stream cost the step.
    ``count_on_device``: keep a device-side counter of executed bucket collectives (``device_collectives``;
    also switched on by ``PDT_RCCL_COUNT=1``).
    """

    def __init__(self, comm: RcclComm, wire_dtype: Optional[torch.dtype] = None, algorithm: str = "allreduce",
                 keep_one_rank: Optional[bool] = None, count_on_device: bool = False):
        if wire_dtype not in (None, torch.bfloat16, torch.float32):
            raise ValueError("wire_dtype must be None (the bucket's own type), torch.float32 or torch.bfloat16")
        if algorithm not in ("allreduce", "rs_ag"):
            raise ValueError("algorithm must be 'allreduce' or 'rs_ag'")
        self.comm = comm
        self.world_size = comm.world_size
        self.wire_dtype = wire_dtype if wire_dtype == torch.bfloat16 else None
        self.algorithm = algorithm
        # shards of a reduce-scatter start on 16-byte boundaries for either element size; a wire copy needs whole
        # 8-element vectors
        self.bucket_multiple = 8 * self.world_size if algorithm == "rs_ag" else 8
        if keep_one_rank is None:
            keep_one_rank = os.environ.get("PDT_SELF_ALLREDUCE", "0") == "1"
        self.keep_one_rank = bool(keep_one_rank)
        self.count_on_device = bool(count_on_device) or os.environ.get("PDT_RCCL_COUNT", "0") == "1"
        self._lib = native.rccl()
        self._state = None
        self._keep = None

    def bind(self, buckets):
        n = len(buckets)
        for b in buckets:
            if not b.is_cuda or b.dtype not in DTYPE_CODE:
                raise TypeError("RCCL transport needs GPU buckets of fp32/bf16/fp16")
            if b.numel() % self.bucket_multiple:
                raise ValueError(f"bucket length {b.numel()} is not a multiple of {self.bucket_multiple}")
        wires = [
            torch.empty(b.numel(), dtype=torch.bfloat16, device=b.device)
            if (self.wire_dtype == torch.bfloat16 and b.dtype == torch.float32) else None
            for b in buckets
        ]
        ptrs = (ctypes.c_void_p * n)(*[b.data_ptr() for b in buckets])
        wire_ptrs = (ctypes.c_void_p * n)(*[w.data_ptr() if w is not None else None for w in wires])
        counts = (ctypes.c_int64 * n)(*[b.numel() for b in buckets])
        dtypes = (ctypes.c_int32 * n)(*[DTYPE_CODE[b.dtype] for b in buckets])
        if self._state:
            self.comm.synchronize()  # nothing of the old event set may still be in flight
            self._lib.pdt_rccl_buckets_destroy(self._state)
        self._keep = (list(buckets), wires, ptrs, wire_ptrs, counts, dtypes)
        flags = FLAG_AVERAGE
        flags |= FLAG_SCATTER_GATHER if self.algorithm == "rs_ag" else 0
        flags |= FLAG_KEEP_ONE_RANK if self.keep_one_rank else 0
        flags |= FLAG_COUNT_ON_DEVICE if self.count_on_device else 0
        self._state = self._lib.pdt_rccl_buckets_create(self.comm._h, n, ptrs, wire_ptrs, counts, dtypes, flags)
        if not self._state:
            raise RcclError(self._lib.pdt_rccl_last_error().decode())
        return (
            ctypes.c_void_p(self._lib.pdt_rccl_launch_fn()),
            ctypes.c_void_p(self._lib.pdt_rccl_wait_fn()),
            ctypes.c_void_p(self._state),
        )

    @property
    def device_collectives(self) -> int:
        """Bucket collectives executed on the device so far, hipGraph replays included (``count_on_device`` only)."""
        return int(self._lib.pdt_rccl_buckets_device_count(self._state)) if self._state else -1

    @property
    def bytes_on_wire(self) -> int:
        return int(self._lib.pdt_rccl_buckets_wire_bytes(self._state)) if self._state else 0

    def post_finalize(self):
        if self._state and self._lib.pdt_rccl_buckets_errors(self._state):
            raise RcclError(f"RCCL bucket collective failed: {self._lib.pdt_rccl_last_error().decode()}")

    def broadcast_(self, tensor, src=0):
        """Broadcast on the communication stream, ordered against the current stream (no host wait)."""
        self.comm.ordered_broadcast_(tensor, src)

    def close(self):
        if self._state:
            try:
                self.comm.synchronize()
            except RcclError:
                pass
            self._lib.pdt_rccl_buckets_destroy(self._state)
            self._state = None
