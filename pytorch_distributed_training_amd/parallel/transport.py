"""Bucket all-reduce transports for :class:`GradReducer`.

A transport turns "bucket k is complete" into an asynchronous all-reduce and later makes
the consumer of the gradients wait for it.  Three exist:

* :class:`RcclTransport`    - own RCCL communicator on a dedicated high-priority HIP stream
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
from typing import List, Optional, Sequence

import torch
import torch.distributed as dist

from .. import native

DTYPE_CODE = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}


class Transport:
    """Interface: ``bind(buckets)`` returns ``(launch_ptr, wait_ptr, user_ptr)`` for the C++ core."""

    world_size: int = 1
    averages_in_flight: bool = True  # False => reducer divides by world size after the wait

    def bind(self, buckets: Sequence[torch.Tensor]):
        raise NotImplementedError

    def post_finalize(self) -> None:
        """Called once after every bucket has been waited for."""

    def broadcast_(self, tensor: torch.Tensor, src: int = 0) -> None:
        raise NotImplementedError

    def close(self) -> None:
        pass


class TorchDistTransport(Transport):
    """``dist.all_reduce(async_op=True)`` per bucket; SUM on the wire, mean applied after the wait."""

    def __init__(self, group: Optional[dist.ProcessGroup] = None):
        self.group = group
        self.world_size = dist.get_world_size(group)
        self.averages_in_flight = False
        self._buckets: List[torch.Tensor] = []
        self._works: List[Optional[dist.Work]] = []
        self._cb_launch = None
        self._cb_wait = None

    def bind(self, buckets):
        self._buckets = list(buckets)
        self._works = [None] * len(self._buckets)

        def launch(_user, b):
            self._works[b] = dist.all_reduce(self._buckets[b], op=dist.ReduceOp.SUM, group=self.group, async_op=True)

        def wait(_user, b):
            w = self._works[b]
            if w is not None:
                w.wait()
                self._works[b] = None
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

    def allreduce_(self, rank: int, tensor: torch.Tensor, average: bool = True) -> None:
        assert tensor.dtype == torch.float32 and tensor.is_contiguous() and tensor.device.type == "cpu"
        self._lib.pdt_loopback_allreduce(self._h, rank, tensor.data_ptr(), tensor.numel(), int(average))

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.pdt_loopback_destroy(self._h)
            self._h = None


class LoopbackTransport(Transport):
    def __init__(self, fabric: LoopbackFabric, rank: int):
        self.fabric = fabric
        self.rank = rank
        self.world_size = fabric.world_size
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
        self._ctx = self._lib.pdt_loopback_rank_create(self.fabric._h, self.rank, n, ptrs, numel, 1)
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


class RcclComm:
    """Own RCCL communicator + dedicated HIP stream, bootstrapped through torch's store."""

    _counter = 0

    def __init__(self, rank: int, world_size: int, device: torch.device, store=None, high_priority: bool = True):
        self._lib = native.rccl()
        self.rank, self.world_size, self.device = rank, world_size, torch.device(device)
        nbytes = self._lib.pdt_rccl_unique_id_bytes()
        key = f"pdt/rccl_uid/{RcclComm._counter}"
        RcclComm._counter += 1
        if world_size > 1:
            if store is None:
                store = dist.distributed_c10d._get_default_store()
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
        self._h = self._lib.pdt_rccl_comm_create(uid, rank, world_size, self.device.index or 0, int(high_priority))
        if not self._h:
            raise RuntimeError(f"RCCL communicator creation failed: {self._lib.pdt_rccl_last_error().decode()}")

    def _check(self, rc: int, what: str) -> None:
        if rc != 0:
            raise RuntimeError(f"{what} failed: {self._lib.pdt_rccl_last_error().decode()}")

    @property
    def stream_handle(self) -> int:
        return int(self._lib.pdt_rccl_comm_stream(self._h) or 0)

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

    def time_all_reduce(self, tensor: torch.Tensor, warmup: int = 5, iters: int = 20) -> float:
        """Mean microseconds per in-place all-reduce of ``tensor`` on the communication stream."""
        torch.cuda.synchronize(self.device)
        us = self._lib.pdt_rccl_time_allreduce(
            self._h, tensor.data_ptr(), tensor.numel(), DTYPE_CODE[tensor.dtype], warmup, iters
        )
        if us < 0:
            raise RuntimeError(self._lib.pdt_rccl_last_error().decode())
        return float(us)

    def synchronize(self) -> None:
        self._check(self._lib.pdt_rccl_stream_sync(self._h), "hipStreamSynchronize")

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._lib.pdt_rccl_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class RcclTransport(Transport):
    """Bucket all-reduce (ncclAvg) on the communicator's dedicated stream, event-ordered in C++."""

    def __init__(self, comm: RcclComm):
        self.comm = comm
        self.world_size = comm.world_size
        self._lib = native.rccl()
        self._state = None
        self._keep = None

    def bind(self, buckets):
        n = len(buckets)
        for b in buckets:
            if not b.is_cuda or b.dtype not in DTYPE_CODE:
                raise TypeError("RCCL transport needs GPU buckets of fp32/bf16/fp16")
        ptrs = (ctypes.c_void_p * n)(*[b.data_ptr() for b in buckets])
        counts = (ctypes.c_int64 * n)(*[b.numel() for b in buckets])
        dtypes = (ctypes.c_int32 * n)(*[DTYPE_CODE[b.dtype] for b in buckets])
        self._keep = (list(buckets), ptrs, counts, dtypes)
        if self._state:
            self._lib.pdt_rccl_buckets_destroy(self._state)
        stream = torch.cuda.current_stream(self.comm.device).cuda_stream
        self._state = self._lib.pdt_rccl_buckets_create(self.comm._h, n, ptrs, counts, dtypes, 1, stream)
        if not self._state:
            raise RuntimeError(self._lib.pdt_rccl_last_error().decode())
        return (
            ctypes.c_void_p(self._lib.pdt_rccl_launch_fn()),
            ctypes.c_void_p(self._lib.pdt_rccl_wait_fn()),
            ctypes.c_void_p(self._state),
        )

    def set_compute_stream(self, stream_handle: int) -> None:
        if self._state:
            self._lib.pdt_rccl_buckets_set_compute_stream(self._state, stream_handle)

    def post_finalize(self):
        if self._state and self._lib.pdt_rccl_buckets_errors(self._state):
            raise RuntimeError(f"RCCL bucket all-reduce failed: {self._lib.pdt_rccl_last_error().decode()}")

    def broadcast_(self, tensor, src=0):
        """Blocking broadcast (control-plane use only): runs on the communication stream, fenced on both sides."""
        torch.cuda.current_stream(self.comm.device).synchronize()
        self.comm.broadcast_(tensor, src=src)
        self.comm.synchronize()

    def close(self):
        if self._state:
            self._lib.pdt_rccl_buckets_destroy(self._state)
            self._state = None
