"""ctypes loaders for the in-tree native libraries.

``import torch`` must happen before any of them is loaded: the libraries link
``libamdhip64.so.7`` / ``librccl.so.1`` by soname, and the copies PyTorch has already
mapped are the ones that must be shared (one HIP runtime per process, so streams, events
and device pointers are interchangeable between PyTorch and these libraries).

A missing library raises immediately; nothing here falls back to another implementation.
"""

from __future__ import annotations

import ctypes
import os
from pathlib import Path
from typing import Optional

import torch  # noqa: F401  (maps the HIP runtime and RCCL first)

LIB_DIR = Path(__file__).resolve().parent / "_lib"

c_i32p = ctypes.POINTER(ctypes.c_int32)
c_i64p = ctypes.POINTER(ctypes.c_int64)
BUCKET_FN = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p)

_core: Optional[ctypes.CDLL] = None
_rccl: Optional[ctypes.CDLL] = None
_kernels: Optional[ctypes.CDLL] = None


class NativeLibraryError(RuntimeError):
    pass


def _load(name: str) -> ctypes.CDLL:
    path = LIB_DIR / name
    if not path.exists():
        raise NativeLibraryError(
            f"{path} is missing; build the native libraries first: `python csrc/build.py` "
            f"(or `python -c 'import __graft_entry__ as g; g.build()'`)"
        )
    try:
        return ctypes.CDLL(str(path))
    except OSError as e:  # unresolved HIP/RCCL symbols, wrong arch, ...
        raise NativeLibraryError(f"cannot load {path}: {e}") from e


def core() -> ctypes.CDLL:
    """Reducer core + loopback fabric (no GPU runtime needed)."""
    global _core
    if _core is None:
        lib = _load("libpdt_core.so")
        lib.pdt_plan_buckets.restype = ctypes.c_int
        lib.pdt_plan_buckets.argtypes = [
            c_i64p, c_i32p, ctypes.c_int, c_i32p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
            c_i32p, c_i64p, c_i64p,
        ]
        lib.pdt_reducer_create.restype = ctypes.c_void_p
        lib.pdt_reducer_create.argtypes = [
            ctypes.c_int, c_i32p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
        ]
        lib.pdt_reducer_destroy.argtypes = [ctypes.c_void_p]
        lib.pdt_reducer_destroy.restype = None
        lib.pdt_reducer_prepare.argtypes = [ctypes.c_void_p]
        lib.pdt_reducer_prepare.restype = None
        lib.pdt_reducer_mark_ready.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        lib.pdt_reducer_mark_ready.restype = ctypes.c_int
        lib.pdt_reducer_finalize.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        lib.pdt_reducer_finalize.restype = ctypes.c_int
        lib.pdt_reducer_ready_order.argtypes = [ctypes.c_void_p, c_i32p]
        lib.pdt_reducer_ready_order.restype = None
        lib.pdt_reducer_launch_count.argtypes = [ctypes.c_void_p]
        lib.pdt_reducer_launch_count.restype = ctypes.c_int64
        lib.pdt_reducer_pending_buckets.argtypes = [ctypes.c_void_p]
        lib.pdt_reducer_pending_buckets.restype = ctypes.c_int
        # loopback fabric
        lib.pdt_loopback_create.argtypes = [ctypes.c_int]
        lib.pdt_loopback_create.restype = ctypes.c_void_p
        lib.pdt_loopback_destroy.argtypes = [ctypes.c_void_p]
        lib.pdt_loopback_destroy.restype = None
        lib.pdt_loopback_rank_create.argtypes = [
            ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p), c_i64p, ctypes.c_int,
        ]
        lib.pdt_loopback_rank_create.restype = ctypes.c_void_p
        lib.pdt_loopback_rank_destroy.argtypes = [ctypes.c_void_p]
        lib.pdt_loopback_rank_destroy.restype = None
        lib.pdt_loopback_rank_collectives.argtypes = [ctypes.c_void_p]
        lib.pdt_loopback_rank_collectives.restype = ctypes.c_int64
        lib.pdt_loopback_allreduce.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int]
        lib.pdt_loopback_allreduce.restype = None
        lib.pdt_loopback_launch_fn.restype = ctypes.c_void_p
        lib.pdt_loopback_wait_fn.restype = ctypes.c_void_p
        _core = lib
    return _core


def rccl() -> ctypes.CDLL:
    """RCCL transport (needs the HIP runtime; loadable on a GPU-less box, unusable there)."""
    global _rccl
    if _rccl is None:
        lib = _load("libpdt_rccl.so")
        lib.pdt_rccl_last_error.restype = ctypes.c_char_p
        lib.pdt_rccl_unique_id_bytes.restype = ctypes.c_int
        lib.pdt_rccl_unique_id.argtypes = [ctypes.c_char_p]
        lib.pdt_rccl_unique_id.restype = ctypes.c_int
        lib.pdt_rccl_comm_create.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        lib.pdt_rccl_comm_create.restype = ctypes.c_void_p
        lib.pdt_rccl_comm_destroy.argtypes = [ctypes.c_void_p]
        lib.pdt_rccl_comm_destroy.restype = None
        lib.pdt_rccl_comm_stream.argtypes = [ctypes.c_void_p]
        lib.pdt_rccl_comm_stream.restype = ctypes.c_void_p
        lib.pdt_rccl_allreduce.argtypes = [
            ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
        ]
        lib.pdt_rccl_allreduce.restype = ctypes.c_int
        lib.pdt_rccl_broadcast.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p]
        lib.pdt_rccl_broadcast.restype = ctypes.c_int
        lib.pdt_rccl_stream_sync.argtypes = [ctypes.c_void_p]
        lib.pdt_rccl_stream_sync.restype = ctypes.c_int
        lib.pdt_rccl_time_allreduce.argtypes = [
            ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int,
        ]
        lib.pdt_rccl_time_allreduce.restype = ctypes.c_double
        lib.pdt_rccl_comm_count.argtypes = [ctypes.c_void_p]
        lib.pdt_rccl_comm_count.restype = ctypes.c_int
        lib.pdt_rccl_comm_async_error.argtypes = [ctypes.c_void_p]
        lib.pdt_rccl_comm_async_error.restype = ctypes.c_int
        lib.pdt_rccl_comm_abort.argtypes = [ctypes.c_void_p]
        lib.pdt_rccl_comm_abort.restype = ctypes.c_int
        lib.pdt_rccl_ordered.argtypes = [
            ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
        ]
        lib.pdt_rccl_ordered.restype = ctypes.c_int
        lib.pdt_rccl_buckets_create.argtypes = [
            ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p), c_i64p, c_i32p,
            ctypes.c_int,
        ]
        lib.pdt_rccl_buckets_create.restype = ctypes.c_void_p
        lib.pdt_rccl_buckets_destroy.argtypes = [ctypes.c_void_p]
        lib.pdt_rccl_buckets_destroy.restype = None
        lib.pdt_rccl_buckets_errors.argtypes = [ctypes.c_void_p]
        lib.pdt_rccl_buckets_errors.restype = ctypes.c_int
        lib.pdt_rccl_buckets_wire_bytes.argtypes = [ctypes.c_void_p]
        lib.pdt_rccl_buckets_wire_bytes.restype = ctypes.c_int64
        lib.pdt_rccl_buckets_device_count.argtypes = [ctypes.c_void_p]
        lib.pdt_rccl_buckets_device_count.restype = ctypes.c_int64
        lib.pdt_rccl_live_streams.restype = ctypes.c_int
        lib.pdt_rccl_launch_fn.restype = ctypes.c_void_p
        lib.pdt_rccl_wait_fn.restype = ctypes.c_void_p
        _rccl = lib
    return _rccl


def kernels() -> ctypes.CDLL:
    """HIP kernel library; signatures are declared next to their Python wrappers in ``ops``."""
    global _kernels
    if _kernels is None:
        # PDT_KERNELS_LIB: file name under _lib/ of an alternative build (kernel A/B experiments)
        _kernels = _load(os.environ.get("PDT_KERNELS_LIB", "libpdt_kernels.so"))
    return _kernels


def kernels_available() -> bool:
    return (LIB_DIR / "libpdt_kernels.so").exists()
