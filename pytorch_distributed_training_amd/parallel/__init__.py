"""Data parallelism: process-group bring-up, bucket transports and the gradient reducer."""

from __future__ import annotations

import torch
import torch.distributed as dist

from ..utils.logging import get_logger
from .env import DistEnv, barrier, init_distributed, select_backend, shutdown  # noqa: F401
from .reducer import BucketedDataParallel, BucketPlan, GradReducer, plan_buckets  # noqa: F401
from .transport import (  # noqa: F401
    LoopbackFabric,
    LoopbackTransport,
    RcclComm,
    RcclTransport,
    TorchDistTransport,
    Transport,
)


def make_transport(env: DistEnv, native_rccl: bool = True) -> Transport:
    """Pick the bucket transport for this process: own RCCL communicator on GPUs, gloo otherwise."""
    if not env.distributed:
        raise ValueError("no process group: data parallelism needs --distributed")
    if env.backend == "nccl" and native_rccl:
        try:
            comm = RcclComm(env.rank, env.world_size, env.device)
        except Exception as e:  # e.g. the native library is missing on this box: say so, keep the job alive
            ok = torch.tensor([0.0], device=env.device)
            get_logger("pdt.parallel").warning("own RCCL communicator unavailable (%s); using torch.distributed all-reduce", e)
        else:
            ok = torch.tensor([1.0], device=env.device)
        # every rank must pick the same transport
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if float(ok) > 0:
            return RcclTransport(comm)
    return TorchDistTransport()


def wrap_data_parallel(module, env: DistEnv, bucket_mb: float = 25.0, first_bucket_mb: float = 1.0,
                       broadcast_buffers: bool = True, native_rccl: bool = True) -> BucketedDataParallel:
    return BucketedDataParallel(
        module,
        make_transport(env, native_rccl),
        bucket_mb=bucket_mb,
        first_bucket_mb=first_bucket_mb,
        broadcast_buffers=broadcast_buffers,
    )
