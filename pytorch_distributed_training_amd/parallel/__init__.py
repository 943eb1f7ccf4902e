"""Data parallelism: process-group bring-up, bucket transports and the gradient reducer."""

from __future__ import annotations

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
        comm = RcclComm(env.rank, env.world_size, env.device)
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
