"""Data parallelism: process-group bring-up, bucket transports and the gradient reducer."""

from __future__ import annotations

import torch

from .env import (  # noqa: F401
    DistEnv,
    all_gather_values,
    all_reduce_values,
    barrier,
    init_distributed,
    select_backend,
    shutdown,
)
from .env import _comm_store
from .reducer import BucketedDataParallel, BucketPlan, GradReducer, plan_buckets  # noqa: F401
from .transport import (  # noqa: F401
    LoopbackFabric,
    LoopbackTransport,
    RcclComm,
    RcclError,
    RcclTransport,
    TorchDistTransport,
    Transport,
)


WIRE_DTYPES = {"fp32": None, "bf16": torch.bfloat16}


def make_transport(env: DistEnv, grad_dtype: str = "fp32", algorithm: str = "allreduce") -> Transport:
    """The bucket transport of this process: this package's own RCCL communicator on GPUs, gloo on the CPU.

    There is no second choice on a GPU: when the native library or the communicator cannot be brought up the job
    stops with the reason.  (Falling back per rank would also deadlock the peers waiting in ``ncclCommInitRank``.)
    ``grad_dtype``: element type of gradients on the wire (``fp32`` | ``bf16``); ``algorithm``: ``allreduce`` or
    ``rs_ag`` (reduce-scatter + all-gather, RCCL only)."""
    if not env.distributed:
        raise ValueError("no process group: data parallelism needs --distributed")
    if grad_dtype not in WIRE_DTYPES:
        raise ValueError(f"grad_dtype must be one of {sorted(WIRE_DTYPES)}")
    wire = WIRE_DTYPES[grad_dtype]
    if env.backend == "nccl":
        if env.comm is None:  # an environment that was not built by init_distributed
            env.comm = RcclComm(env.rank, env.world_size, env.device, store=_comm_store(env))
        # the process's one communicator: bucket collectives share it (and its stream) with the control collectives
        return RcclTransport(env.comm, wire_dtype=wire, algorithm=algorithm)
    if algorithm != "allreduce":
        raise ValueError("the reduce-scatter + all-gather form needs the RCCL transport (GPU ranks)")
    return TorchDistTransport(wire_dtype=wire)


def wrap_data_parallel(module, env: DistEnv, bucket_mb: float = 25.0, first_bucket_mb: float = 1.0,
                       broadcast_buffers: bool = True, grad_dtype: str = "fp32", algorithm: str = "allreduce",
                       last_bucket_mb: float = 1.0) -> BucketedDataParallel:
    return BucketedDataParallel(
        module,
        make_transport(env, grad_dtype, algorithm),
        bucket_mb=bucket_mb,
        first_bucket_mb=first_bucket_mb,
        broadcast_buffers=broadcast_buffers,
        last_bucket_mb=last_bucket_mb,
    )
