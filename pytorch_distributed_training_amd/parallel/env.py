"""Process-group bring-up and teardown (one process per GPU, ``env://`` rendezvous).

Behavioural contract taken from /root/reference/src/main.py:35-42,50-59:
``--distributed`` initialises the default process group from torchrun's
environment, picking ``nccl`` (RCCL on ROCm) when usable and ``gloo`` otherwise,
and prints ``Process group initialized - WORLD_SIZE: .., RANK: ..``.

Deliberate differences from the reference:

* the local device is ``LOCAL_RANK`` (fallback ``rank % device_count``); the
  reference computes ``device_count % rank`` (main.py:52), which divides by zero
  on rank 0;
* the device is selected *before* the process group is created so RCCL binds
  each rank to its own GPU (the reference never calls ``set_device``);
* ``--distributed --use-cpu`` is allowed and selects gloo (main.py:36-37 forbids
  it) so the plumbing can be exercised without GPUs;
* a missing ``WORLD_SIZE`` raises a readable error instead of ``KeyError``
  (main.py:38), and the group is destroyed on exit (the reference never does).
"""

from __future__ import annotations

import datetime
import os
from dataclasses import dataclass
from typing import Optional

import torch
import torch.distributed as dist

from ..utils.logging import set_rank


@dataclass
class DistEnv:
    rank: int = 0
    local_rank: int = 0
    world_size: int = 1
    backend: Optional[str] = None  # None => single process, no group
    device: torch.device = torch.device("cpu")
    owns_group: bool = False

    @property
    def distributed(self) -> bool:
        return self.backend is not None

    @property
    def is_main(self) -> bool:
        return self.rank == 0

    @property
    def device_label(self):
        """What the reference prints after ``Device - `` (an int index on GPU, ``cpu`` otherwise)."""
        return self.device.index if self.device.type == "cuda" else self.device


def select_backend(use_cpu: bool) -> str:
    """``nccl`` (= RCCL) when a GPU will be used and the backend is built in, else ``gloo``."""
    if not use_cpu and torch.cuda.is_available() and dist.is_nccl_available():
        return "nccl"
    return "gloo"


def resolve_local_rank(rank: int) -> int:
    if "LOCAL_RANK" in os.environ:
        return int(os.environ["LOCAL_RANK"])
    n = torch.cuda.device_count()
    return rank % n if n > 0 else 0


def init_distributed(distributed: bool, use_cpu: bool = False, timeout_s: float = 600.0) -> DistEnv:
    """Create the default process group (when ``distributed``) and pick this rank's device."""
    gpu_ok = torch.cuda.is_available() and not use_cpu
    if not distributed:
        device = torch.device("cuda", torch.cuda.current_device()) if gpu_ok else torch.device("cpu")
        if gpu_ok:
            torch.cuda.set_device(device)
        return DistEnv(device=device)

    if "WORLD_SIZE" not in os.environ or "RANK" not in os.environ:
        raise RuntimeError(
            "--distributed needs torchrun-style environment variables (RANK, WORLD_SIZE, "
            "MASTER_ADDR, MASTER_PORT); launch with `torchrun --nproc-per-node=N src/main.py --distributed`"
        )
    world_size = int(os.environ["WORLD_SIZE"])
    rank = int(os.environ["RANK"])
    if world_size < 1:
        raise RuntimeError(f"WORLD_SIZE must be >= 1, got {world_size}")
    local_rank = resolve_local_rank(rank)
    backend = select_backend(use_cpu)
    if backend == "nccl":
        n = torch.cuda.device_count()
        if local_rank >= n:
            raise RuntimeError(f"LOCAL_RANK={local_rank} but only {n} GPU(s) are visible")
        device = torch.device("cuda", local_rank)
        torch.cuda.set_device(device)
    else:
        device = torch.device("cpu")

    if backend == "nccl" and os.environ.get("MASTER_ADDR", "") in ("127.0.0.1", "localhost", "::1"):
        # a one-node job that meets on the loopback address: let RCCL bootstrap over the loopback interface too,
        # whatever other interfaces the box has (a container hostname that does not resolve, a NIC without a route)
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
    owns = False
    if not dist.is_initialized():
        kwargs = dict(backend=backend, timeout=datetime.timedelta(seconds=timeout_s))
        if backend == "nccl":
            kwargs["device_id"] = device  # eager communicator bound to this rank's GPU
        store = _restart_scoped_store(world_size, timeout_s)
        if store is not None:
            kwargs.update(store=store, rank=rank, world_size=world_size)
        dist.init_process_group(**kwargs)
        owns = True
    set_rank(dist.get_rank())
    return DistEnv(
        rank=dist.get_rank(),
        local_rank=local_rank,
        world_size=dist.get_world_size(),
        backend=backend,
        device=device,
        owns_group=owns,
    )


def _restart_scoped_store(world_size: int, timeout_s: float):
    """Under a launcher that restarts failed worker groups (``torchrun --max-restarts``) with a fixed
    ``--master-addr/--master-port``, the launcher's key-value store outlives the workers: the addresses the previous
    incarnation published are still in it, and a restarted rank that reads one before its peer has overwritten it
    connects to a process that no longer exists.  Scope every key of this incarnation by the restart count.
    Returns ``None`` when the process was not started by such a launcher (plain ``env://`` initialisation)."""
    if os.environ.get("TORCHELASTIC_USE_AGENT_STORE") != "True" or "TORCHELASTIC_RESTART_COUNT" not in os.environ:
        return None
    if "MASTER_ADDR" not in os.environ or "MASTER_PORT" not in os.environ:
        return None
    base = dist.TCPStore(os.environ["MASTER_ADDR"], int(os.environ["MASTER_PORT"]), world_size, is_master=False,
                         timeout=datetime.timedelta(seconds=timeout_s))
    return dist.PrefixStore(f"/pdt/attempt_{os.environ['TORCHELASTIC_RESTART_COUNT']}", base)


def barrier(env: DistEnv) -> None:
    if env.distributed and dist.is_initialized():
        if env.backend == "nccl":
            dist.barrier(device_ids=[env.device.index])
        else:
            dist.barrier()


def shutdown(env: DistEnv) -> None:
    """Barrier, then destroy the group this module created."""
    if env.distributed and dist.is_initialized():
        try:
            barrier(env)
        finally:
            if env.owns_group:
                dist.destroy_process_group()
