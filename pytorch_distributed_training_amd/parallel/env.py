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

One communicator and no hidden streams on a GPU rank.  The data plane of a GPU rank is this
package's own RCCL communicator (``DistEnv.comm``: gradient buckets, buffer and parameter
broadcasts, stop agreement, barriers, evaluation sums) on its one communication stream.
``torch.distributed`` is brought up on *gloo* there: it keeps rendezvous, rank bookkeeping and host
object exchange working without ever creating a second RCCL communicator or a stream of its own
(the box serves four hardware queues per process; step, copy and communication stream plus a
graph-capture stream are the whole budget).  ``DistEnv.backend`` still reads ``nccl`` for such a
rank - the backend the reference selects at main.py:40 - and ``control_backend`` says ``gloo``.
"""

from __future__ import annotations

import datetime
import os
from dataclasses import dataclass
from typing import Any, List, Optional, Sequence

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
    control_backend: Optional[str] = None  # what torch.distributed itself runs on (gloo, also on GPU ranks)
    comm: Any = None  # RcclComm of a GPU rank: the only communicator of the process
    store: Any = None  # the rendezvous key-value store (scoped to this incarnation of the job)

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
    """Create the default process group (when ``distributed``) and pick this rank's device.

    GPU ranks: torch's group runs on gloo (host control plane) and this package's RCCL communicator is created
    here, once, from the same store; every device collective of the process goes through it."""
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
    if not 0 <= rank < world_size:
        raise RuntimeError(f"RANK must be in [0, {world_size}), got {rank}")
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

    if os.environ.get("MASTER_ADDR", "") in ("127.0.0.1", "localhost", "::1"):
        # a one-node job that meets on the loopback address: let RCCL and gloo bootstrap over the loopback interface
        # too, whatever other interfaces the box has (a container hostname that does not resolve, a NIC without a route)
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
    owns = False
    store = None
    if not dist.is_initialized():
        store = _rendezvous_store(rank, world_size, timeout_s)
        kwargs = dict(backend="gloo", timeout=datetime.timedelta(seconds=timeout_s))
        if store is not None:
            kwargs.update(store=dist.PrefixStore("c10d", store), rank=rank, world_size=world_size)
        dist.init_process_group(**kwargs)
        owns = True
    set_rank(dist.get_rank())
    env = DistEnv(
        rank=dist.get_rank(),
        local_rank=local_rank,
        world_size=dist.get_world_size(),
        backend=backend,
        device=device,
        owns_group=owns,
        control_backend=dist.get_backend(),
        store=store,
    )
    if backend == "nccl":
        from .transport import RcclComm

        env.comm = RcclComm(env.rank, env.world_size, device, store=_comm_store(env))
    return env


def _rendezvous_store(rank: int, world_size: int, timeout_s: float):
    """The key-value store of this job, created explicitly (the group and the RCCL bootstrap both use it, and no
    private accessor of torch is needed to get at it afterwards).

    Under a launcher that restarts failed worker groups (``torchrun --max-restarts``) with a fixed
    ``--master-addr/--master-port``, the launcher's store outlives the workers: the addresses the previous
    incarnation published are still in it, and a restarted rank that reads one before its peer has overwritten it
    connects to a process that no longer exists.  Every key of this incarnation is therefore scoped by the restart
    count.  Without such a launcher rank 0 hosts the store on ``MASTER_PORT``.  Returns ``None`` when the address
    is not in the environment (torch's own ``env://`` handling then reports what is missing)."""
    if "MASTER_ADDR" not in os.environ or "MASTER_PORT" not in os.environ:
        return None
    addr, port = os.environ["MASTER_ADDR"], int(os.environ["MASTER_PORT"])
    timeout = datetime.timedelta(seconds=timeout_s)
    if os.environ.get("TORCHELASTIC_USE_AGENT_STORE") == "True":
        base = dist.TCPStore(addr, port, world_size, is_master=False, timeout=timeout)
        return dist.PrefixStore(f"/pdt/attempt_{os.environ.get('TORCHELASTIC_RESTART_COUNT', '0')}", base)
    return dist.TCPStore(addr, port, world_size, is_master=(rank == 0), timeout=timeout, wait_for_workers=False)


_comm_serial = 0


def _comm_store(env: DistEnv):
    """Store the RCCL unique id travels through: the job's own store when this module created it, else (a group
    initialised by the caller) a host broadcast through that group."""
    global _comm_serial
    _comm_serial += 1
    if env.store is not None:
        return dist.PrefixStore(f"pdt/rccl/{_comm_serial}", env.store)
    return _GroupStore()


class _GroupStore:
    """``set`` / ``get`` of one value from rank 0 through the default group's object broadcast (for process groups
    that were created outside this module, whose store is not ours to use)."""

    def __init__(self):
        self._value = None

    def set(self, _key, value):
        self._value = value
        dist.broadcast_object_list([bytes(value)], src=0)

    def get(self, _key):
        box = [None]
        dist.broadcast_object_list(box, src=0)
        return box[0]


def barrier(env: DistEnv) -> None:
    """Every rank has reached this point (host-level).  GPU ranks: a one-element all-reduce on the RCCL communicator
    behind everything already submitted to the current stream, then a wait for that stream - no other communicator
    or stream is involved."""
    if not (env.distributed and dist.is_initialized()):
        return
    if env.comm is not None:
        t = torch.zeros(1, dtype=torch.float32, device=env.device)
        env.comm.ordered_all_reduce_(t, "sum")
        torch.cuda.current_stream(env.device).synchronize()
    else:
        dist.barrier()


def all_reduce_values(env: DistEnv, values: Sequence[float], op: str = "sum") -> List[float]:
    """Host numbers reduced over ranks (``sum`` | ``max`` | ``min``) in double precision; every rank gets the result.
    Control-plane use only (stop flags, evaluation totals, benchmark timings): GPU ranks go through the RCCL
    communicator, CPU ranks through gloo."""
    vals = [float(v) for v in values]
    if not (env.distributed and dist.is_initialized()) or env.world_size == 1:
        return vals
    sign = -1.0 if op == "min" else 1.0
    kind = "sum" if op == "sum" else "max"
    if env.comm is not None:
        t = torch.tensor([sign * v for v in vals], dtype=torch.float64, device=env.device)
        env.comm.ordered_all_reduce_(t, kind)
        out = t.cpu().tolist()
    else:
        t = torch.tensor([sign * v for v in vals], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.SUM if kind == "sum" else dist.ReduceOp.MAX)
        out = t.tolist()
    return [sign * v for v in out]


def all_gather_values(env: DistEnv, value: float) -> List[float]:
    """One number from every rank, in rank order, on every rank."""
    slots = [0.0] * max(env.world_size, 1)
    slots[env.rank] = float(value)
    return all_reduce_values(env, slots, "sum")


def shutdown(env: DistEnv) -> None:
    """Barrier, close the RCCL communicator, then destroy the group this module created."""
    if env.distributed and dist.is_initialized():
        try:
            barrier(env)
        finally:
            if env.comm is not None:
                env.comm.close()
                env.comm = None
            if env.owns_group:
                dist.destroy_process_group()
