"""Rank bodies shared by the multi-process tests: the same function runs as N GPU ranks over RCCL
(``test_gpu_multi.py``, needs at least two GPUs) and as two CPU ranks over gloo (``test_dist_gloo.py``), so the
logic of the N-GPU test is exercised wherever the suite runs."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def init_rank(rank: int, world: int, port: int) -> None:
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    torch.set_num_threads(2)


def _gather_equal(t: torch.Tensor, world: int) -> bool:
    t = t.detach().cpu().contiguous()  # the process group is the host control plane (gloo), also on GPU ranks
    parts = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(parts, t)
    return all(torch.equal(parts[0], p) for p in parts)


def dp_parity_worker(rank: int, world: int, port: int, tmp: str, use_gpu: bool, variant: str = "plain") -> None:
    """ResNet-18 under the bucketed data-parallel wrapper:

    * the construction-time broadcast makes every rank equal to rank 0 (ranks start from different seeds);
    * after one backward pass every rank's gradients equal the mean over ranks of the ranks' local gradients,
      recomputed here in one process with an unwrapped copy of the model;
    * after four optimizer steps the parameters are bit-identical on every rank;
    * on GPUs the transport is this package's RCCL transport and RCCL itself reports ``world`` ranks.

    ``variant``: plain | bf16 (bf16 on the wire) | rs_ag (reduce-scatter + all-gather, GPU only).
    """
    init_rank(rank, world, port)
    from pytorch_distributed_training_amd import ops
    from pytorch_distributed_training_amd.models import build_model
    from pytorch_distributed_training_amd.optim import build_optimizer
    from pytorch_distributed_training_amd.parallel import (RcclTransport, TorchDistTransport, init_distributed, shutdown,
                                                          wrap_data_parallel)

    env = init_distributed(True, use_cpu=not use_gpu, timeout_s=180)
    assert env.world_size == world and env.rank == rank
    dev = env.device
    per_rank = 4

    def make(seed):
        torch.manual_seed(seed)
        m = build_model("resnet18", num_classes=10, image_size=32).to(dev)
        if use_gpu:
            m = m.to(memory_format=torch.channels_last)
        return ops.prepare_module(m)

    raw = make(seed=100 + rank)
    model = wrap_data_parallel(raw, env, bucket_mb=4.0, first_bucket_mb=0.5,
                               grad_dtype="bf16" if variant == "bf16" else "fp32",
                               algorithm="rs_ag" if variant == "rs_ag" else "allreduce")
    if use_gpu:
        assert isinstance(model.transport, RcclTransport)
        assert model.transport.comm.rccl_ranks == world
    else:
        assert isinstance(model.transport, TorchDistTransport)
    flat = torch.cat([p.detach().reshape(-1) for p in raw.parameters()])
    assert _gather_equal(flat, world), "construction-time broadcast left the ranks different"

    g = torch.Generator().manual_seed(5)
    xs = torch.randn(world * per_rank, 3, 32, 32, generator=g)
    ys = torch.randint(0, 10, (world * per_rank,), generator=g)

    def batch(r):
        x = xs[r * per_rank:(r + 1) * per_rank].to(dev)
        if use_gpu:
            x = ops.to_device_layout(x, torch.bfloat16)
        return x, ys[r * per_rank:(r + 1) * per_rank].to(dev)

    # --- gradient parity against the mean of the ranks' local gradients
    ref = make(seed=0)
    ref.load_state_dict(raw.state_dict())
    model.train()
    ref.train()
    model.reducer.zero_grad()
    x, y = batch(rank)
    ops.softmax_cross_entropy(model(x), y).backward()
    if use_gpu:
        torch.cuda.synchronize()
    got = [p.grad.detach().float().clone() for p in raw.parameters()]
    acc = None
    for r in range(world):
        for p in ref.parameters():
            p.grad = None
        x, y = batch(r)
        ops.softmax_cross_entropy(ref(x), y).backward()
        gs = [p.grad.detach().float().clone() for p in ref.parameters()]
        acc = gs if acc is None else [a + b for a, b in zip(acc, gs)]
    loose = use_gpu or variant == "bf16"  # bf16 activations / atomics / a bf16 wire: compare at bf16 resolution
    for (name, _), a, e in zip(raw.named_parameters(), got, acc):
        e = e / world
        scale = float(e.abs().max()) + 1e-12
        err = float((a - e).abs().max()) / scale
        assert err < (3e-2 if loose else 1e-5), f"{name}: reduced gradient is off by {err:.3e} of its scale"
    assert _gather_equal(torch.cat([t.reshape(-1) for t in got]), world), "ranks hold different reduced gradients"

    # --- four optimizer steps keep the replicas bit-identical
    opt = build_optimizer([p for p in raw.parameters() if p.requires_grad], "sgd", 0.05, 1e-4, 0.9)
    model.reducer.zero_grad()
    for i in range(4):
        x, y = batch((rank + i) % world)
        ops.softmax_cross_entropy(model(x), y).backward()
        opt.step()
        model.reducer.zero_grad()
    flat = torch.cat([p.detach().reshape(-1) for p in raw.parameters()])
    assert _gather_equal(flat, world), "parameters diverged across ranks"
    assert len(model.reducer.buckets) > 2
    with open(os.path.join(tmp, f"ok_{variant}_{rank}"), "w") as f:
        f.write(model.reducer.bucket_summary())
    model.reducer.close()
    shutdown(env)


def graphed_trainer_worker(rank: int, world: int, port: int, tmp: str) -> None:
    """The trainer with ``--hip-graph --distributed`` on GPU ranks: replays of the recorded step execute the
    bucket collectives (device-side counter) and the replicas stay bit-identical."""
    init_rank(rank, world, port)
    os.environ["PDT_RCCL_COUNT"] = "1"
    from pytorch_distributed_training_amd.train import TrainConfig, Trainer

    cfg = TrainConfig(distributed=True, model="resnet18", dataset="synthetic", dataset_length=16 * 8 * world,
                      batch_size=8, num_workers=0, learning_rate=0.05, optimizer="sgd", progress=False,
                      device_data=True, hip_graph=True, graph_after=3, steps=10, bucket_mb=25.0)
    tr = Trainer(cfg)
    transport = tr.model.transport
    res = tr.fit()
    assert res.steps == 10 and tr._graph is not None and tr._graph.replays == 7
    n_buckets = len(tr.model.reducer.buckets)
    executed = transport.device_collectives
    # 3 eager steps (the first one on the initial bucket plan) + 7 replays, every one with all its collectives
    assert executed >= 9 * n_buckets, (executed, n_buckets)
    flat = torch.cat([p.detach().reshape(-1) for p in tr.raw_model.parameters()])
    assert _gather_equal(flat, world), "replicas diverged under the recorded step"
    with open(os.path.join(tmp, f"ok_graph_{rank}"), "w") as f:
        f.write(str(executed))
    tr.close()
