"""The RCCL transport and the reducer on a real GPU.  One GPU is available to the tests, so the process
group has a single rank; the code path (own communicator, dedicated stream, event ordering, bucket launch
from autograd hooks, finalize before the optimizer) is the one N-rank jobs take."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture
def one_rank_group(free_port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    from pytorch_distributed_training_amd.parallel import init_distributed, shutdown

    env = init_distributed(True, use_cpu=False, timeout_s=120)
    assert env.backend == "nccl" and env.world_size == 1
    yield env
    shutdown(env)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        os.environ.pop(k, None)


def test_rccl_comm_allreduce_broadcast_and_timing(one_rank_group):
    from pytorch_distributed_training_amd.parallel import RcclComm

    env = one_rank_group
    comm = RcclComm(env.rank, env.world_size, env.device)
    assert comm.stream_handle != 0
    t = torch.arange(1 << 20, dtype=torch.float32, device=env.device)
    ref = t.clone()
    torch.cuda.synchronize()
    comm.all_reduce_(t, average=True)
    comm.synchronize()
    assert torch.equal(t, ref)
    b = torch.randn(1000, device=env.device).to(torch.bfloat16)
    rb = b.clone()
    comm.broadcast_(b, src=0)
    comm.synchronize()
    assert torch.equal(b, rb)
    us = comm.time_all_reduce(t, warmup=2, iters=5)
    assert us > 0
    comm.close()


def test_reducer_over_rccl_trains_resnet18_like_the_plain_path(one_rank_group):
    from pytorch_distributed_training_amd import ops
    from pytorch_distributed_training_amd.models import build_model
    from pytorch_distributed_training_amd.optim import FusedOptimizer
    from pytorch_distributed_training_amd.parallel import RcclTransport, wrap_data_parallel

    env = one_rank_group
    torch.manual_seed(0)
    dev = env.device
    x = ops.to_device_layout(torch.randn(8, 3, 64, 64, device=dev))
    y = torch.randint(0, 10, (8,), device=dev)

    def run(wrap):
        torch.manual_seed(1)
        m = build_model("resnet18", num_classes=10).to(dev).to(memory_format=torch.channels_last)
        ops.prepare_module(m)
        model = wrap_data_parallel(m, env, bucket_mb=4.0, first_bucket_mb=0.5) if wrap else m
        opt = FusedOptimizer(m.parameters(), mode="sgd", lr=0.05, momentum=0.9)
        losses = []
        for _ in range(4):
            loss = ops.softmax_cross_entropy(model(x), y)
            loss.backward()
            opt.step()
            model.reducer.zero_grad() if wrap else opt.zero_grad()
            losses.append(float(loss.detach()))
        if wrap:
            assert isinstance(model.transport, RcclTransport)
            # the launch counter restarts when the buckets are rebuilt after the first iteration
            assert len(model.reducer.buckets) > 2 and model.reducer.launch_count >= 3 * len(model.reducer.buckets)
            assert all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(model.reducer.params, model.reducer.views))
            assert m.bn1.running_mean.data_ptr() >= model._flat_buffers[0].data_ptr()  # buffers live in the flat arena
            model.reducer.close()
        return losses, [p.detach().clone() for p in m.parameters()]

    la, pa = run(True)
    lb, pb = run(False)
    assert la[-1] < la[0]
    for a, b in zip(la, lb):
        assert abs(a - b) < 5e-2 * max(1.0, abs(b))  # atomics make the two runs differ in the last bits only


def test_bench_entry_point_runs_under_torchrun_env(one_rank_group):
    """`--distributed`-style wrap with WORLD_SIZE=1: state_dict has no prefix, checkpoint is loadable."""
    from pytorch_distributed_training_amd.models import build_model
    from pytorch_distributed_training_amd.parallel import wrap_data_parallel
    from pytorch_distributed_training_amd.train import portable_state_dict

    env = one_rank_group
    m = build_model("resnet18", num_classes=10).to(env.device).to(memory_format=torch.channels_last)
    w = wrap_data_parallel(m, env)
    sd = portable_state_dict(w)
    assert "conv1.weight" in sd and sd["conv1.weight"].is_contiguous() and not any(k.startswith("module.") for k in sd)
    build_model("resnet18", num_classes=10).load_state_dict(sd)
    w.reducer.close()
