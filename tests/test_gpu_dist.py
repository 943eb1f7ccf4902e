"""The RCCL transport and the reducer on a real GPU.  One GPU is available to the tests, so the process
group has a single rank; the code path (own communicator, dedicated stream, event ordering, bucket launch
from autograd hooks, finalize before the optimizer) is the one N-rank jobs take."""
import os
import subprocess
import sys

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


def test_wrapped_state_dict_has_no_prefix_and_loads_into_a_plain_model(one_rank_group):
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


def _resnet18_dp(env, transport, seed=1):
    from pytorch_distributed_training_amd import ops
    from pytorch_distributed_training_amd.models import build_model
    from pytorch_distributed_training_amd.optim import FusedOptimizer
    from pytorch_distributed_training_amd.parallel import BucketedDataParallel

    torch.manual_seed(seed)
    m = build_model("resnet18", num_classes=10).to(env.device).to(memory_format=torch.channels_last)
    ops.prepare_module(m)
    model = BucketedDataParallel(m, transport, bucket_mb=4.0, first_bucket_mb=0.5, last_bucket_mb=0.25) if transport else m
    opt = FusedOptimizer(m.parameters(), mode="sgd", lr=0.05, momentum=0.9)
    return m, model, opt


@pytest.mark.parametrize("wire,algorithm", [(None, "allreduce"), (torch.bfloat16, "allreduce"), (None, "rs_ag")])
def test_recorded_step_executes_the_bucket_collectives_on_every_replay(one_rank_group, wire, algorithm):
    """Reducer inside a hipGraph: the collectives (kept at world size 1) are nodes of the recorded step - a device
    counter advances by one per bucket per replay - and the recorded training matches the eager one."""
    from pytorch_distributed_training_amd import ops
    from pytorch_distributed_training_amd.parallel import RcclComm, RcclTransport
    from pytorch_distributed_training_amd.train.graph import GraphedStep

    env = one_rank_group
    dev = env.device
    torch.manual_seed(0)
    x = ops.to_device_layout(torch.randn(8, 3, 64, 64, device=dev))
    y = torch.randint(0, 10, (8,), device=dev)

    def run(graphed):
        comm = RcclComm(env.rank, env.world_size, dev)
        tr = RcclTransport(comm, wire_dtype=wire, algorithm=algorithm, keep_one_rank=True, count_on_device=True)
        m, model, opt = _resnet18_dp(env, tr)

        def body(a, b):
            loss = ops.softmax_cross_entropy(model(a), b)
            loss.backward()
            opt.step()
            model.reducer.zero_grad()
            return loss.detach()

        model.reducer.zero_grad()
        losses = [float(body(x, y)) for _ in range(3)]
        nb = len(model.reducer.buckets)
        if graphed:
            c0 = tr.device_collectives
            g = GraphedStep(body, opt, (x, y))
            c1 = tr.device_collectives
            assert c1 == c0, "capturing must record, not execute"
            for k in range(4):
                losses.append(float(g(x, y)))
                assert tr.device_collectives == c1 + (k + 1) * nb, "a replay did not run every bucket collective"
        else:
            opt.set_capturable(True)
            losses += [float(body(x, y)) for _ in range(4)]
            assert tr.device_collectives >= 6 * nb
        params = [p.detach().clone() for p in m.parameters()]
        model.reducer.close()
        comm.close()
        return losses, params

    la, pa = run(True)
    lb, pb = run(False)
    assert la[-1] < la[0]
    for a, b in zip(la, lb):
        assert abs(a - b) < 5e-2 * max(1.0, abs(b)), (la, lb)


def test_collectives_follow_the_stream_the_step_runs_on(one_rank_group):
    """A step on a side stream: the bucket collective must wait for the gradient kernels of *that* stream.  The
    side stream is held back by a long spin kernel, and the wire is bf16 (the unpack kernel rewrites the gradients
    rounded to bf16): if the collective were ordered against any other stream it would run before the gradients
    exist, and the gradients found afterwards would not be bf16-representable."""
    from pytorch_distributed_training_amd import ops
    from pytorch_distributed_training_amd.parallel import RcclComm, RcclTransport

    env = one_rank_group
    dev = env.device
    comm = RcclComm(env.rank, env.world_size, dev)
    tr = RcclTransport(comm, wire_dtype=torch.bfloat16, keep_one_rank=True, count_on_device=True)
    m, model, _ = _resnet18_dp(env, tr)
    x = ops.to_device_layout(torch.randn(8, 3, 64, 64, device=dev))
    y = torch.randint(0, 10, (8,), device=dev)
    model.reducer.zero_grad()
    ops.softmax_cross_entropy(model(x), y).backward()  # first pass: buckets are re-planned after it
    model.reducer.zero_grad()
    torch.cuda.synchronize()
    side = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(side):
        torch.cuda._sleep(200_000_000)  # ~0.1 s before anything else on this stream runs
        ops.softmax_cross_entropy(model(x), y).backward()
    side.synchronize()
    comm.synchronize()
    torch.cuda.synchronize()
    assert tr.device_collectives == len(model.reducer.buckets)  # the counter restarted with the re-planned buckets
    for name, p in m.named_parameters():
        g = p.grad
        assert float(g.abs().max()) > 0, name
        assert torch.equal(g, g.to(torch.bfloat16).to(torch.float32)), f"{name}: gradient did not pass through the wire"
    model.reducer.close()
    comm.close()


def test_ordered_collectives_and_failure_probes(one_rank_group):
    from pytorch_distributed_training_amd.parallel import RcclComm

    env = one_rank_group
    comm = RcclComm(env.rank, env.world_size, env.device)
    assert comm.rccl_ranks == 1
    t = torch.full((16,), 3.0, device=env.device)
    comm.ordered_all_reduce_(t, "sum")
    comm.ordered_all_reduce_(t, "max")
    comm.ordered_broadcast_(t, 0)
    assert torch.equal(t.cpu(), torch.full((16,), 3.0))  # the copy runs on the current stream, after the fence
    assert comm.async_error() is None
    comm.abort()
    assert comm.async_error() is not None
    comm.close()


def test_bench_py_prints_one_json_line_with_sync_proofs():
    """bench.py as the driver starts it on one GPU (small batch so it takes seconds)."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
                          "--batch-size", "32"], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 1 and rec["config"]["model"] == "resnet50" and rec["value"] > 0
    ex = rec["extras"]
    assert ex["transport"] == "RcclTransport" and ex["rccl_ranks"] == 1 and ex["params_in_sync"] is True
    assert ex["rank_sum"]["got"] == ex["rank_sum"]["expected"] == 1.0 and "error" not in ex


@pytest.mark.parametrize("main", ["null", "pool"])
def test_data_parallel_step_with_a_prefetch_stream_and_host_run_ahead_makes_progress(main, free_port):
    """Step stream + copy stream + communication stream, host several steps ahead: with a raised-priority
    communication stream and one more stream than hardware queues this constellation stopped the device (the waiting
    priority queue starved the queues that produce what it waits for); the communicator's stream has ordinary priority
    and the process has no other communicator."""
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dp_stream_worker.py")
    env = dict(os.environ, PORT=str(free_port), PROBE_MODEL="resnet50", PROBE_BATCH="256", PROBE_SIZE="224",
               PROBE_STEPS="20", PROBE_LIMIT="60", PROBE_MAIN=main)
    env.pop("PDT_COMM_PRIORITY", None)
    r = subprocess.run([sys.executable, "-u", worker], env=env, capture_output=True, text=True, timeout=120)
    assert "DONE" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_trainer_constellation_stays_within_the_hardware_queue_budget(free_port, tmp_path):
    """The full trainer: host-fed input pipeline (copy stream), data parallelism (communication stream), the step
    recorded into a hipGraph (capture stream) - the process must own at most as many streams as it is served hardware
    queues (4), torch.distributed must not have brought up a communicator of its own, and the run must finish."""
    code = r"""
import json, os, sys
sys.path.insert(0, os.environ["PDT_ROOT"])
import torch, torch.distributed as dist
from pytorch_distributed_training_amd.train import TrainConfig, Trainer
from pytorch_distributed_training_amd.utils.streams import stream_census
cfg = TrainConfig(distributed=True, model="resnet18", dataset="synthetic", dataset_length=512, batch_size=32, num_workers=2,
                  learning_rate=0.01, optimizer="sgd", progress=False, hip_graph=True, graph_after=3, steps=12, evaluate=True)
tr = Trainer(cfg)
res = tr.fit()
c = stream_census()
print("CENSUS " + json.dumps(dict(c, backend=dist.get_backend(), control=tr.env.control_backend, steps=res.steps,
                                  replays=tr._graph.replays if tr._graph else 0, comm=tr.env.comm is not None)), flush=True)
tr.close()
"""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               PDT_ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    env.pop("GPU_MAX_HW_QUEUES", None)
    r = subprocess.run([sys.executable, "-u", "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    import json

    line = [ln for ln in r.stdout.splitlines() if ln.startswith("CENSUS ")][-1]
    c = json.loads(line[len("CENSUS "):])
    assert c["backend"] == "gloo" and c["control"] == "gloo" and c["comm"] is True and c["torch_nccl"] == 0
    assert c["native"] == 1, c  # one communicator, one communication stream
    assert c["steps"] == 12 and c["replays"] > 0
    assert c["total"] <= 4 and c["budget"] == 4, c


def test_bucket_autotune_runs_end_to_end_on_the_rccl_communicator(one_rank_group):
    """The whole --bucket-mb auto procedure forced in a one-rank group: timing marks behind the gradients of an ordinary
    step, probe collectives on the process's RCCL communicator, the decision, and the buckets re-planned under a live
    fused optimizer (whose launch plan holds gradient addresses).  In deterministic mode the run must end bit-identical
    to one that never re-planned: the tuner changes where gradients live, never what they are."""
    from pytorch_distributed_training_amd import ops
    from pytorch_distributed_training_amd.models import build_model
    from pytorch_distributed_training_amd.optim import build_optimizer
    from pytorch_distributed_training_amd.parallel import wrap_data_parallel
    from pytorch_distributed_training_amd.parallel.autotune import BucketAutoTuner

    env = one_rank_group
    dev = env.device
    x = ops.to_device_layout(torch.randn(16, 3, 64, 64, device=dev))
    y = torch.randint(0, 10, (16,), device=dev)

    def run(tune: bool):
        torch.manual_seed(3)
        m = build_model("resnet18", num_classes=10).to(dev).to(memory_format=torch.channels_last)
        ops.prepare_module(m)
        model = wrap_data_parallel(m, env, bucket_mb=25.0, first_bucket_mb=1.0)
        opt = build_optimizer(list(m.parameters()), "sgd", 0.05, 1e-4, 0.9)
        model.reducer.zero_grad()
        decision = None
        for i in range(5):
            tuner = BucketAutoTuner(model, env, force=True) if (tune and i == 1) else None
            if tuner is not None:
                tuner.arm()
            ops.softmax_cross_entropy(model(x), y).backward()
            opt.step()
            model.reducer.zero_grad()
            if tuner is not None:
                torch.cuda.synchronize()
                decision = tuner.finish()
                model.reducer.zero_grad()
        torch.cuda.synchronize()
        flat = torch.cat([p.detach().reshape(-1) for p in m.parameters()]).clone()
        nb = len(model.reducer.buckets)
        model.reducer.close()
        return flat, decision, nb

    ops.set_deterministic(True)
    try:
        a, _, nb_a = run(False)
        b, d, nb_b = run(True)
    finally:
        ops.set_deterministic(False)
    assert d is not None and d["bucket_mb"] > 0 and d["backward_ms"] > 0 and len(d["probe_us"]) == 5
    assert nb_b == d["buckets"]
    assert torch.equal(a, b)


@pytest.mark.parametrize("variant", ["plain", "bf16", "rs_ag"])
def test_rank_body_of_the_multi_gpu_parity_test_as_one_gpu_rank(tmp_path, free_port, variant):
    """`test_gpu_multi.py` needs two GPUs; its rank body runs here as a one-rank RCCL job (own communicator, bucket
    collectives of world size 1, the bf16 wire's pack / unpack kernels, reduce-scatter + all-gather in place), so the
    code the N-GPU run executes is known to be current on hardware."""
    import torch.multiprocessing as mp

    from dist_workers import dp_parity_worker

    mp.spawn(dp_parity_worker, args=(1, free_port, str(tmp_path), True, variant), nprocs=1, join=True)
    assert (tmp_path / f"ok_{variant}_0").exists()


def test_rank_body_of_the_multi_gpu_recorded_step_test_as_one_gpu_rank(tmp_path, free_port):
    import torch.multiprocessing as mp

    from dist_workers import graphed_trainer_worker

    mp.spawn(graphed_trainer_worker, args=(1, free_port, str(tmp_path)), nprocs=1, join=True)
    assert (tmp_path / "ok_graph_0").exists()
