"""Bucket auto-tuning (``--bucket-mb auto``): the decision model on synthetic timings, and the whole procedure on two
gloo ranks (every rank must end with the same bucket plan; a single rank has nothing to tune)."""
import os
import sys

import torch
import torch.multiprocessing as mp

from pytorch_distributed_training_amd.parallel.autotune import (CommCost, choose_capacities, simulate_exposed_us)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MiB = 1 << 20


def test_comm_cost_interpolates_and_is_monotone():
    c = CommCost.fit([MiB, 4 * MiB, 16 * MiB], [100.0, 90.0, 400.0])  # a noisy dip is smoothed away
    assert c(MiB) == 100.0 and c(4 * MiB) == 100.0 and c(16 * MiB) == 400.0
    assert c(10 * MiB) == 250.0
    assert c(32 * MiB) == 800.0  # extrapolated with the last slope
    assert 50.0 <= c(1024) <= 100.0  # below the smallest probe: latency stays
    xs = [2 ** k for k in range(8, 27)]
    ys = [c(x) for x in xs]
    assert all(b >= a for a, b in zip(ys, ys[1:]))


def test_simulated_exposed_time():
    cost = lambda n: 10.0 + n / 1000.0  # noqa: E731
    # two buckets ready at 100 and 200 us, backward ends at 200: the second one (1000 bytes -> 11 us) is exposed
    assert simulate_exposed_us([1000, 1000], [100.0, 200.0], 200.0, cost) == 11.0
    # the first bucket still running when the second becomes ready: they queue
    assert simulate_exposed_us([500_000, 1000], [100.0, 200.0], 200.0, cost) == 100.0 + 510.0 + 11.0 - 200.0


def _uniform_model(n_params=40, mb_each=8.0):
    numels = [int(mb_each * MiB / 4)] * n_params
    esize = [4] * n_params
    order = list(range(n_params - 1, -1, -1))
    return numels, esize, order


def test_latency_bound_links_choose_large_buckets_and_bandwidth_bound_links_a_small_final_bucket():
    numels, esize, order = _uniform_model()
    n = len(numels)
    backward_us = 100_000.0
    ready = [0.0] * n
    for k, i in enumerate(order):
        ready[i] = backward_us * (k + 1) / n
    # (a) latency-dominated collective: 5 ms fixed + 100 GB/s.  Many small buckets cannot keep up.
    slow_start = lambda nbytes: 5000.0 + nbytes / 1e5  # noqa: E731
    a = choose_capacities(numels, esize, order, ready, backward_us, slow_start)
    # (b) no latency, modest bandwidth: the exposed tail is the last bucket, so it should be as small as allowed
    thin = lambda nbytes: nbytes / 2e4  # 20 GB/s  # noqa: E731
    b = choose_capacities(numels, esize, order, ready, backward_us, thin)
    assert a["bucket_mb"] >= 50.0, a
    assert b["exposed_us"] <= a["exposed_us"] or b["bucket_mb"] <= a["bucket_mb"]
    # with 8 MiB parameters a bucket cannot be smaller than one parameter: the final bucket is one parameter
    by_key = {(r["bucket_mb"], r["last_bucket_mb"]): r for r in b["table"]}
    assert by_key[(200.0, 0.0)]["exposed_us"] > by_key[(8.0, 0.0)]["exposed_us"]
    # monotone in the link speed: a faster link never makes the modelled exposed time of a fixed plan worse
    fast = lambda nbytes: nbytes / 2e5  # noqa: E731
    c = choose_capacities(numels, esize, order, ready, backward_us, fast)
    for rb, rc in zip(b["table"], c["table"]):
        assert rc["exposed_us"] <= rb["exposed_us"] + 1e-6
    # ties are resolved towards fewer buckets: with a free interconnect everything fits any plan
    free = choose_capacities(numels, esize, order, ready, backward_us, lambda nbytes: 0.0)
    assert free["exposed_us"] == 0.0 and free["buckets"] == min(r["buckets"] for r in free["table"])


def test_small_late_gradients_get_their_own_final_bucket():
    # ResNet-like: big tensors ready early, a few tiny ones (the stem) ready at the very end
    numels = [6_000_000] * 8 + [2_000, 500, 100]
    esize = [4] * len(numels)
    order = list(range(len(numels)))
    backward_us = 50_000.0
    ready = [backward_us * 0.1 * (i + 1) for i in range(8)] + [49_000.0, 49_500.0, 50_000.0]
    cost = lambda nbytes: 30.0 + nbytes / 1e5  # noqa: E731
    d = choose_capacities(numels, esize, order, ready, backward_us, cost)
    by_key = {(r["bucket_mb"], r["last_bucket_mb"]): r for r in d["table"]}
    # without a separate final bucket the stem rides in a 24 MB bucket that can only start when backward ends
    assert by_key[(25.0, 0.0)]["exposed_us"] > 200.0
    assert d["exposed_us"] < 100.0 and d["last_bucket_mb"] > 0.0


def _tune_worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    sys.path.insert(0, ROOT)
    torch.set_num_threads(2)
    from pytorch_distributed_training_amd.train import TrainConfig, Trainer

    cfg = TrainConfig(distributed=True, use_cpu=True, dataset="synthetic", dataset_length=64, batch_size=4, num_workers=0,
                      learning_rate=1e-3, progress=False, steps=4, bucket_mb="auto", seed=rank)  # different seeds per rank
    tr = Trainer(cfg)
    tr.fit()
    d = tr.bucket_decision
    assert d is not None and d["bucket_mb"] > 0 and "probe_us" in d and d["backward_ms"] > 0
    plan = (d["bucket_mb"], d["last_bucket_mb"], len(tr.model.reducer.buckets),
            [int(b.numel()) for b in tr.model.reducer.buckets])
    flat = torch.cat([p.detach().reshape(-1) for p in tr.raw_model.parameters()])
    torch.save({"plan": plan, "params": flat}, os.path.join(tmp, f"tune{rank}.pt"))
    tr.close()


def test_two_gloo_ranks_end_with_the_same_plan_and_stay_in_sync(tmp_path, free_port):
    mp.start_processes(_tune_worker, args=(2, free_port, str(tmp_path)), nprocs=2, join=True, start_method="spawn")
    a = torch.load(tmp_path / "tune0.pt", weights_only=True)
    b = torch.load(tmp_path / "tune1.pt", weights_only=True)
    assert a["plan"] == b["plan"]
    assert torch.equal(a["params"], b["params"])  # the tuned run is still a synchronised data-parallel job


def test_single_rank_has_nothing_to_tune():
    from pytorch_distributed_training_amd.train import TrainConfig, Trainer

    tr = Trainer(TrainConfig(use_cpu=True, dataset="synthetic", dataset_length=16, batch_size=4, num_workers=0,
                             progress=False, bucket_mb="auto"))
    tr.fit()
    assert tr.bucket_decision is None
