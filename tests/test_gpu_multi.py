"""Data parallelism over RCCL with real ranks: one process per GPU, ``min(device_count, 8)`` of them.  Skipped on
a box with fewer than two GPUs (the same rank bodies run over gloo on the CPU in ``test_dist_gloo.py``)."""
import pytest
import torch
import torch.multiprocessing as mp

from dist_workers import dp_parity_worker, graphed_trainer_worker

pytestmark = pytest.mark.gpu


def _world() -> int:
    return min(torch.cuda.device_count(), 8)


needs_two = pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs at least two GPUs")


@needs_two
@pytest.mark.parametrize("variant", ["plain", "bf16", "rs_ag"])
def test_resnet18_data_parallel_over_rccl(tmp_path, free_port, variant):
    world = _world()
    mp.spawn(dp_parity_worker, args=(world, free_port, str(tmp_path), True, variant), nprocs=world, join=True)
    assert all((tmp_path / f"ok_{variant}_{r}").exists() for r in range(world))


@needs_two
def test_recorded_step_with_data_parallel_over_rccl(tmp_path, free_port):
    world = _world()
    mp.spawn(graphed_trainer_worker, args=(world, free_port, str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f"ok_graph_{r}").exists() for r in range(world))
