import os
import time

import numpy as np
import pytest
import torch

from pytorch_distributed_training_amd.data import CIFAR10, DeviceSyntheticLoader, SyntheticImageDataset, build_dataset
from pytorch_distributed_training_amd.utils.meter import AverageMeter, ThroughputMeter
from pytorch_distributed_training_amd.utils.timers import ChromeTrace, StepTimer
from pytorch_distributed_training_amd.utils.watchdog import GracefulStop, StepWatchdog, maybe_inject_fault


def test_throughput_meter_excludes_warmup_and_counts_global_samples():
    t = [0.0]
    m = ThroughputMeter(world_size=4, warmup_steps=2, clock=lambda: t[0])
    m.start()
    for _ in range(2):
        t[0] += 5.0  # slow warm-up steps
        m.step(8)
    for _ in range(10):
        t[0] += 0.5
        m.step(8)
    m.stop()
    assert m.timed_steps == 10 and m.timed_samples == 10 * 8 * 4
    assert m.elapsed == pytest.approx(5.0)
    assert m.samples_per_sec == pytest.approx(64.0)
    assert m.ms_per_step == pytest.approx(500.0)
    assert ThroughputMeter.scaling_efficiency(720.0, 8, 100.0) == pytest.approx(0.9)


def test_average_meter():
    a = AverageMeter()
    a.update(1.0, 2)
    a.update(4.0, 1)
    assert a.avg == pytest.approx(2.0) and a.last == 4.0


def test_step_timer_and_trace(tmp_path):
    st = StepTimer(torch.device("cpu"))
    with st.phase("fwd"):
        time.sleep(0.01)
    assert st.summary()["fwd"]["count"] == 1 and "fwd" in st.format()
    tr = ChromeTrace()
    with tr.span("x", step=1):
        pass
    tr.instant("mark")
    p = tr.dump(str(tmp_path / "t" / "trace.json"))
    assert os.path.getsize(p) > 0


def test_watchdog_fires_and_stays_quiet_when_beating():
    fired = []
    w = StepWatchdog(0.05, on_timeout=lambda: fired.append(1), poll_s=0.01).start()
    time.sleep(0.3)
    w.stop()
    assert fired and w.fired
    fired2 = []
    w2 = StepWatchdog(0.2, on_timeout=lambda: fired2.append(1), poll_s=0.01).start()
    for i in range(10):
        time.sleep(0.02)
        w2.beat(i)
    w2.stop()
    assert not fired2


def test_graceful_stop_records_signal():
    import signal

    g = GracefulStop(signals=(signal.SIGUSR1,)).install()
    os.kill(os.getpid(), signal.SIGUSR1)
    time.sleep(0.05)
    g.uninstall()
    assert g.requested and g.signum == signal.SIGUSR1


def test_fault_injection_raise(monkeypatch):
    monkeypatch.setenv("PDT_FAULT_INJECT", "1:3:raise")
    maybe_inject_fault(0, 3)
    maybe_inject_fault(1, 2)
    with pytest.raises(RuntimeError):
        maybe_inject_fault(1, 3)


def test_cifar10_binary_reader(tmp_path):
    d = tmp_path / "cifar-10-batches-bin"
    d.mkdir()
    rng = np.random.default_rng(0)
    rec = rng.integers(0, 256, size=(20, 3073), dtype=np.uint8)
    rec[:, 0] = rng.integers(0, 10, size=20)
    rec.tofile(d / "test_batch.bin")
    ds = CIFAR10(str(tmp_path), train=False)
    assert len(ds) == 20 and len(ds.classes) == 10
    x, y = ds[3]
    assert x.shape == (3, 32, 32) and x.dtype == torch.float32 and 0 <= float(x.min()) and float(x.max()) <= 1
    assert y == int(rec[3, 0])
    assert torch.allclose(x, torch.from_numpy(rec[3, 1:].reshape(3, 32, 32).astype(np.float32) / 255))
    ds2, desc = build_dataset("auto", str(tmp_path))
    assert isinstance(ds2, CIFAR10) and "CIFAR-10" in desc


def test_dataset_fallback_and_errors(tmp_path):
    ds, desc = build_dataset("auto", str(tmp_path / "nothing"))
    assert isinstance(ds, SyntheticImageDataset) and "synthetic" in desc and len(ds) == 10000
    with pytest.raises(FileNotFoundError):
        build_dataset("cifar10", str(tmp_path / "nothing"))
    lm, _ = build_dataset("synthetic", "", model_kind="lm", seq_len=16, vocab_size=100, length=8)
    a, b = lm[0]
    assert a.shape == (16,) and torch.equal(a[1:], b[:-1])


def test_device_synthetic_loader_cycles():
    ld = DeviceSyntheticLoader(lambda i: (torch.full((2,), i), torch.zeros(2)), steps=5, n_resident=2)
    assert len(ld) == 5
    assert [int(x[0]) for x, _ in ld] == [0, 1, 0, 1, 0]


def test_batch_loader_delivers_vectorised_uint8_batches_in_order(tmp_path):
    import numpy as np

    from pytorch_distributed_training_amd.data import BatchLoader, build_loader

    ds = SyntheticImageDataset(length=37, image_size=8, num_classes=5, seed=1)
    loader, sampler = build_loader(ds, batch_size=8, num_workers=3, rank=1, world_size=2, shuffle=True, seed=4)
    assert isinstance(loader, BatchLoader) and len(loader) == 3  # ceil(19 / 8): rank 1 holds 19 of the padded 38
    sampler.set_epoch(2)
    want = list(iter(sampler))
    for _pass in range(2):  # buffers are recycled between passes
        sampler.set_epoch(2)
        seen = []
        for k, (imgs, labels) in enumerate(loader):
            assert imgs.dtype == torch.uint8 and imgs.shape[1:] == (8, 8, 3)
            idx = torch.tensor(want[k * 8:(k + 1) * 8])
            ref_imgs, ref_labels = ds.get_batch(idx)
            assert torch.equal(imgs, ref_imgs) and torch.equal(labels, ref_labels)
            # the per-item path (what the reference's transform yields) describes the same data
            item, lab = ds[int(idx[0])]
            assert torch.allclose(item, imgs[0].permute(2, 0, 1).float() / 255) and lab == int(labels[0])
            seen += idx.tolist()
        assert seen == want


def test_batch_loader_with_uneven_workers_never_starves_a_batch_and_awaits_upload_events():
    """Workers that take turns being slow: whoever holds a batch number holds a buffer too, so the batch the consumer
    waits for can always be filled.  A buffer with a registered upload event goes back to the ring only after the
    event has been awaited."""
    import random
    import time

    from pytorch_distributed_training_amd.data import BatchLoader
    from pytorch_distributed_training_amd.data.sampler import ShardedSampler

    class Slow:
        def __len__(self):
            return 64 * 4

        def get_batch(self, idx, out=None):
            time.sleep(random.Random(int(idx[0])).choice([0.0, 0.0, 0.02]))
            imgs = idx.to(torch.uint8).view(-1, 1, 1, 1).expand(-1, 2, 2, 3)
            if out is not None:
                out.copy_(imgs)
                imgs = out
            return imgs, idx.clone()

    class Event:
        waited = 0

        def synchronize(self):
            Event.waited += 1

    ds = Slow()
    loader = BatchLoader(ds, 4, ShardedSampler(ds, shuffle=False), num_workers=4)
    t0 = time.time()
    got = []
    for imgs, labels in loader:
        loader.note_upload(imgs, Event())
        got.append(int(labels[0]))
        assert int(imgs[0, 0, 0, 0]) == int(labels[0]) % 256
    assert got == list(range(0, 256, 4)) and time.time() - t0 < 30
    assert Event.waited == 64  # every buffer's event was awaited before the buffer was reused or retired
    it = iter(loader)  # a consumer that stops early leaves no worker behind
    next(it)
    it.close()


def test_cifar10_python_batches_are_read_without_executing_pickle_code(tmp_path):
    import pickle

    import numpy as np
    import pytest

    d = tmp_path / "cifar-10-batches-py"
    d.mkdir()
    data = np.arange(4 * 3072, dtype=np.uint8).reshape(4, 3072)
    with open(d / "test_batch", "wb") as f:
        pickle.dump({"data": data, "labels": [1, 2, 3, 4], "filenames": ["a", "b", "c", "d"]}, f, protocol=2)
    ds = CIFAR10(str(tmp_path), train=False)
    assert len(ds) == 4 and ds.data.shape == (4, 32, 32, 3)
    img, lab = ds[2]
    assert lab == 3 and torch.equal((img * 255).round().to(torch.uint8), torch.from_numpy(data[2].reshape(3, 32, 32)))
    imgs, labels = ds.get_batch(torch.tensor([3, 0]))
    assert imgs.shape == (2, 32, 32, 3) and labels.tolist() == [4, 1]

    class Evil:
        def __reduce__(self):
            import os

            return (os.system, ("echo pwned > /dev/null",))

    with open(d / "test_batch", "wb") as f:
        pickle.dump({"data": data, "labels": [1, 2, 3, 4], "x": Evil()}, f, protocol=2)
    with pytest.raises(pickle.UnpicklingError):
        CIFAR10(str(tmp_path), train=False)
