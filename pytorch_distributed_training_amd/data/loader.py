"""Batch delivery to the device.

The reference uploads every batch with a blocking ``.to(device)`` from pageable memory, after a per-item
``ToTensor`` in worker processes (/root/reference/src/main.py:44-47,61,69-70): four bytes per element cross
PCIe and every image is touched several times on the host.  Here

* :class:`BatchLoader` asks the dataset for whole batches (``get_batch``: one vectorised gather of raw uint8 HWC
  images) from a few worker threads, straight into a ring of pinned buffers - no per-item Python, no pickling
  between processes, one byte per element;
* :class:`DevicePrefetcher` copies batch *k+1* on a separate HIP stream while step *k* runs and converts it there
  (uint8 NHWC -> normalised bf16 NHWC, one kernel), so the training stream only waits for an event;
* datasets without ``get_batch`` go through ``torch.utils.data.DataLoader`` (worker processes, pinned collation);
* :class:`DeviceSyntheticLoader` keeps a fixed set of batches resident in HBM so throughput benchmarks measure
  the trainer, not the host.
"""

from __future__ import annotations

import queue
import threading
from typing import Iterator, List, Optional, Tuple

import torch
from torch.utils.data import DataLoader, Dataset

from .sampler import ShardedSampler


def build_loader(
    dataset: Dataset,
    batch_size: int,
    num_workers: int,
    rank: int = 0,
    world_size: int = 1,
    shuffle: bool = False,
    drop_last: bool = False,
    pin_memory: bool = False,
    seed: int = 0,
) -> Tuple[object, ShardedSampler]:
    """``batch_size`` is per process, as in the reference (main.py:61)."""
    # one sampler class for any world size (a single process is a one-replica shard): the order depends only on
    # (seed, epoch), and a pass can start behind the samples a resumed job has already seen
    sampler = ShardedSampler(dataset, world_size, rank, shuffle=shuffle, seed=seed, drop_last=drop_last)
    if hasattr(dataset, "get_batch"):
        return BatchLoader(dataset, batch_size, sampler, num_workers, drop_last, pin_memory), sampler
    loader = DataLoader(
        dataset,
        batch_size=batch_size,
        num_workers=num_workers,
        sampler=sampler,
        shuffle=False,
        drop_last=drop_last,
        pin_memory=pin_memory,
        persistent_workers=num_workers > 0,
    )
    return loader, sampler


class BatchLoader:
    """Whole batches from a dataset with ``get_batch(indices, out) -> (uint8 [B, H, W, C], labels)``.

    ``num_workers`` threads (at least one) each gather a batch into a buffer taken from a ring of pinned host
    tensors; batches are delivered in sampler order.  The gather is a single ``index_select`` (it releases the
    interpreter lock), so a handful of threads saturates host memory bandwidth where per-item loading would need
    dozens of processes.  A worker takes a buffer first and a batch number second, so whoever holds a batch number
    can always fill it (numbers handed out in order + buffers handed out before numbers = no batch can starve behind
    later ones).  A buffer returns to the ring when the batch after the next one is requested; if the consumer
    registered an upload event for it (:meth:`note_upload`, the prefetcher does), that event is awaited first, so
    a pinned buffer is never refilled while its host-to-device copy could still be reading it."""

    def __init__(self, dataset, batch_size: int, sampler, num_workers: int = 2, drop_last: bool = False,
                 pin_memory: bool = False):
        self.dataset, self.batch_size, self.sampler = dataset, int(batch_size), sampler
        self.workers = max(1, int(num_workers))
        self.drop_last = drop_last
        self.pin = bool(pin_memory) and torch.cuda.is_available()
        self._ring: List[torch.Tensor] = []
        self._uploads = {}  # storage address of a ring buffer -> event recorded behind its latest upload

    def note_upload(self, host_tensor: torch.Tensor, event) -> None:
        """The consumer started an asynchronous copy out of ``host_tensor`` (a batch this loader yielded) and
        ``event`` completes when the copy has."""
        self._uploads[host_tensor.untyped_storage().data_ptr()] = event

    def _release(self, free: "queue.Queue", buf: torch.Tensor) -> None:
        ev = self._uploads.pop(buf.untyped_storage().data_ptr(), None)
        if ev is not None:
            ev.synchronize()  # long done in practice: two batches were fetched since
        free.put(buf)

    def __len__(self) -> int:
        n = len(self.sampler)
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def _buffers(self, shape, dtype, count: int) -> "queue.Queue":
        if len(self._ring) != count or self._ring[0].shape != shape or self._ring[0].dtype != dtype:
            self._ring = [torch.empty(shape, dtype=dtype, pin_memory=self.pin) for _ in range(count)]
        q: "queue.Queue" = queue.Queue()
        for b in self._ring:
            q.put(b)
        return q

    def __iter__(self) -> Iterator:
        idx = torch.tensor(list(iter(self.sampler)), dtype=torch.int64)
        bs = self.batch_size
        starts = list(range(0, len(idx) - (bs - 1 if self.drop_last else 0), bs)) if len(idx) else []
        if not starts:
            return
        probe, _ = self.dataset.get_batch(idx[:1])
        shape = (bs,) + tuple(probe.shape[1:])
        depth = self.workers + 3  # in flight in the workers + the two batches the consumer may still be using
        free = self._buffers(shape, probe.dtype, depth)
        results = {}
        cond = threading.Condition()
        todo: "queue.Queue" = queue.Queue()
        for k, s0 in enumerate(starts):
            todo.put((k, s0))
        stop = threading.Event()
        errors: List[BaseException] = []

        def work():
            while not stop.is_set():
                buf = free.get()  # the buffer first: a batch number is only ever held together with a buffer
                try:
                    k, s0 = todo.get_nowait()
                except queue.Empty:
                    free.put(buf)
                    return
                if stop.is_set():
                    free.put(buf)
                    return
                sel = idx[s0:s0 + bs]
                try:
                    out = buf if len(sel) == bs else buf[: len(sel)]
                    imgs, labels = self.dataset.get_batch(sel, out=out)
                except BaseException as e:  # surfaced in the consumer
                    errors.append(e)
                    imgs, labels = None, None
                with cond:
                    results[k] = (imgs, labels, buf)
                    cond.notify_all()

        # workers take batch numbers in order, so holding `depth` buffers bounds how far they run ahead
        threads = [threading.Thread(target=work, name=f"pdt-loader-{i}", daemon=True) for i in range(self.workers)]
        for t in threads:
            t.start()
        held: List[torch.Tensor] = []
        try:
            for k in range(len(starts)):
                with cond:
                    while k not in results:
                        cond.wait(timeout=1.0)
                    imgs, labels, buf = results.pop(k)
                if errors:
                    raise errors[0]
                held.append(buf)
                if len(held) > 2:
                    self._release(free, held.pop(0))
                yield imgs, labels
        finally:
            stop.set()
            for b in held:
                self._release(free, b)
            for _ in threads:  # unblock workers waiting for a buffer
                free.put(self._ring[0])
            for t in threads:
                t.join(timeout=5)


_copy_streams = {}


def copy_stream(device: torch.device) -> "torch.cuda.Stream":
    """The process's one copy stream for ``device``.  Every prefetcher (training, evaluation, a second trainer in the
    same process) uploads on it: a stream per prefetcher would push the process over the hardware queues it is
    served (step + copy + communication + graph capture is the whole budget of four)."""
    device = torch.device(device)
    key = device.index if device.index is not None else torch.cuda.current_device()
    if key not in _copy_streams:
        _copy_streams[key] = torch.cuda.Stream(device=device)
    return _copy_streams[key]


class DevicePrefetcher:
    """Wraps a host loader; overlaps the H2D copy of the next batch - and, for raw uint8 image batches, their
    conversion to the models' input form - with the current step.  ``convert(images_on_device) -> tensor`` runs on
    the copy stream right behind the upload."""

    def __init__(self, loader, device: torch.device, convert=None):
        self.loader = loader
        self.device = torch.device(device)
        self.convert = convert
        self.stream = copy_stream(self.device) if self.device.type == "cuda" else None

    def __len__(self) -> int:
        return len(self.loader)

    def _upload(self, batch):
        if self.stream is None:
            out = tuple(t.to(self.device) for t in batch)
            return (self.convert(out[0]),) + out[1:] if (self.convert and out[0].dtype == torch.uint8) else out
        with torch.cuda.stream(self.stream):
            out = tuple(t.to(self.device, non_blocking=True) for t in batch)
            note = getattr(self.loader, "note_upload", None)
            if note is not None and isinstance(batch[0], torch.Tensor) and batch[0].device.type == "cpu":
                ev = torch.cuda.Event()
                ev.record(self.stream)
                note(batch[0], ev)  # the loader's pinned buffer may be refilled once this copy is through
            if self.convert is not None and out[0].dtype == torch.uint8:
                raw = out[0]
                out = (self.convert(raw),) + out[1:]
                raw.record_stream(self.stream)
            return out

    def __iter__(self) -> Iterator:
        it = iter(self.loader)
        try:
            nxt = self._upload(next(it))
        except StopIteration:
            return
        while nxt is not None:
            if self.stream is not None:
                torch.cuda.current_stream(self.device).wait_stream(self.stream)
                for t in nxt:
                    t.record_stream(torch.cuda.current_stream(self.device))
            cur = nxt
            try:
                nxt = self._upload(next(it))
            except StopIteration:
                nxt = None
            yield cur


class DeviceSyntheticLoader:
    """``steps`` batches per epoch drawn round-robin from ``n_resident`` device-resident batches."""

    def __init__(self, make_batch, steps: int, n_resident: int = 2):
        self.batches = [make_batch(i) for i in range(max(1, n_resident))]
        self.steps = steps
        self._skip = 0

    def __len__(self) -> int:
        return self.steps

    def skip_next(self, n: int) -> None:
        """Start the next pass at batch ``n`` (resuming inside an epoch)."""
        self._skip = max(0, int(n))

    def __iter__(self):
        start, self._skip = self._skip, 0
        for i in range(start, self.steps):
            yield self.batches[i % len(self.batches)]


def synthetic_image_batches(batch_size: int, image_size: int, num_classes: int, device, dtype=torch.float32,
                            channels_last: bool = False, seed: int = 0):
    def make(i: int):
        g = torch.Generator(device="cpu").manual_seed(seed + i)
        x = torch.rand((batch_size, 3, image_size, image_size), generator=g).to(device=device, dtype=dtype)
        if channels_last:
            x = x.contiguous(memory_format=torch.channels_last)
        y = torch.randint(0, num_classes, (batch_size,), generator=g).to(device)
        return x, y

    return make


def synthetic_token_batches(batch_size: int, seq_len: int, vocab_size: int, device, seed: int = 0):
    def make(i: int):
        g = torch.Generator(device="cpu").manual_seed(seed + i)
        t = torch.randint(0, vocab_size, (batch_size, seq_len + 1), generator=g).to(device)
        return t[:, :-1].contiguous(), t[:, 1:].contiguous()

    return make
