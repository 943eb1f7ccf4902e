"""Batch delivery to the device.

The reference uploads every batch with a blocking ``.to(device)`` from pageable memory
(/root/reference/src/main.py:61,69-70).  Here the host loader pins its batches and
:class:`DevicePrefetcher` copies batch *k+1* on a separate HIP stream while step *k* runs;
:class:`DeviceSyntheticLoader` keeps a fixed set of batches resident in HBM so throughput
benchmarks measure the trainer, not the host.
"""

from __future__ import annotations

from typing import Iterator, Optional, Tuple

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
) -> Tuple[DataLoader, ShardedSampler]:
    """``batch_size`` is per process, as in the reference (main.py:61)."""
    # one sampler class for any world size (a single process is a one-replica shard): the order depends only on
    # (seed, epoch), and a pass can start behind the samples a resumed job has already seen
    sampler = ShardedSampler(dataset, world_size, rank, shuffle=shuffle, seed=seed, drop_last=drop_last)
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


class DevicePrefetcher:
    """Wraps a host loader; overlaps the H2D copy of the next batch with the current step."""

    def __init__(self, loader, device: torch.device):
        self.loader = loader
        self.device = torch.device(device)
        self.stream = torch.cuda.Stream(device=self.device) if self.device.type == "cuda" else None

    def __len__(self) -> int:
        return len(self.loader)

    def _upload(self, batch):
        if self.stream is None:
            return tuple(t.to(self.device) for t in batch)
        with torch.cuda.stream(self.stream):
            return tuple(t.to(self.device, non_blocking=True) for t in batch)

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
