"""Sharded sampling for data parallelism.

The reference passes the whole dataset to every rank's ``DataLoader``
(/root/reference/src/main.py:61), so under DDP each rank sees identical batches and the
work is duplicated.  ``ShardedSampler`` gives each rank a disjoint, equally sized slice of
a (per-epoch reshuffled) permutation - the semantics of
``torch.utils.data.DistributedSampler``, re-implemented here so sharding does not depend on
an initialised process group and can be unit-tested on plain integers.
"""

from __future__ import annotations

import math
from typing import Iterator, Optional, Sized

import torch
from torch.utils.data import Sampler


class ShardedSampler(Sampler[int]):
    def __init__(
        self,
        dataset: Sized,
        num_replicas: int = 1,
        rank: int = 0,
        shuffle: bool = True,
        seed: int = 0,
        drop_last: bool = False,
    ):
        if num_replicas < 1 or not (0 <= rank < num_replicas):
            raise ValueError(f"invalid rank {rank} for {num_replicas} replicas")
        self.n = len(dataset)
        self.num_replicas = num_replicas
        self.rank = rank
        self.shuffle = shuffle
        self.seed = seed
        self.drop_last = drop_last
        self.epoch = 0
        self._start = 0
        if drop_last and self.n % num_replicas != 0:
            self.num_samples = self.n // num_replicas  # tail dropped: every index used at most once
        else:
            self.num_samples = math.ceil(self.n / num_replicas)  # padded by wrapping around
        self.total_size = self.num_samples * num_replicas

    @property
    def valid_samples(self) -> int:
        """How many leading samples of this rank's shard are not wrap-around padding.  Shard position ``j`` is global
        position ``rank + j * num_replicas``; positions from ``n`` on repeat earlier samples.  (An evaluation that
        counts every sample once stops there.)"""
        if self.rank >= self.n:
            return 0
        return min(self.num_samples, (self.n - self.rank + self.num_replicas - 1) // self.num_replicas)

    def set_epoch(self, epoch: int) -> None:
        """Change the permutation; call once per epoch with the same value on every rank."""
        self.epoch = int(epoch)

    def set_start(self, n_samples: int) -> None:
        """Skip the first ``n_samples`` of this rank's shard in the *next* pass only (resuming inside an epoch)."""
        self._start = max(0, int(n_samples))

    def _global_order(self) -> list:
        if self.shuffle:
            g = torch.Generator()
            g.manual_seed(self.seed + self.epoch)
            idx = torch.randperm(self.n, generator=g).tolist()
        else:
            idx = list(range(self.n))
        if self.total_size > len(idx):
            pad = self.total_size - len(idx)
            reps = math.ceil(pad / max(len(idx), 1))
            idx = idx + (idx * reps)[:pad]
        return idx[: self.total_size]

    def __iter__(self) -> Iterator[int]:
        order = self._global_order()
        start, self._start = self._start, 0
        return iter(order[self.rank : self.total_size : self.num_replicas][start:])

    def __len__(self) -> int:
        return self.num_samples
