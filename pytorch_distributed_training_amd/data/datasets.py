"""Datasets: synthetic image/token data and an in-repo CIFAR-10 reader.

The reference trains on ``torchvision.datasets.CIFAR10(train=False, download=True)`` with
``ToTensor()`` (/root/reference/src/main.py:44-47): uint8 HWC images become float32 CHW in
[0, 1], labels are ints in [0, 10).  torchvision is not a dependency here, so the same
on-disk formats are parsed directly.  When the files are absent (there is no network on
the training boxes) the loader falls back to synthetic data of the same shape.

Image datasets keep their samples as **uint8 HWC** - the form decoders produce - and offer two access paths:
``__getitem__`` returns what the reference's transform returns (float32 CHW in [0, 1]); ``get_batch(indices, out)``
gathers a whole batch of raw uint8 NHWC images with one vectorised copy, which is what the batch loader ships to
the GPU (one byte per element over PCIe; ``ToTensor`` and the layout change happen in a GPU kernel).
"""

from __future__ import annotations

import io
import os
import pickle
import tarfile
from pathlib import Path
from typing import List, Optional, Tuple

import numpy as np
import torch
from torch.utils.data import Dataset

CIFAR10_CLASSES = [
    "airplane", "automobile", "bird", "cat", "deer", "dog", "frog", "horse", "ship", "truck",
]


class SyntheticImageDataset(Dataset):
    """Deterministic random images/labels of a given shape: uint8 HWC storage, float32 CHW in [0, 1] per item."""

    def __init__(self, length: int = 10000, image_size: int = 32, channels: int = 3, num_classes: int = 10, seed: int = 0):
        self.length = length
        self.shape = (channels, image_size, image_size)
        self.num_classes = num_classes
        self.classes = [str(i) for i in range(num_classes)]
        self.seed = seed
        g = torch.Generator().manual_seed(seed)
        # A small bank of distinct images keeps memory flat for ImageNet-sized runs.
        self._bank = torch.randint(0, 256, (min(length, 256), image_size, image_size, channels), dtype=torch.uint8, generator=g)
        self._labels = torch.randint(0, num_classes, (length,), generator=g)

    def __len__(self) -> int:
        return self.length

    def __getitem__(self, i: int) -> Tuple[torch.Tensor, int]:
        img = self._bank[i % self._bank.shape[0]]
        return img.permute(2, 0, 1).to(torch.float32).div_(255.0), int(self._labels[i])

    def get_batch(self, indices: torch.Tensor, out: Optional[torch.Tensor] = None):
        """uint8 ``[B, H, W, C]`` images and int64 labels of the samples ``indices`` (one gather, no per-item work)."""
        imgs = torch.index_select(self._bank, 0, indices % self._bank.shape[0], out=out)
        return imgs, self._labels[indices]


class SyntheticTokenDataset(Dataset):
    """Random token sequences for language-model throughput runs: (input_ids, next-token targets)."""

    def __init__(self, length: int = 4096, seq_len: int = 1024, vocab_size: int = 50257, seed: int = 0):
        self.length, self.seq_len, self.vocab_size = length, seq_len, vocab_size
        self.classes = list(range(vocab_size))
        g = torch.Generator().manual_seed(seed)
        self._bank = torch.randint(0, vocab_size, (min(length, 64), seq_len + 1), generator=g)

    def __len__(self) -> int:
        return self.length

    def __getitem__(self, i: int):
        row = self._bank[i % self._bank.shape[0]]
        return row[:-1], row[1:]


class CIFAR10(Dataset):
    """CIFAR-10 from ``<root>/cifar-10-batches-py`` (pickle) or ``<root>/cifar-10-batches-bin``.

    Returns float32 CHW tensors scaled to [0, 1] (the effect of ``ToTensor``) and int labels.
    ``train=False`` selects the 10 000-image test split, which is what the reference trains
    on (main.py:47).

    ``wait_for_main``: in a multi-rank job, ``(is_main, barrier)``.  Only the main rank unpacks the archive; the
    others wait at the barrier and then read the unpacked files (every rank of the reference downloads and
    unpacks into the same directory at once, main.py:47).
    The python-format batches are pickles: they are read with an unpickler that can build nothing but the numpy
    arrays, lists, dicts and strings those files consist of.
    """

    py_dir = "cifar-10-batches-py"
    bin_dir = "cifar-10-batches-bin"
    archive = "cifar-10-python.tar.gz"

    def __init__(self, root: str, train: bool = False, wait_for_main=None):
        self.root = Path(os.path.expanduser(root))
        self.train = train
        self.classes = list(CIFAR10_CLASSES)
        self._wait_for_main = wait_for_main
        data, labels = self._load()
        # stored planar (CHW) on disk; kept interleaved (HWC) like decoded images
        self.data = torch.from_numpy(np.ascontiguousarray(data.transpose(0, 2, 3, 1)))  # uint8 [N, 32, 32, 3]
        self.labels = torch.from_numpy(labels.astype(np.int64))

    @classmethod
    def available(cls, root: str) -> bool:
        r = Path(os.path.expanduser(root))
        return (r / cls.py_dir).is_dir() or (r / cls.bin_dir).is_dir() or (r / cls.archive).is_file()

    def _unpack_archive(self) -> None:
        have_files = (self.root / self.py_dir).is_dir() or (self.root / self.bin_dir).is_dir()
        need = not have_files and (self.root / self.archive).is_file()
        is_main, barrier = self._wait_for_main if self._wait_for_main is not None else (True, None)
        if need and is_main:
            tmp = self.root / (self.py_dir + ".unpacking")
            with tarfile.open(self.root / self.archive, "r:gz") as tf:
                tf.extractall(tmp, filter="data")
            os.replace(tmp / self.py_dir, self.root / self.py_dir)  # appears complete or not at all
            try:
                tmp.rmdir()
            except OSError:
                pass
        if barrier is not None:
            barrier()

    def _load(self) -> Tuple[np.ndarray, np.ndarray]:
        self._unpack_archive()
        if (self.root / self.py_dir).is_dir():
            names = [f"data_batch_{i}" for i in range(1, 6)] if self.train else ["test_batch"]
            xs: List[np.ndarray] = []
            ys: List[np.ndarray] = []
            for n in names:
                with open(self.root / self.py_dir / n, "rb") as f:
                    entry = _ArraysOnlyUnpickler(f, encoding="latin1").load()
                xs.append(np.asarray(entry["data"], dtype=np.uint8).reshape(-1, 3, 32, 32))
                ys.append(np.asarray(entry.get("labels", entry.get("fine_labels")), dtype=np.int64))
            return np.concatenate(xs), np.concatenate(ys)
        if (self.root / self.bin_dir).is_dir():
            names = [f"data_batch_{i}.bin" for i in range(1, 6)] if self.train else ["test_batch.bin"]
            raw = np.concatenate(
                [np.fromfile(self.root / self.bin_dir / n, dtype=np.uint8).reshape(-1, 3073) for n in names]
            )
            return raw[:, 1:].reshape(-1, 3, 32, 32).copy(), raw[:, 0].astype(np.int64)
        raise FileNotFoundError(
            f"CIFAR-10 not found under {self.root} (expected {self.py_dir}/, {self.bin_dir}/ or {self.archive})"
        )

    def __len__(self) -> int:
        return self.data.shape[0]

    def __getitem__(self, i: int):
        return self.data[i].permute(2, 0, 1).to(torch.float32).div_(255.0), int(self.labels[i])

    def get_batch(self, indices: torch.Tensor, out: Optional[torch.Tensor] = None):
        """uint8 ``[B, 32, 32, 3]`` images and int64 labels of the samples ``indices``."""
        return torch.index_select(self.data, 0, indices, out=out), self.labels[indices]


class _ArraysOnlyUnpickler(pickle.Unpickler):
    """Unpickler for the CIFAR python batches: numpy arrays and plain containers, nothing else.  A pickle can name
    any importable callable; a dataset file fetched from somewhere must not be able to run one."""

    _ALLOWED = {
        ("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"),
        ("numpy", "ndarray"), ("numpy", "dtype"),
        ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar"),
        ("_codecs", "encode"),  # how protocol-2 pickles written by Python 2 carry array bytes
    }

    def find_class(self, module, name):
        if (module, name) in self._ALLOWED:
            return super().find_class(module, name)
        raise pickle.UnpicklingError(f"dataset file wants {module}.{name}: only numpy arrays and plain containers are accepted")


def build_dataset(
    name: str,
    data_dir: str,
    model_kind: str = "image",
    image_size: int = 32,
    num_classes: int = 10,
    length: Optional[int] = None,
    seq_len: int = 1024,
    vocab_size: int = 50257,
    train: bool = False,
    seed: int = 0,
    wait_for_main=None,
) -> Tuple[Dataset, str]:
    """Returns ``(dataset, description)``.  ``name`` is ``auto`` | ``cifar10`` | ``synthetic``."""
    if model_kind == "lm":
        ds = SyntheticTokenDataset(length or 4096, seq_len, vocab_size, seed)
        return ds, f"synthetic tokens seq={seq_len} vocab={vocab_size}"
    if name in ("auto", "cifar10") and image_size == 32 and num_classes == 10:
        if CIFAR10.available(data_dir):
            return (CIFAR10(data_dir, train=train, wait_for_main=wait_for_main),
                    f"CIFAR-10 ({'train' if train else 'test'} split) from {data_dir}")
        if name == "cifar10":
            raise FileNotFoundError(
                f"--dataset cifar10: no CIFAR-10 files under {data_dir} and no network to download them; "
                f"use --dataset synthetic"
            )
    ds = SyntheticImageDataset(length or 10000, image_size, 3, num_classes, seed)
    return ds, f"synthetic images {image_size}x{image_size} classes={num_classes}"
