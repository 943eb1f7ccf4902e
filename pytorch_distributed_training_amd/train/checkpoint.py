"""Rank-0 checkpointing and resume.

The reference never saves anything (no ``torch.save``/``state_dict`` in
/root/reference/src/main.py:1-88), so the format follows the plain PyTorch convention the
reference's model would use: the *model file* is exactly ``unwrapped_model.state_dict()`` -
torchvision parameter names, no ``module.`` prefix, logical NCHW fp32 tensors in standard
contiguous layout - so ``torchvision.models.resnet18(num_classes=10).load_state_dict(
torch.load(path))`` works on it.  Everything needed to *resume* (optimizer state, epoch,
step, position inside the epoch, RNG) goes into a sidecar file ``<path>.resume`` so the model
file stays vanilla.

Only rank 0 writes; writes are atomic (temp file + rename); every rank passes a barrier
afterwards so nobody races ahead or exits while the file is incomplete.  The two files are
replaced one after the other, so a crash in between leaves new weights next to an old sidecar:
the sidecar therefore records the SHA-256 of the model file it belongs to, and a resume that
finds a different model file refuses the pair instead of silently mixing two steps.  Both
files hold nothing but tensors and plain values and are read with ``weights_only=True``.
"""

from __future__ import annotations

import hashlib
import os
import tempfile
from typing import Any, Dict, Optional

import torch
import torch.nn as nn

from ..parallel.env import DistEnv, barrier


def unwrap(model: nn.Module) -> nn.Module:
    return model.module if hasattr(model, "module") and isinstance(model.module, nn.Module) else model


def portable_state_dict(model: nn.Module) -> Dict[str, torch.Tensor]:
    """CPU copy of the model state with vanilla layout (contiguous, fp32 for floating tensors)."""
    out: Dict[str, torch.Tensor] = {}
    for k, v in unwrap(model).state_dict().items():
        t = v.detach().to("cpu")
        if t.is_floating_point():
            t = t.to(torch.float32)
        out[k] = t.contiguous().clone()
    return out


class TornCheckpointError(RuntimeError):
    """The model file and its resume sidecar were written by different saves."""


class _HashingWriter:
    """File-like object that hashes everything written through it."""

    def __init__(self, f):
        self._f = f
        self.sha = hashlib.sha256()

    def write(self, b):
        self.sha.update(b)
        return self._f.write(b)

    def flush(self):
        self._f.flush()


def _atomic_save(obj: Any, path: str) -> str:
    """Write ``obj`` with ``torch.save`` to a temporary file, fsync, rename over ``path``; returns the SHA-256
    of the bytes written."""
    d = os.path.dirname(os.path.abspath(path))
    os.makedirs(d, exist_ok=True)
    fd, tmp = tempfile.mkstemp(dir=d, prefix="tmp_ckpt_", suffix=".part")
    try:
        with os.fdopen(fd, "wb") as f:
            w = _HashingWriter(f)
            # the zip container seeks backwards to patch headers when given a real file; a write-only stream keeps
            # the bytes hashed equal to the bytes on disk
            torch.save(obj, w)
            f.flush()
            os.fsync(f.fileno())
        os.replace(tmp, path)
        return w.sha.hexdigest()
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)


def file_sha256(path: str) -> str:
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for block in iter(lambda: f.read(1 << 22), b""):
            h.update(block)
    return h.hexdigest()


def resume_path(path: str) -> str:
    return path + ".resume"


def save_checkpoint(
    path: str,
    model: nn.Module,
    env: DistEnv,
    optimizer: Optional[torch.optim.Optimizer] = None,
    epoch: int = 0,
    step: int = 0,
    extra: Optional[Dict[str, Any]] = None,
) -> None:
    if env.is_main:
        digest = _atomic_save(portable_state_dict(model), path)
        if optimizer is not None:
            side = {
                "optimizer": optimizer.state_dict(),
                "epoch": epoch,
                "step": step,
                "torch_rng": torch.get_rng_state(),
                "extra": extra or {},
                "model_sha256": digest,
                "format": 2,
            }
            _atomic_save(side, resume_path(path))
    barrier(env)


def load_model_state(path: str, model: nn.Module, strict: bool = True) -> None:
    """Every rank reads the same file; accepts files written with a ``module.`` prefix too."""
    sd = torch.load(path, map_location="cpu", weights_only=True)
    if sd and all(k.startswith("module.") for k in sd):
        sd = {k[len("module."):]: v for k, v in sd.items()}
    unwrap(model).load_state_dict(sd, strict=strict)


def load_checkpoint(
    path: str,
    model: nn.Module,
    optimizer: Optional[torch.optim.Optimizer] = None,
    strict: bool = True,
) -> Dict[str, Any]:
    """Restore model (+ optimizer/progress when the sidecar exists).  Returns ``{"epoch", "step", "extra"}``."""
    load_model_state(path, model, strict)
    info: Dict[str, Any] = {"epoch": 0, "step": 0, "extra": {}}
    side_path = resume_path(path)
    if optimizer is not None and os.path.exists(side_path):
        side = torch.load(side_path, map_location="cpu", weights_only=True)
        want = side.get("model_sha256")
        if want is not None and file_sha256(path) != want:
            raise TornCheckpointError(
                f"{path} and {side_path} come from different saves (the job died between the two writes, or one file "
                f"was replaced): the optimizer state and step do not belong to these weights"
            )
        optimizer.load_state_dict(side["optimizer"])
        info.update(epoch=side.get("epoch", 0), step=side.get("step", 0), extra=side.get("extra", {}))
        info["torch_rng"] = side.get("torch_rng")
    return info
