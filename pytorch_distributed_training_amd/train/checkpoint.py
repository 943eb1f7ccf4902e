"""Rank-0 checkpointing and resume.

The reference never saves anything (no ``torch.save``/``state_dict`` in
/root/reference/src/main.py:1-88), so the format follows the plain PyTorch convention the
reference's model would use: the *model file* is exactly ``unwrapped_model.state_dict()`` -
torchvision parameter names, no ``module.`` prefix, logical NCHW fp32 tensors in standard
contiguous layout - so ``torchvision.models.resnet18(num_classes=10).load_state_dict(
torch.load(path))`` works on it.  Everything needed to *resume* (optimizer state, epoch,
step, RNG) goes into a sidecar file ``<path>.resume`` so the model file stays vanilla.

Only rank 0 writes; writes are atomic (temp file + rename); every rank passes a barrier
afterwards so nobody races ahead or exits while the file is incomplete.
"""

from __future__ import annotations

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


def _atomic_save(obj: Any, path: str) -> None:
    d = os.path.dirname(os.path.abspath(path))
    os.makedirs(d, exist_ok=True)
    fd, tmp = tempfile.mkstemp(dir=d, prefix="tmp_ckpt_", suffix=".part")
    try:
        with os.fdopen(fd, "wb") as f:
            torch.save(obj, f)
            f.flush()
            os.fsync(f.fileno())
        os.replace(tmp, path)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)


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
        _atomic_save(portable_state_dict(model), path)
        if optimizer is not None:
            side = {
                "optimizer": optimizer.state_dict(),
                "epoch": epoch,
                "step": step,
                "torch_rng": torch.get_rng_state(),
                "extra": extra or {},
                "format": 1,
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
        side = torch.load(side_path, map_location="cpu", weights_only=False)  # written by save_checkpoint above
        optimizer.load_state_dict(side["optimizer"])
        info.update(epoch=side.get("epoch", 0), step=side.get("step", 0), extra=side.get("extra", {}))
        if side.get("torch_rng") is not None:
            torch.set_rng_state(side["torch_rng"])
    return info
