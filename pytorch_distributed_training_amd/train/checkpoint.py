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
afterwards so nobody races ahead or exits while the file is incomplete.  The sidecar records the
SHA-256 of the model file it belongs to, so a pair whose halves come from different saves (the
job died between the two writes) is recognised as torn instead of silently mixing two steps.

Durability: every save first writes a *generation* pair ``<path>.step<N>`` /
``<path>.step<N>.resume``, then points ``<path>`` and ``<path>.resume`` at it (hard links, each
replaced atomically), then prunes all but the newest ``keep`` generations.  A resume takes the
main pair when it is intact and otherwise the newest intact generation, and says so; a torn
newest checkpoint therefore costs the steps since the previous save, never the job.  All files hold
nothing but tensors and plain values and are read with ``weights_only=True``.
"""

from __future__ import annotations

import glob
import hashlib
import os
import re
import shutil
import sys
import tempfile
from typing import Any, Dict, List, Optional, Tuple

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


_GEN = re.compile(r"\.step(\d+)$")
_saves = 0  # saves done by this process (fault injection: PDT_FAULT_INJECT_SAVE)


def generation_path(path: str, step: int) -> str:
    return f"{path}.step{int(step):09d}"


def list_generations(path: str) -> List[Tuple[int, str]]:
    """``[(step, model_file)]`` of the generation files next to ``path``, newest first."""
    out = []
    for f in glob.glob(glob.escape(path) + ".step*"):
        m = _GEN.search(f)
        if m:
            out.append((int(m.group(1)), f))
    return sorted(out, reverse=True)


def _link_over(src: str, dst: str) -> None:
    """Make ``dst`` the same file as ``src`` atomically (hard link; a copy on file systems without links)."""
    d = os.path.dirname(os.path.abspath(dst))
    fd, tmp = tempfile.mkstemp(dir=d, prefix="tmp_ckpt_", suffix=".lnk")
    os.close(fd)
    os.remove(tmp)
    try:
        try:
            os.link(src, tmp)
        except OSError:
            shutil.copyfile(src, tmp)
        os.replace(tmp, dst)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)


def _maybe_die(stage: str) -> None:
    """``PDT_FAULT_INJECT_SAVE=<n>:<stage>``: end the process hard during the n-th save of this process, right after
    ``stage`` (``model`` = generation model file written, ``sidecar``, ``link_model`` = main model file replaced).
    ``PDT_FAULT_INJECT_ONCE_FILE`` limits it to the first incarnation of a restarted job."""
    spec = os.environ.get("PDT_FAULT_INJECT_SAVE")
    if not spec:
        return
    n, _, want = spec.partition(":")
    if int(n) != _saves or want != stage:
        return
    marker = os.environ.get("PDT_FAULT_INJECT_ONCE_FILE")
    if marker:
        if os.path.exists(marker):
            return
        open(marker, "w").close()
    sys.stderr.write(f"[pdt fault] dying in save #{_saves} after stage {stage!r}\n")
    sys.stderr.flush()
    os._exit(13)


def save_checkpoint(
    path: str,
    model: nn.Module,
    env: DistEnv,
    optimizer: Optional[torch.optim.Optimizer] = None,
    epoch: int = 0,
    step: int = 0,
    extra: Optional[Dict[str, Any]] = None,
    keep: int = 2,
) -> None:
    """Rank 0 writes generation ``step`` and re-points the main pair at it; ``keep`` newest generations stay on disk
    (``keep <= 0``: no generations, the main pair is overwritten in place)."""
    global _saves
    if env.is_main:
        _saves += 1
        side = None
        if optimizer is not None:
            side = {
                "optimizer": optimizer.state_dict(),
                "epoch": epoch,
                "step": step,
                "torch_rng": torch.get_rng_state(),
                "extra": extra or {},
                "format": 3,
            }
        if keep <= 0 or side is None:
            digest = _atomic_save(portable_state_dict(model), path)
            _maybe_die("model")
            if side is not None:
                side["model_sha256"] = digest
                _atomic_save(side, resume_path(path))
        else:
            gen = generation_path(path, step)
            side["model_sha256"] = _atomic_save(portable_state_dict(model), gen)
            _maybe_die("model")
            _atomic_save(side, resume_path(gen))
            _maybe_die("sidecar")
            _link_over(gen, path)
            _maybe_die("link_model")
            _link_over(resume_path(gen), resume_path(path))
            for _, old in list_generations(path)[keep:]:
                for f in (resume_path(old), old):  # sidecar first: a generation without one is never taken for intact
                    if os.path.exists(f):
                        os.remove(f)
    barrier(env)


def load_model_state(path: str, model: nn.Module, strict: bool = True) -> None:
    """Every rank reads the same file; accepts files written with a ``module.`` prefix too."""
    sd = torch.load(path, map_location="cpu", weights_only=True)
    if sd and all(k.startswith("module.") for k in sd):
        sd = {k[len("module."):]: v for k, v in sd.items()}
    unwrap(model).load_state_dict(sd, strict=strict)


def pair_state(model_file: str) -> str:
    """``intact`` (sidecar present and its digest matches), ``torn`` (digest differs, or a file is unreadable),
    ``weights_only`` (no sidecar) or ``missing``."""
    if not os.path.exists(model_file):
        return "missing"
    side_file = resume_path(model_file)
    if not os.path.exists(side_file):
        return "weights_only"
    try:
        side = torch.load(side_file, map_location="cpu", weights_only=True)
        want = side.get("model_sha256")
        return "intact" if (want is None or file_sha256(model_file) == want) else "torn"
    except Exception:
        return "torn"


def find_resumable(path: str) -> Optional[str]:
    """The model file a resume from ``path`` should read: ``path`` itself when its pair is intact (or it is a plain
    weights file with no generations beside it), else the newest generation whose pair is intact, else ``None``."""
    state = pair_state(path)
    gens = list_generations(path)
    if state == "intact" or (state == "weights_only" and not gens):
        return path
    for _, gen in gens:
        if pair_state(gen) == "intact":
            return gen
    return path if state == "weights_only" else None


def load_checkpoint(
    path: str,
    model: nn.Module,
    optimizer: Optional[torch.optim.Optimizer] = None,
    strict: bool = True,
    fallback: bool = True,
) -> Dict[str, Any]:
    """Restore model (+ optimizer/progress when the sidecar exists).  Returns ``{"epoch", "step", "extra", "source"}``.

    ``fallback``: when the pair at ``path`` is torn, take the newest intact generation next to it instead of raising;
    ``source`` names the file that was read."""
    source = path
    if pair_state(path) == "torn" or not os.path.exists(path):
        alt = find_resumable(path) if fallback else None
        if alt is None:
            raise TornCheckpointError(
                f"{path} and {resume_path(path)} come from different saves (the job died between the two writes, or "
                f"one file was replaced): the optimizer state and step do not belong to these weights, and no intact "
                f"earlier generation ({path}.step*) exists"
            )
        source = alt
    load_model_state(source, model, strict)
    info: Dict[str, Any] = {"epoch": 0, "step": 0, "extra": {}, "source": source}
    side_path = resume_path(source)
    if optimizer is not None and os.path.exists(side_path):
        side = torch.load(side_path, map_location="cpu", weights_only=True)
        optimizer.load_state_dict(side["optimizer"])
        info.update(epoch=side.get("epoch", 0), step=side.get("step", 0), extra=side.get("extra", {}))
        info["torch_rng"] = side.get("torch_rng")
    return info
