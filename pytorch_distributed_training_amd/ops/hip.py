"""GPU implementations of the fused operators: ctypes bindings to ``libpdt_kernels.so`` and the
``autograd.Function`` wrappers that give them backward passes.

Conventions
-----------
* Activations are bf16.  Image tensors are logical NCHW with channels_last strides, i.e. physical
  NHWC - the layout every kernel indexes.  Matrices are row-major ``[rows, features]``.
* Parameters stay fp32 ("master" weights).  Kernels read bf16 copies kept in :class:`ShadowCache`;
  the fused optimizer refreshes those copies in the same launch that updates the masters.
* Weight gradients are produced in fp32, in the parameter's own physical element order.
* Every launcher returns a status; a non-zero status raises.  Nothing here falls back to PyTorch ops.

Role parity: the ATen/cuDNN/cuBLAS kernels implicitly launched by
/root/reference/src/main.py:74-79.
"""

from __future__ import annotations

import ctypes
import math
import os as _os
import threading as _threading
import weakref
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn as nn

from .. import native

_lib = native.kernels()
_vp, _i, _f, _ll = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_longlong
_ip = ctypes.POINTER(ctypes.c_int)

STAT_SLOTS = 32


def _sig(name: str, argtypes):
    fn = getattr(_lib, name)
    fn.argtypes = argtypes
    fn.restype = ctypes.c_int
    return fn


_gather_gemm = _sig("pdt_gather_gemm", [_vp, _vp, _vp, _vp, _vp, _vp, _vp] + [_i] * 16 + [_ip, _ip, _ip, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp])
_colsum = _sig("pdt_colsum", [_vp, _vp, _i, _i, _i, _vp])
_wgrad = _sig("pdt_wgrad", [_vp, _vp, _vp] + [_i] * 11 + [_ip, _ip, _ip, _i, _i, _i, _vp, _vp, _vp, _ll, _vp])
_bn_stats_rows = _sig("pdt_bn_stats_rows", [_vp, _vp, _vp, _i, _i, _vp])
_lib.pdt_set_deterministic.argtypes = [_i]
_lib.pdt_set_deterministic.restype = None
_lib.pdt_wgrad_group_begin.argtypes = []
_lib.pdt_wgrad_group_begin.restype = None
_wgrad_group_end = _sig("pdt_wgrad_group_end", [_vp])
_bn_finalize = _sig("pdt_bn_finalize", [_vp, _i, _f, _vp, _vp, _vp, _vp, _vp, _f, _f, _i, _vp, _vp, _vp, _vp])
_bn_apply_fwd = _sig("pdt_bn_apply_fwd", [_vp, _vp, _vp, _vp, _vp, _vp, _ll, _i, _i, _vp])
_bn_bwd_reduce = _sig("pdt_bn_bwd_reduce", [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp])
_bn_bwd_finalize = _sig("pdt_bn_bwd_finalize", [_vp, _i, _i, _f, _vp, _vp, _vp, _vp, _vp, _i, _vp])
_bn_bwd_apply = _sig("pdt_bn_bwd_apply", [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _ll, _i, _i, _vp])
_bn_relu_maxpool_fwd = _sig("pdt_bn_relu_maxpool_fwd", [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp])
_maxpool_bn_bwd_apply = _sig("pdt_maxpool_bn_bwd_apply", [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp])
_avgpool_fwd = _sig("pdt_avgpool_fwd", [_vp, _vp, _i, _i, _i, _vp])
_avgpool_bwd = _sig("pdt_avgpool_bwd", [_vp, _vp, _i, _i, _i, _vp])
_ce_fwd = _sig("pdt_softmax_ce_fwd", [_vp, _vp, _vp, _vp, _vp, _i, _i, _ll, _i, _vp])
_ce_bwd = _sig("pdt_softmax_ce_bwd", [_vp, _vp, _vp, _vp, _vp, _i, _i, _ll, _ll, _i, _i, _vp])
_ce_fused = _sig("pdt_softmax_ce_fused", [_vp, _vp, _vp, _vp, _vp, _i, _i, _ll, _ll, _i, _vp])
_ce_fused_chunk = _sig("pdt_softmax_ce_fused_chunk", [_vp, _vp, _vp, _vp, _i, _i, _ll, _ll, _i, _ll, _vp])
_mean_f32 = _sig("pdt_mean_f32", [_vp, _vp, _ll, _f, _vp])
_axpy_dev = _sig("pdt_axpy_dev_scalar_f32", [_vp, _vp, _vp, _ll, _vp])
_scale_by_scalar = _sig("pdt_scale_by_scalar", [_vp, _vp, _ll, _vp])
_nchw_to_nhwc = _sig("pdt_nchw_to_nhwc_bf16", [_vp, _vp, _i, _i, _i, _i, _i, _vp])
_u8_rgb_to_bf16 = _sig("pdt_u8_rgb_to_bf16", [_vp, _vp, _ll, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float), _vp])
_pad_pixels = _sig("pdt_pad_pixels", [_vp, _vp] + [_i] * 8 + [_vp])
_cast_pad = _sig("pdt_cast_pad_f32_bf16", [_vp, _vp, _ll, _i, _i, _vp])
_transpose_w = _sig("pdt_transpose_w", [_vp, _vp, _i, _i, _i, _vp])
_transpose_w_multi = _sig("pdt_transpose_w_multi", [_vp, _i, _i, _vp])
_lib.pdt_transpose_desc_bytes.restype = ctypes.c_int
_mt_step = _sig("pdt_multi_tensor_step", [_vp, _vp, _i, _i] + [_f] * 9 + [_i, _vp])
_mt_step_dev = _sig("pdt_multi_tensor_step_dev", [_vp, _vp, _i, _vp, _vp])
_opt_set_hyper = _sig("pdt_opt_set_hyper", [_vp, _i] + [_f] * 9 + [_i, _vp])
_lib.pdt_set_out_nt_bytes.argtypes = [_ll]
_lib.pdt_set_out_nt_bytes.restype = None
if "PDT_OUT_NT_MB" in _os.environ:  # size from which GEMM outputs are stored non-temporally (tuning knob)
    _lib.pdt_set_out_nt_bytes(int(float(_os.environ["PDT_OUT_NT_MB"]) * (1 << 20)))
_lib.pdt_opt_chunk_elems.restype = ctypes.c_int
_lib.pdt_opt_desc_bytes.restype = ctypes.c_int
_lib.pdt_opt_chunk_bytes.restype = ctypes.c_int


class KernelError(RuntimeError):
    pass


def _check(rc: int, name: str) -> None:
    if rc != 0:
        why = "invalid arguments" if rc == -1 else f"hipError {rc}"
        raise KernelError(f"{name}: {why}")


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream() -> int:
    """Handle of the current HIP stream.  Called once per kernel launch: the raw accessor is ~20x cheaper than
    building a ``torch.cuda.Stream`` object (0.5 ms per step on a launch-bound ResNet-18 step)."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


def _int_array(vals: Sequence[int]):
    return (ctypes.c_int * max(len(vals), 1))(*vals)


_tap_array_cache: Dict[tuple, tuple] = {}


def _tap_arrays(taps):
    """ctypes arrays (dh, dw, weight offset) of a tap list, built once per distinct list."""
    key = tuple(taps)
    hit = _tap_array_cache.get(key)
    if hit is None:
        hit = _tap_array_cache[key] = (_int_array([t[0] for t in key]), _int_array([t[1] for t in key]),
                                       _int_array([t[2] for t in key]))
    return hit


# ------------------------------------------------------------------------------------------------
# deterministic mode
# ------------------------------------------------------------------------------------------------
DETERMINISTIC = False


def set_deterministic(on: bool) -> None:
    """Bit-identical results from run to run (``PDT_DETERMINISTIC=1`` switches it on at import), for telling a real
    divergence between replicas or runs from floating-point noise.  Everything whose result depends on the order of
    floating-point atomic adds takes another route:

    * weight gradients: every pixel-range split of the reduction adds into a zeroed slab of its own and the slabs are
      summed in a fixed order (instead of all splits adding into the gradient with atomics); bias gradients come from
      a column-sum kernel with one adder per column;
    * BatchNorm: forward statistics and backward sums are taken by reduction passes with one writer per accumulator slot
      (instead of in the convolution / data-gradient epilogues with atomics);
    * embedding gradients: tokens are sorted by id and summed per id in order.

    Slower (extra passes over activations, small reduction grids); the arithmetic is otherwise the same."""
    global DETERMINISTIC
    DETERMINISTIC = bool(on)
    _lib.pdt_set_deterministic(int(DETERMINISTIC))


if _os.environ.get("PDT_DETERMINISTIC", "0") not in ("", "0"):
    set_deterministic(True)


# ------------------------------------------------------------------------------------------------
# zero-initialised scratch (statistic accumulators) and direct gradient accumulation
# ------------------------------------------------------------------------------------------------
class ZeroArena:
    """Bump allocator over one fp32 buffer that is cleared with a single memset per training step.

    Kernels that accumulate with atomics (BatchNorm statistic slots) need zeroed outputs; allocating
    each with ``torch.zeros`` costs one fill launch apiece (hundreds per step).  Regions handed out
    since the last :meth:`reset` are zero on first use; ``reset`` (called by the fused optimizer after
    its step, i.e. once the step's autograd graph is gone) clears the used prefix in one launch.
    Requests that do not fit fall back to ``torch.zeros``.
    """

    def __init__(self, capacity_floats: int = 32 * 1024 * 1024):
        self.capacity = capacity_floats
        self.buf: Optional[torch.Tensor] = None
        self.offset = 0
        self.enabled = True
        self._threads: set = set()

    def _check_thread(self) -> None:
        """The arena (like the shadow-weight cache and the padded-gradient cache of this module) is process-wide state
        that assumes ONE training loop: regions are handed out by the thread that runs forward and by autograd's
        device thread that runs backward, strictly one after the other.  A third thread means two loops share the
        process (two models trained concurrently): their steps would clear each other's accumulators."""
        t = _threading.get_ident()
        if t not in self._threads:
            if len(self._threads) >= 2 and not _os.environ.get("PDT_ALLOW_THREADS"):
                raise RuntimeError(
                    "the GPU kernels' step arena is used from a third thread: this package assumes one training loop per "
                    "process (forward on one thread, backward on autograd's device thread); run concurrent models in "
                    "separate processes (PDT_ALLOW_THREADS=1 disables this check)")
            self._threads.add(t)

    def alloc(self, shape, device) -> torch.Tensor:
        self._check_thread()
        n = 1
        for s_ in shape:
            n *= int(s_)
        n_al = _round_up(max(n, 1), 64)
        if self.buf is None and self.enabled:
            self.buf = torch.zeros(self.capacity, dtype=torch.float32, device=device)
            self.offset = 0
        if not self.enabled or self.buf.device != torch.device(device) or self.offset + n_al > self.capacity:
            return torch.zeros(shape, dtype=torch.float32, device=device)
        out = self.buf[self.offset:self.offset + n].view(shape)
        self.offset += n_al
        return out

    def reset(self) -> None:
        if self.buf is not None and self.offset > 0:
            self.buf[: self.offset].zero_()
            self.offset = 0


ARENA = ZeroArena()


def _grad_sink(param: torch.Tensor, phys_shape) -> Optional[torch.Tensor]:
    """fp32 buffer to accumulate ``param``'s gradient into directly (its ``.grad`` in physical element order),
    or ``None`` when the gradient has to be returned to autograd.

    Possible when ``.grad`` already exists with the parameter's dense layout and the parameter was whitelisted
    by :func:`enable_grad_sink`.  The backward then returns ``None`` for the parameter; autograd still runs its
    AccumulateGrad node (adding contributions from other paths, if any) and fires post-accumulate hooks, so
    gradient reducers are notified exactly as for ordinary gradients.
    """
    g = param.grad
    if g is None or g.dtype != torch.float32 or not getattr(param, "_pdt_sink_ok", False) or torch.is_grad_enabled():
        return None
    if g.shape != param.shape or g.stride() != param.stride():
        return None
    if param.dim() == 4:
        view = g.permute(0, 2, 3, 1)
        if not view.is_contiguous() or tuple(view.shape) != tuple(phys_shape):
            return None
        return view
    if not g.is_contiguous() or tuple(g.shape) != tuple(phys_shape):
        return None
    return g


def enable_grad_sink(module: nn.Module) -> int:
    """Allow the kernels to accumulate parameter gradients of ``module`` straight into ``.grad``.

    Every kernel that produces a parameter gradient adds to its destination (atomics into a zeroed buffer),
    so a parameter that is used several times in one forward (tied embeddings) simply receives several
    accumulations; autograd runs the parameter's AccumulateGrad node, and with it any gradient-ready hook,
    only after all of its producers have run, whether they returned a tensor or ``None``.  Gradients from
    operators without a sink (plain PyTorch ops) still arrive through autograd and are added on top.
    Returns the number of parameters enabled.
    """
    n = 0
    for p in module.parameters():
        if p.requires_grad and p.dtype == torch.float32:
            p._pdt_sink_ok = True
            n += 1
    return n


# ------------------------------------------------------------------------------------------------
# layout helpers
# ------------------------------------------------------------------------------------------------
def _as_nhwc(x: torch.Tensor) -> torch.Tensor:
    """bf16, logical NCHW, physically NHWC-contiguous."""
    if x.dtype != torch.bfloat16:
        x = x.to(torch.bfloat16)
    if x.dim() != 4:
        raise ValueError(f"expected a 4-D image tensor, got shape {tuple(x.shape)}")
    if not x.is_contiguous(memory_format=torch.channels_last):
        x = x.contiguous(memory_format=torch.channels_last)
    n, c, h, w = x.shape
    if x.stride() != (h * w * c, 1, w * c, c):  # size-1 dims make channels_last strides ambiguous
        x = torch.as_strided(x, x.shape, (h * w * c, 1, w * c, c), x.storage_offset())
    return x


def _empty_nhwc(n: int, c: int, h: int, w: int, device) -> torch.Tensor:
    return torch.empty((n, h, w, c), dtype=torch.bfloat16, device=device).permute(0, 3, 1, 2)


def _check_size(t: torch.Tensor) -> None:
    """The gather loaders address activations with unsigned 32-bit element offsets (pixel index x channels; everything
    behind them - output rows, BatchNorm / pooling passes, weight-gradient operands - uses 64-bit addresses), so one
    activation tensor may hold up to 2^32 elements (8 GiB of bf16): ResNet-50 up to a batch of 5 300 per GPU."""
    if t.numel() >= 2 ** 32:
        raise ValueError(f"activation of {t.numel()} elements: the kernels address at most 2^32 elements per tensor")


def to_device_layout(x: torch.Tensor, dtype=torch.bfloat16) -> torch.Tensor:
    if dtype != torch.bfloat16:
        raise ValueError("GPU kernels compute in bf16")
    if x.dim() != 4:
        return x.to(dtype)
    if x.dtype == torch.bfloat16 and x.is_contiguous(memory_format=torch.channels_last):
        return x
    if x.dtype not in (torch.float32, torch.bfloat16) or not x.is_contiguous():
        x = x.to(torch.float32).contiguous()
    n, c, h, w = x.shape
    out = _empty_nhwc(n, c, h, w, x.device)
    _check(_nchw_to_nhwc(x.data_ptr(), out.data_ptr(), n, c, h, w, 0 if x.dtype == torch.float32 else 1, _stream()),
           "nchw_to_nhwc")
    return out


def u8_to_device_layout(x: torch.Tensor, mean=None, std=None) -> torch.Tensor:
    """uint8 ``[N, H, W, 3]`` image batch on the GPU -> bf16 NHWC (logical NCHW, channels_last) with
    ``(x / 255 - mean) / std`` per channel (``ToTensor`` alone when ``mean`` / ``std`` are not given)."""
    if x.dtype != torch.uint8 or x.dim() != 4 or x.shape[-1] != 3 or not x.is_contiguous():
        raise ValueError("expected a contiguous uint8 [N, H, W, 3] batch")
    n, h, w, _ = x.shape
    mean = tuple(mean) if mean is not None else (0.0, 0.0, 0.0)
    std = tuple(std) if std is not None else (1.0, 1.0, 1.0)
    scale = (ctypes.c_float * 3)(*[1.0 / (255.0 * s_) for s_ in std])
    shift = (ctypes.c_float * 3)(*[-m / s_ for m, s_ in zip(mean, std)])
    out = torch.empty((n, h, w, 3), dtype=torch.bfloat16, device=x.device)
    _check(_u8_rgb_to_bf16(x.data_ptr(), out.data_ptr(), n * h * w, scale, shift, _stream()), "u8_rgb_to_bf16")
    return out.permute(0, 3, 1, 2)


# ------------------------------------------------------------------------------------------------
# bf16 weight copies
# ------------------------------------------------------------------------------------------------
class _Shadow:
    __slots__ = ("ref", "version", "ptr", "fwd", "fwd_cols", "dgrad", "dgrad_epoch", "fresh_epoch")


class ShadowCache:
    """bf16 copies of fp32 parameters, keyed by parameter identity.

    ``fwd``   - same physical element order as the parameter ([Cout][taps][Cin] for conv weights),
                optionally with zero-padded rows (the stem's K = 147 -> 192).
    ``dgrad`` - transposed to [Cin][taps][Cout] (padded) for the data-gradient GEMM; rebuilt lazily
                after every optimizer step.
    """

    def __init__(self):
        self._entries: Dict[int, _Shadow] = {}
        self.epoch = 0  # bumped by the fused optimizer after it rewrote parameters in place

    def _entry(self, p: torch.Tensor) -> _Shadow:
        e = self._entries.get(id(p))
        if e is None or e.ref() is not p:
            e = _Shadow()
            e.ref = weakref.ref(p)
            e.version = -1
            e.ptr = 0
            e.fwd = e.dgrad = None
            e.fwd_cols = 0
            e.dgrad_epoch = e.fresh_epoch = -1
            self._entries[id(p)] = e
        return e

    @staticmethod
    def physical_matrix(p: torch.Tensor) -> torch.Tensor:
        """fp32 [rows, cols] view (or copy) of the parameter in its physical element order."""
        d = p.detach()
        if d.dim() == 4:
            d = d.permute(0, 2, 3, 1)  # [N, R, S, C]
            if not d.is_contiguous():
                d = d.contiguous()
            return d.reshape(d.shape[0], -1)
        if d.dim() == 2:
            return d if d.is_contiguous() else d.contiguous()
        raise ValueError("weights must be 2-D or 4-D")

    def fwd(self, p: torch.Tensor, pad_cols_to: int = 8) -> torch.Tensor:
        e = self._entry(p)
        stale = (
            e.fwd is None
            or e.version != p._version
            or e.ptr != p.data_ptr()
            or (e.fresh_epoch != self.epoch)
        )
        mat = None
        if stale:
            mat = self.physical_matrix(p)
            cols = _round_up(mat.shape[1], pad_cols_to)
            if e.fwd is None or e.fwd.shape != (mat.shape[0], cols):
                e.fwd = torch.empty((mat.shape[0], cols), dtype=torch.bfloat16, device=p.device)
            _check(_cast_pad(mat.data_ptr(), e.fwd.data_ptr(), mat.shape[0], mat.shape[1], cols, _stream()), "cast_pad")
            e.version, e.ptr, e.fresh_epoch = p._version, p.data_ptr(), self.epoch
        return e.fwd

    def optimizer_target(self, p: torch.Tensor) -> Optional[torch.Tensor]:
        """bf16 buffer the optimizer kernel may refresh in place (same element order, no padding)."""
        e = self._entries.get(id(p))
        if e is None or e.ref() is not p or e.fwd is None or e.fwd.numel() != p.numel():
            return None
        if p.dim() == 4 and not p.permute(0, 2, 3, 1).is_contiguous():
            return None
        if p.dim() != 4 and not p.is_contiguous():
            return None
        return e.fwd

    def mark_refreshed(self, p: torch.Tensor) -> None:
        e = self._entries.get(id(p))
        if e is not None and e.ref() is p:
            e.version, e.ptr, e.fresh_epoch = p._version, p.data_ptr(), self.epoch

    def dgrad(self, p: torch.Tensor, taps: int) -> Tuple[torch.Tensor, int]:
        """([Cin][taps][Npad] bf16, Npad) with Npad = Cout rounded up to 64."""
        e = self._entry(p)
        mat = self.physical_matrix(p)  # [N, taps*C]
        n = mat.shape[0]
        c = mat.shape[1] // taps
        npad = _round_up(n, 64)
        if e.dgrad is None or e.dgrad_epoch != self.epoch or e.version != p._version or e.dgrad.shape != (c, taps * npad):
            if e.dgrad is None or e.dgrad.shape != (c, taps * npad):
                e.dgrad = torch.zeros((c, taps * npad), dtype=torch.bfloat16, device=p.device)
            if npad == n:
                _check(_transpose_w(mat.data_ptr(), e.dgrad.data_ptr(), n, taps, c, _stream()), "transpose_w")
            else:  # rows of the transposed matrix are longer than N: transpose per tap into strided slots
                tmp = torch.empty((c, taps * n), dtype=torch.bfloat16, device=p.device)
                _check(_transpose_w(mat.data_ptr(), tmp.data_ptr(), n, taps, c, _stream()), "transpose_w")
                e.dgrad.view(c, taps, npad)[:, :, :n].copy_(tmp.view(c, taps, n))
            e.dgrad_epoch = self.epoch
            if e.fwd is None:
                e.version = p._version
        return e.dgrad, npad


    def refresh_dgrads(self, plan) -> None:
        """Rebuild, in one launch, the transposed data-gradient copies of the plan's parameters that already exist
        (they were used by the previous backward pass) - right after the optimizer kernel, instead of one small
        transpose launch per layer spread over the next backward pass.  The launch table is cached on the plan
        and rebuilt when a buffer appeared or moved."""
        import numpy as np

        items = []
        for p in plan.params:
            e = self._entries.get(id(p))
            if e is None or e.ref() is not p or e.dgrad is None or p.dim() not in (2, 4):
                continue
            mat = self.physical_matrix(p)
            if mat.data_ptr() != p.data_ptr():
                continue  # the physical order needs a copy: left to the lazy path
            n = mat.shape[0]
            if n % 64 != 0:
                continue  # padded rows: lazy path
            taps = e.dgrad.shape[1] // n
            c = e.dgrad.shape[0]
            if taps * c != mat.shape[1]:
                continue
            items.append((p, e, n, taps, c))
        if not items:
            return
        key = tuple((p.data_ptr(), e.dgrad.data_ptr()) for p, e, *_ in items)
        if getattr(plan, "_tr_key", None) != key:
            assert _lib.pdt_transpose_desc_bytes() == 32
            tab = np.zeros((len(items), 4), dtype=np.int64)
            start = 0
            for i, (p, e, n, taps, c) in enumerate(items):
                tab[i, 0] = p.data_ptr()
                tab[i, 1] = e.dgrad.data_ptr()
                tab[i, 2] = n | (taps << 32)      # int32 N, int32 T
                tab[i, 3] = c | (start << 32)     # int32 C, int32 tile_start
                start += -(-n // 64) * -(-c // 64) * taps
            plan._tr_key = key
            plan._tr_table = torch.from_numpy(tab).to(items[0][0].device)
            plan._tr_tiles = start
        _check(_transpose_w_multi(plan._tr_table.data_ptr(), len(items), plan._tr_tiles, _stream()), "transpose_w_multi")
        for p, e, *_ in items:
            e.dgrad_epoch = self.epoch
            if e.fwd is None:
                e.version = p._version


SHADOWS = ShadowCache()


# ------------------------------------------------------------------------------------------------
# GEMM / convolution launch helpers
# ------------------------------------------------------------------------------------------------
_TILE_WIDE = int(_os.environ.get("PDT_GEMM_TILE_WIDE", "-1"))  # -1: choose per shape
_TILE_NARROW = int(_os.environ.get("PDT_GEMM_TILE_NARROW", "-1"))  # -1: choose per shape
_RING_MIN_TILES = int(_os.environ.get("PDT_GEMM_RING_MIN_TILES", "768"))
# Data-gradient launches that also add a shortcut gradient and take a block tail's BatchNorm sums stream four tensors
# through the epilogue.  They used to run on the 256x128 tile (more register room for operands requested ahead: -1.1 ms
# per ResNet-50 step at the time); since the epilogue's loads are awaited with counted waits and its constants are out
# of the row groups, the tile the shape rule picks is as fast or slightly faster (-0.3 % of the step), so there is no
# special case any more (-1).  A tile id here forces that tile for these launches.
_TILE_FUSED_TAIL = int(_os.environ.get("PDT_GEMM_TILE_FUSED_TAIL", "-1"))
# Single-tap 256x256 launches with at least eight K-steps run on the ping-pong schedule (tile 17: the wave rows of a
# workgroup alternate between MFMA and LDS / DMA phases): +5-15 % on linear layers (bench/pp_check.py).  Multi-tap
# launches (3x3 convolutions) stay on tile 16: their loader's address arithmetic hides better under a wave's own MFMAs
# (ResNet-50 3x3 layers ran 8-12 % slower on the ping-pong schedule), and launches with one or two K-steps per tile are
# bound by their stores either way.  0 switches the schedule off.
_GEMM_PP = int(_os.environ.get("PDT_GEMM_PP", "1"))


_ZERO_PAGES: dict = {}


def _zero_page(device: torch.device) -> torch.Tensor:
    """256 bytes of zeros that nothing ever writes: the source of padding rows for the LDS-DMA loaders."""
    z = _ZERO_PAGES.get(device)
    if z is None:
        z = _ZERO_PAGES[device] = torch.zeros(128, dtype=torch.bfloat16, device=device)
    return z


_NUM_CUS = 0


def _num_cus() -> int:
    global _NUM_CUS
    if _NUM_CUS == 0:
        n = torch.cuda.get_device_properties(torch.cuda.current_device()).multi_processor_count
        k = int(_os.environ.get("PDT_RESERVE_CUS", "0"))  # CUs left to the collectives' kernels (see pdt_num_cus)
        _NUM_CUS = n - k if 0 < k < n else n
    return _NUM_CUS


def _pick_tile(m: int, n: int, ksteps: int = 1) -> int:
    """Tile variant of the gather-GEMM kernel for an M x N output.

    0/1/2: one tile per workgroup (128x128, 128x64, 64x64).  13/14/15/16: persistent workgroups streaming
    256x128 / 256x64 / 512x64 / 256x256 tiles through an LDS-DMA ring; they win once every CU gets several tiles
    (measured per layer with bench/conv_bench.py and bench/gemm_bench.py), below that the tail of a partially
    filled last round costs more than it saves.  The 256x256 tile moves a third less operand data per FLOP
    through the L2-to-LDS path, which is what bounds the 256x128 kernel on large GEMMs."""
    if n <= 64:
        if m < 128:
            return 2
        if _TILE_NARROW >= 0:
            return _TILE_NARROW
        if -(-m // 512) >= _RING_MIN_TILES and ksteps >= 2:
            # 512x64: a wave covers 64 rows, so the weight fragments (re-read by all 8 waves) cost half as many LDS
            # reads per FLOP as in the 256-row tile; 6-9 % on the 64-channel 3x3 and 1x1 layers with several K-steps,
            # slower on single-K-step launches (two ring stages instead of three)
            return 15
        return 14 if -(-m // 256) >= _RING_MIN_TILES else 1  # 256x64 ring once every CU gets several tiles
    if _TILE_WIDE >= 0:
        return _TILE_WIDE
    # Ring kernels hold one workgroup per CU, so a launch costs whole rounds of tiles.  A 256x256 tile takes about
    # 1.65x the time of a 256x128 tile for twice the work; a ring tile is used when its rounds are at least 60 % full,
    # below that the one-tile-per-workgroup kernel (several workgroups per CU) wastes less.  Calibrated with
    # `bench/gemm_bench.py --tile-sweep` on the ResNet-50 and transformer shapes.
    cus = _num_cus()
    t13 = -(-m // 256) * -(-n // 128)
    r13 = -(-t13 // cus)
    if n >= 256 and -(-n // 256) * 256 <= 1.15 * n:
        t16 = -(-m // 256) * -(-n // 256)
        r16 = -(-t16 // cus)
        if t16 >= 0.6 * r16 * cus and 1.65 * r16 <= r13:
            return 16
    if t13 >= _RING_MIN_TILES or t13 >= 0.6 * r13 * cus:
        return 13
    return 0


def _launch_gather_gemm(a, b, out, stats, bias, M, N, ldb, ldo, C, taps, Hin, Win, istr, OA, OB, Hout, Wout, ostr,
                        oph, opw, addend=None, smallc=0, addend_bits=None, bn_stat=None, gelu_out=None, gelu_of=None, xf=None,
                        pivot=None) -> None:
    """``bn_stat = (y, consts, bits_or_None[, y2, consts2])``: also accumulate BatchNorm-backward sums of the result into ``stats``
    ([slots][3][N]); only the ring kernels can do that (see :func:`_dgrad_can_fuse_stats`).
    ``gelu_out``: tensor (layout of ``out``) that receives gelu(result); ``out`` receives gelu'(result), the factor the
    backward pass needs (the pre-activation itself is never stored).
    ``gelu_of``: that derivative tensor (layout of ``out``); the result is multiplied by it.
    ``xf``: [4, C] BatchNorm constants; the A operand is relu(a * scale + shift) per input channel, applied in the
    kernel (single-tap launches on the 256x256 tile only, see :func:`_can_fuse_bn_into_conv`).
    ``pivot``: [N] fp32; the forward statistics in ``stats`` are taken about it (see :func:`_stat_pivot`)."""
    bn_y = bn_consts = bn_bits = bn_y2 = bn_consts2 = None
    if bn_stat is not None:
        bn_y, bn_consts, bn_bits = bn_stat[:3]
        if len(bn_stat) == 5:  # a second BatchNorm shares dz (projected shortcut): its sums go to the third row
            bn_y2, bn_consts2 = bn_stat[3:]
    dh, dw, wo = _tap_arrays(taps)
    det_stats = None
    if DETERMINISTIC and stats is not None and bn_stat is None:
        # forward statistics: not in the epilogue (atomics), but by a pass of their own over the finished output
        if ldo != N or ostr != 1 or oph or opw or OA != Hout or OB != Wout:
            raise NotImplementedError("deterministic statistics need a densely stored output")
        det_stats, stats = stats, None
    tile = _pick_tile(M, N, len(taps) * max(C // 64, 1))
    if _TILE_FUSED_TAIL >= 0 and bn_bits is not None and tile == 16:
        tile = _TILE_FUSED_TAIL
    if tile == 16 and xf is None and _GEMM_PP and (_GEMM_PP > 1 or (len(taps) == 1 and C >= 512)):
        tile = 17
    _check(
        _gather_gemm(a.data_ptr(), b.data_ptr(), out.data_ptr(), _ptr(stats), _ptr(bias), _ptr(addend), _ptr(addend_bits), M, N,
                     ldb, ldo, C, len(taps),
                     Hin, Win, istr, OA, OB, Hout, Wout, ostr, oph, opw, dh, dw, wo,
                     tile, int(smallc),
                     _zero_page(a.device).data_ptr(), _ptr(bn_y), _ptr(bn_consts), _ptr(bn_bits), _ptr(bn_y2), _ptr(bn_consts2),
                     1 if gelu_out is not None else (2 if gelu_of is not None else 0), _ptr(gelu_out), _ptr(gelu_of), _ptr(xf),
                     _ptr(pivot) if stats is not None else None, _stream()),
        "gather_gemm",
    )
    if det_stats is not None:
        _check(_bn_stats_rows(out.data_ptr(), _ptr(pivot), det_stats.data_ptr(), M, N, _stream()), "bn_stats_rows")


WGRAD_TARGET_BLOCKS = int(_os.environ.get("PDT_WGRAD_TARGET_BLOCKS", "640"))
WGRAD_ATOMIC_CAP_MB = int(_os.environ.get("PDT_WGRAD_ATOMIC_CAP_MB", "8"))
WGRAD_MIN_BLOCKS = int(_os.environ.get("PDT_WGRAD_MIN_BLOCKS", "384"))


WGRAD_GROUP = int(_os.environ.get("PDT_WGRAD_GROUP", "1"))  # 0: never pair weight-gradient problems into one launch
WGRAD_RING = int(_os.environ.get("PDT_WGRAD_RING", "1"))  # 0: one-step register-staged kernel everywhere
# ring kernels hold one workgroup per CU: one full round of workgroups, not two (twice the atomic volume)
WGRAD_RING_TARGET_BLOCKS = int(_os.environ.get("PDT_WGRAD_RING_TARGET_BLOCKS", "0"))  # 0: the number of CUs
# Single-tap 256x256 weight gradients run on the ping-pong schedule (tile 17: the two wave rows of a workgroup alternate
# between MFMA and LDS / DMA phases, as in the forward GEMM); 0 switches it off.
WGRAD_PP = int(_os.environ.get("PDT_WGRAD_PP", "1"))
# 3x3 / stride 1 / 64 -> 64 and 128 -> 128 weight gradients (and the stem's) on the kernels that share the input rows
# between the taps (tiles 18, 19); 0: the tap-table ring kernels
WGRAD_HALO = int(_os.environ.get("PDT_WGRAD_HALO", "1"))
_TAPS_3X3 = sorted((i, j) for i in (-1, 0, 1) for j in (-1, 0, 1))


def _wgrad_splits(M: int, N: int, C: int, ntaps: int, bn: int, bc: Optional[int] = None, target: Optional[int] = None) -> int:
    """How many ways to split the pixel reduction across workgroups.

    More splits fill the chip when the weight tile count is small, but every split adds one fp32 atomic
    pass over the whole weight gradient (global float atomics run at roughly a quarter of the store rate),
    so the atomic volume is capped."""
    tiles = math.ceil(N / bn) * math.ceil(C / (bc or bn)) * ntaps
    target = target or WGRAD_TARGET_BLOCKS
    want = max(1, target // max(tiles, 1))
    weight_bytes = 4 * N * C * ntaps
    cap = max(1, (WGRAD_ATOMIC_CAP_MB << 20) // max(weight_bytes, 1))
    floor = min(want, math.ceil(min(WGRAD_MIN_BLOCKS, target) / max(tiles, 1)))
    return max(1, min(max(min(want, cap), floor), M // 256))


# LDS-DMA ring kernels for weight gradients with N >= 128 and >= 128 columns: (tile id, BN, BC, relative MFMA rate
# of the main loop measured on large problems)
_WGRAD_RING_TILES = ((16, 256, 256, 1.0), (11, 256, 128, 0.87), (12, 128, 256, 0.78), (10, 128, 128, 0.6))
# the fp32-atomic epilogue of one workgroup costs about as much as this many 64-pixel K-steps of its main loop
_WGRAD_EPILOGUE_KSTEPS = 25
_wgrad_choice_cache: Dict[Tuple[int, int, int], int] = {}


def _wgrad_ring_choice(M: int, N: int, kc: int) -> int:
    """Tile with the smallest modelled run time: rounds of workgroups x (K-steps + epilogue) x tile area / rate.
    The model reproduces the measured ranking on the transformer and ResNet shapes (bench/gemm_bench.py
    --wgrad-only under PDT_WGRAD_RING=10/11/12/16): large tiles win while their tile count still divides the
    CUs well, small ones when few K-steps are left per workgroup and the epilogue dominates."""
    key = (M, N, kc)
    hit = _wgrad_choice_cache.get(key)
    if hit is not None:
        return hit
    cus = WGRAD_RING_TARGET_BLOCKS or _num_cus()
    best, best_cost = 11, None
    for tile, bn, bc, rate in _WGRAD_RING_TILES:
        tiles = math.ceil(N / bn) * math.ceil(kc / bc)
        splits = _wgrad_splits(M, N, kc, 1, bn, bc, cus)
        rounds = math.ceil(tiles * splits / cus)
        ksteps = math.ceil(math.ceil(M / splits) / 64)
        cost = rounds * (ksteps + _WGRAD_EPILOGUE_KSTEPS) * bn * bc / rate
        if best_cost is None or cost < best_cost:
            best, best_cost = tile, cost
    _wgrad_choice_cache[key] = best
    return best


def _launch_wgrad(dy, x, dw, M, N, ldy, C, taps, ldw, Hin, Win, istr, OA, OB, smallc=0, dbias=None, xf=None) -> bool:
    """Launch the weight-gradient kernel.  ``dbias``: fp32 [N] buffer that should also receive the column sums
    of ``dy`` (the bias gradient); returns True when the chosen kernel did that, False when the caller still
    has to (only the ring kernels can)."""
    kc = (smallc if smallc else C) * len(taps)  # columns of the weight gradient: flattened (tap, channel) axis
    tile = -1
    if WGRAD_RING >= 10:
        tile = WGRAD_RING
    elif WGRAD_RING and smallc:
        tile = 15
    elif WGRAD_RING and N >= 128 and kc >= 128:
        tile = _wgrad_ring_choice(M, N, kc)
    elif WGRAD_RING and N < 128 and kc >= 192:
        tile = 13  # 64 x 256: all output channels of a narrow layer, several taps per tile
    elif WGRAD_RING and N >= 256 and kc < 128:
        tile = 14  # 256 x 64
    if tile == 16 and WGRAD_PP and len(taps) == 1 and xf is None:
        tile = 17  # (the launcher falls back to 16 when the tap is not the identity mapping)
    if (WGRAD_HALO and tile >= 10 and not smallc and len(taps) == 9 and C in (64, 128) and N == C and istr == 1 and OA == Hin
            and OB == Win and Win <= (61 if C == 64 else 29) and (xf is None or C == 64) and dbias is None and M >= 64 * _num_cus()
            and sorted((t[0], t[1]) for t in taps) == _TAPS_3X3):
        # 3x3 / stride 1 / 64 -> 64 (128 -> 128): all columns of 64 output channels in one workgroup, the input rows shared
        # between the taps
        tile = 18
        splits = _num_cus() * 64 // C
    elif (WGRAD_HALO and tile == 15 and smallc == 32 and C == 4 and len(taps) == 7 and N == 64 and istr == 2 and 64 < OB <= 128
          and dbias is None and xf is None and M >= 64 * _num_cus()):
        # the ImageNet stem: the padded input streamed through LDS once, all seven kernel rows read from it
        tile = 19
        splits = _num_cus()
    elif tile >= 10:
        bn, bc = {10: (128, 128), 11: (256, 128), 12: (128, 256), 13: (64, 256), 14: (256, 64), 15: (64, 256),
                  16: (256, 256), 17: (256, 256)}[tile]
        splits = _wgrad_splits(M, N, kc, 1, bn, bc, WGRAD_RING_TARGET_BLOCKS or _num_cus())
    else:
        tile = 0 if (N > 64 and C > 64 and not smallc) else 1
        bt = 128 if tile == 0 else 64
        splits = _wgrad_splits(M, N, smallc * len(taps), 1, bt) if smallc else _wgrad_splits(M, N, C, len(taps), bt)
    dh, dwi, wo = _tap_arrays(taps)
    fuse_bias = dbias is not None and tile >= 10 and tile != 15 and not DETERMINISTIC
    target, stride, slabs = dw, 0, None
    if DETERMINISTIC:
        # one zeroed slab per split of the pixel range: every element of a slab has exactly one adder
        if not dw.is_contiguous():
            raise NotImplementedError("deterministic weight gradients need a densely stored destination")
        stride = dw.numel()
        slabs = torch.zeros((splits, stride), dtype=torch.float32, device=dw.device)
        target = slabs
    _check(
        _wgrad(dy.data_ptr(), x.data_ptr(), target.data_ptr(), M, N, ldy, C, len(taps), ldw, Hin, Win, istr, OA, OB, dh, dwi,
               wo, splits, tile, int(smallc), _zero_page(dy.device).data_ptr(), _ptr(dbias) if fuse_bias else None, _ptr(xf),
               stride, _stream()),
        "wgrad",
    )
    if slabs is not None:  # added in split order by one kernel
        _check(_lib.pdt_sum_slabs(ctypes.c_void_p(dw.data_ptr()), ctypes.c_void_p(slabs.data_ptr()), splits,
                                  ctypes.c_longlong(stride), ctypes.c_longlong(stride), ctypes.c_void_p(_stream())), "sum_slabs")
    return fuse_bias


def _conv_out(h: int, k: int, s: int, p: int) -> int:
    return (h + 2 * p - k) // s + 1


class StatLink:
    """Lets the data-gradient kernel of a convolution accumulate the backward sums of the BatchNorm that produced
    the convolution's input (sum dz, sum dz * xhat per channel), so BatchNorm's backward skips its own
    reduction pass over that gradient.

    The BatchNorm side fills ``y`` / ``consts`` (and ``bits`` for a block tail, whose ReLU mask is stored) in
    forward; the convolution side fills ``slots`` in backward when its kernels support it; BatchNorm's backward,
    which runs next, uses ``slots`` if present and reduces by itself otherwise."""

    __slots__ = ("y", "consts", "bits", "y2", "consts2", "slots", "grad_ptr")

    def __init__(self):
        self.y = self.consts = self.bits = self.y2 = self.consts2 = self.slots = None
        self.grad_ptr = 0  # address of the gradient tensor the sums in ``slots`` were taken over


def _dgrad_can_fuse_stats(launches) -> bool:
    """True when every data-gradient launch (M, N) of a convolution runs on a ring kernel (never in deterministic
    mode: the epilogue's sums are accumulated with atomics)."""
    return not DETERMINISTIC and bool(launches) and all(_pick_tile(m, n) >= 13 for m, n in launches)


class ResidualLink:
    """Carries the shortcut gradient of a residual block from the block-tail backward to the data-gradient
    kernel of the block's first convolution, which adds it in its epilogue.  Both consume the same block
    input ``x``; without the link autograd would materialise both gradients and add them in a separate pass."""

    __slots__ = ("grad", "bits", "armed", "consumed")

    def __init__(self):
        self.grad = None   # gradient tensor to add (the block output's gradient, or the projection's input gradient)
        self.bits = None   # optional ReLU mask bits gating it (then grad * mask is what gets added)
        self.armed = False  # set in forward by the convolution that will consume the gradient
        self.consumed = False  # set in backward once that convolution has run: later producers must not use the link


_PACKED_W: Dict[int, tuple] = {}


def _packed_rows_weight(weight: torch.Tensor, cp: int, tw: int, kpad: int) -> torch.Tensor:
    """bf16 [N][kpad] weights of a few-channel convolution in packed-row order: column kh*tw + kw*cp + c, zeros
    at padding channels, beyond the kernel width and beyond the last kernel row.  Rebuilt once per optimizer step."""
    key = id(weight)
    hit = _PACKED_W.get(key)
    if hit is not None and hit[0]() is weight and hit[1] == (SHADOWS.epoch, weight._version, weight.data_ptr(), cp, tw, kpad):
        return hit[2]
    n, c, r, s = weight.shape
    w = weight.detach().permute(0, 2, 3, 1)  # [n][r][s][c]
    # the buffer of the previous step is reused: its padding positions are zero already, only the live entries change
    buf = hit[2] if (hit is not None and hit[0]() is weight and hit[2].shape == (n, kpad) and hit[1][3:] == (cp, tw, kpad)) \
        else torch.zeros((n, kpad), dtype=torch.bfloat16, device=weight.device)
    buf[:, : r * tw].view(n, r, tw // cp, cp)[:, :, :s, :c].copy_(w)
    import weakref

    _PACKED_W[key] = (weakref.ref(weight), (SHADOWS.epoch, weight._version, weight.data_ptr(), cp, tw, kpad), buf)
    return buf


class _ConvStats(torch.autograd.Function):
    """y = conv2d(x, w) (no bias) and per-channel statistic accumulators of y."""

    @staticmethod
    def forward(ctx, x, weight, stride: int, padding: int, want_stats: bool, link: Optional[ResidualLink] = None,
                link_out: Optional[ResidualLink] = None, stat_link: Optional[StatLink] = None):
        x = _as_nhwc(x)
        nb, c, h, w_ = x.shape
        n, c2, r, s = weight.shape
        if c != c2:
            raise ValueError(f"channel mismatch: input {c}, weight {c2}")
        ho, wo = _conv_out(h, r, stride, padding), _conv_out(w_, s, stride, padding)
        M = nb * ho * wo
        _check_size(x)
        if n % 8 != 0:
            raise ValueError("output channels must be a multiple of 8")
        y = _empty_nhwc(nb, n, ho, wo, x.device)
        _check_size(y)
        stats = ARENA.alloc((STAT_SLOTS + 1, 2, n), x.device) if want_stats else None  # (+ the slot that carries the pivot)
        pivot = _stat_pivot(weight) if want_stats else None
        ctx.geom = (nb, c, h, w_, n, r, s, ho, wo, stride, padding)
        ctx.col = None
        ctx.link = None
        # producer side: usable only if the consuming convolution (called earlier in forward) armed the link
        ctx.link_out = link_out if (link_out is not None and link_out.armed and ctx.needs_input_grad[0]) else None
        ctx.stat_link = stat_link if (stat_link is not None and stat_link.y is not None and c % 64 == 0) else None
        ctx.set_materialize_grads(False)  # the statistics output never carries a gradient: do not zero-fill one
        if link is not None and c % 64 == 0 and stride == 1 and ctx.needs_input_grad[0]:
            link.armed = True
            ctx.link = link
        if c % 64 == 0:
            wb = SHADOWS.fwd(weight)
            taps = [(i - padding, j - padding, (i * s + j) * c) for i in range(r) for j in range(s)]
            _launch_gather_gemm(x, wb, y, stats, None, M, n, r * s * c, n, c, taps, h, w_, stride, ho, wo, ho, wo, 1, 0, 0,
                                pivot=pivot)
            ctx.save_for_backward(x, weight)
        else:
            # few input channels (the stem), packed rows: the image is copied once into a zero-bordered buffer with
            # CP elements per pixel; a kernel row of an output pixel is then `tw` contiguous elements and the
            # convolution is a GEMM with K = r * tw whose operand chunks are plain 16-byte loads (no patch matrix)
            cp = 4 if c <= 4 else 8
            tw = _round_up(s * cp, 32)
            if c > 8 or tw > 64 or r > 49:
                raise NotImplementedError("few-channel convolution: needs Cin <= 8 and a kernel row of at most 64 elements")
            hp, wp = (ho - 1) * stride + r, (wo - 1) * stride + tw // cp
            # one more (zeroed) image row behind the last image: the ring kernel's last K-step addresses the kernel row
            # after the last one (its weights are zero, but the address must be readable and hold finite values)
            xp_rows = torch.empty((nb * hp + 1, wp, cp), dtype=torch.bfloat16, device=x.device)
            xp_rows[-1].zero_()
            xp = xp_rows[: nb * hp].view(nb, hp, wp, cp)
            _check_size(xp_rows)
            _check(_pad_pixels(x.data_ptr(), xp.data_ptr(), nb, h, w_, c, cp, hp, wp, padding, _stream()), "pad_pixels")
            kpad = _round_up(r * tw, 64)
            wb = _packed_rows_weight(weight, cp, tw, kpad)
            _launch_gather_gemm(xp, wb, y, stats, None, M, n, kpad, n, cp, [(0, 0, 0)] * r, hp, wp, stride, ho, wo, ho, wo,
                                1, 0, 0, smallc=tw, pivot=pivot)
            ctx.save_for_backward(xp, weight)
            ctx.col = (kpad, cp, tw, hp, wp)
        if want_stats:
            ctx.mark_non_differentiable(stats)
        return y, stats

    @staticmethod
    def backward(ctx, dy, _dstats):
        saved, weight = ctx.saved_tensors
        nb, c, h, w_, n, r, s, ho, wo, stride, padding = ctx.geom
        if ctx.link is not None:
            ctx.link.consumed = True
        if dy is None:
            return None, None, None, None, None, None, None, None
        dy = _as_nhwc(dy)
        M = nb * ho * wo
        dx = dw = None
        if ctx.needs_input_grad[1]:
            if ctx.col is None:
                sink = _grad_sink(weight, (n, r, s, c))
                dwp = sink if sink is not None else torch.zeros((n, r, s, c), dtype=torch.float32, device=dy.device)
                taps = [(i - padding, j - padding, (i * s + j) * c) for i in range(r) for j in range(s)]
                _launch_wgrad(dy, saved, dwp, M, n, n, c, taps, r * s * c, h, w_, stride, ho, wo)
                if sink is None:
                    dw = dwp.permute(0, 3, 1, 2)
            else:
                kpad, cp, tw, hp, wp = ctx.col
                dwp = torch.zeros((n, kpad), dtype=torch.float32, device=dy.device)
                _launch_wgrad(dy, saved, dwp, M, n, n, cp, [(0, 0, 0)] * r, kpad, hp, wp, stride, ho, wo, smallc=tw)
                real = dwp[:, : r * tw].view(n, r, tw // cp, cp)[:, :, :s, :c]  # [n][r][s][c]: drop the padding positions
                sink = _grad_sink(weight, (n, r, s, c))
                if sink is not None:
                    sink.add_(real)
                else:
                    dw = real.permute(0, 3, 1, 2)
        if ctx.needs_input_grad[0]:
            if ctx.col is not None:
                raise NotImplementedError("data gradient of a convolution with Cin % 64 != 0")
            wt, npad = SHADOWS.dgrad(weight, r * s)
            if npad != n:
                raise NotImplementedError("data gradient needs Cout % 64 == 0")
            dx = _empty_nhwc(nb, c, h, w_, dy.device)
            addend = addend_bits = None
            if ctx.link is not None and ctx.link.grad is not None:
                addend, addend_bits = _as_nhwc(ctx.link.grad), ctx.link.bits
                ctx.link.grad = ctx.link.bits = None
                if addend.shape != dx.shape:
                    raise RuntimeError("residual link carries a gradient of the wrong shape")
            launches = []
            for ph in range(stride):
                oa = (h - ph + stride - 1) // stride
                if oa <= 0:
                    continue
                for pw in range(stride):
                    ob = (w_ - pw + stride - 1) // stride
                    if ob <= 0:
                        continue
                    taps = []
                    for i in range(r):
                        if (ph + padding - i) % stride != 0:
                            continue
                        for j in range(s):
                            if (pw + padding - j) % stride != 0:
                                continue
                            taps.append(((ph + padding - i) // stride, (pw + padding - j) // stride, (i * s + j) * npad))
                    launches.append((oa, ob, ph, pw, taps))
            # BatchNorm-backward sums of dx in the same kernels, when the producer of x asked for them: a block
            # tail (mask bits, comes with the shortcut addend) or a plain BatchNorm + ReLU (mask recomputed)
            bn_stat = slots = None
            sl = ctx.stat_link
            if sl is not None and sl.y is not None and sl.y.shape == dx.shape and \
                    (sl.bits is not None) == (addend is not None) and \
                    _dgrad_can_fuse_stats([(nb * oa * ob, c) for oa, ob, _, _, _ in launches]):
                slots = ARENA.alloc((STAT_SLOTS, 3, c), dy.device)
                bn_stat = (sl.y, sl.consts, sl.bits) if sl.y2 is None else (sl.y, sl.consts, sl.bits, sl.y2, sl.consts2)
            for oa, ob, ph, pw, taps in launches:
                _launch_gather_gemm(dy, wt, dx, slots, None, nb * oa * ob, c, r * s * npad, c, n, taps, ho, wo, 1,
                                    oa, ob, h, w_, stride, ph, pw, addend, addend_bits=addend_bits, bn_stat=bn_stat)
            if bn_stat is not None:
                sl.slots, sl.grad_ptr = slots, dx.data_ptr()
        elif ctx.link is not None and ctx.link.grad is not None:
            raise RuntimeError("residual link armed but the convolution input needs no gradient")
        if dx is not None and ctx.link_out is not None and not ctx.link_out.consumed and ctx.link_out.grad is None:
            # the consumer's data-gradient kernel (it runs later: it was created earlier in forward) adds this
            ctx.link_out.grad, ctx.link_out.bits = dx, None
            dx = None
        return dx, dw, None, None, None, None, None, None


def conv2d_stats(x, weight, stride: int = 1, padding: int = 0, want_stats: bool = True, link=None, link_out=None):
    # a block's first convolution (it owns the residual link, so its data-gradient kernel forms the whole gradient
    # of the block input) also reduces for the BatchNorm of the previous block's tail when that tail left a StatLink
    # on its output tensor
    stat_link = getattr(x, "_pdt_stat_link", None) if link is not None else None
    y, stats = _ConvStats.apply(x, weight, stride, padding, want_stats, link, link_out, stat_link)
    if stats is not None:
        stats._pdt_pivot = _stat_pivot(weight)  # BatchNorm's finalize leaves this batch's mean there for the next step
    return y, stats


STAT_PIVOT = int(_os.environ.get("PDT_STAT_PIVOT", "1"))  # 0: statistics about zero (sum y, sum y^2), for A/B runs


def _stat_pivot(weight: torch.Tensor) -> Optional[torch.Tensor]:
    """Per-output-channel pivot of a convolution's statistics (fp32, device memory, zero at first).

    The convolution's epilogue accumulates sum(y - P) and sum((y - P)^2); with P = 0 the variance BatchNorm derives
    from them is a difference of two large numbers for a channel whose mean is large against its spread.  The buffer
    lives on the weight (one BatchNorm consumes one convolution's statistics), the kernel copies the value it used
    into the statistics tensor, and ``bn_finalize`` writes the new batch mean back: from the second step on the pivot
    tracks the mean.  Device-side state only: nothing is synchronised, a recorded step replays it."""
    if not STAT_PIVOT:
        return None
    pv = getattr(weight, "_pdt_pivot", None)
    if pv is None or pv.device != weight.device or pv.numel() != weight.shape[0]:
        pv = torch.zeros(weight.shape[0], dtype=torch.float32, device=weight.device)
        weight._pdt_pivot = pv
    return pv


_nonfinite_flags: Dict[torch.device, torch.Tensor] = {}


def _nonfinite_flag(device) -> torch.Tensor:
    """One int32 word per device that kernels raise when they meet statistics that are not finite."""
    device = torch.device(device)
    t = _nonfinite_flags.get(device)
    if t is None:
        t = _nonfinite_flags[device] = torch.zeros(1, dtype=torch.int32, device=device)
    return t


def nonfinite_seen(device, clear: bool = True) -> bool:
    """Whether a kernel has seen non-finite batch statistics since the last call (BatchNorm followed by ReLU turns a
    NaN into 0, so the loss alone does not show an Inf / NaN that entered further up).  Synchronises with the device."""
    t = _nonfinite_flags.get(torch.device(device))
    if t is None:
        return False
    seen = bool(int(t.item()))
    if seen and clear:
        t.zero_()
    return seen


def residual_link():
    return ResidualLink()


# ------------------------------------------------------------------------------------------------
# BatchNorm family
# ------------------------------------------------------------------------------------------------
def _bn_consts(stats: Optional[torch.Tensor], bn: nn.BatchNorm2d, count: int, device) -> torch.Tensor:
    """[4, C] fp32: mean, rstd, scale, shift.  Training mode also updates the running statistics."""
    c = bn.num_features
    consts = torch.empty((4, c), dtype=torch.float32, device=device)
    training = bn.training or not bn.track_running_stats
    if training and stats is None:
        raise ValueError("training-mode BatchNorm needs the statistics produced by the convolution")
    if training and tuple(stats.shape) != (STAT_SLOTS + 1, 2, c):
        raise ValueError(f"statistics tensor of shape {tuple(stats.shape)}: expected {(STAT_SLOTS + 1, 2, c)} "
                         "(accumulator slots + the slot that carries the pivot)")
    pivot = getattr(stats, "_pdt_pivot", None) if training else None
    if pivot is not None and (pivot.numel() != c or pivot.device != torch.device(device)):
        pivot = None
    momentum = bn.momentum if bn.momentum is not None else 0.1
    track = bn.track_running_stats and bn.running_mean is not None
    _check(
        _bn_finalize(_ptr(stats), c, float(count), _ptr(bn.weight), _ptr(bn.bias),
                     _ptr(bn.running_mean) if track else None, _ptr(bn.running_var) if track else None,
                     _ptr(bn.num_batches_tracked) if (track and training) else None, momentum, bn.eps,
                     1 if training else 0, consts.data_ptr(), _ptr(pivot), _nonfinite_flag(device).data_ptr(), _stream()),
        "bn_finalize",
    )
    return consts


def _bn_param_grads(slots, which, count, gamma, beta, consts, c, dev):
    """Runs bn_bwd_finalize; dgamma / dbeta go straight into ``gamma.grad`` / ``beta.grad`` when allowed.
    Returns (dgamma | None, dbeta | None, coef)."""
    coef = torch.empty((3, c), dtype=torch.float32, device=dev)
    sg = _grad_sink(gamma, (c,)) if gamma is not None else None
    sb = _grad_sink(beta, (c,)) if beta is not None else None
    if sg is not None and sb is not None:
        _check(_bn_bwd_finalize(slots.data_ptr(), c, which, float(count), _ptr(gamma), consts.data_ptr(), sg.data_ptr(),
                                sb.data_ptr(), coef.data_ptr(), 1, _stream()), "bn_bwd_finalize")
        return None, None, coef
    dgamma = torch.empty(c, dtype=torch.float32, device=dev)
    dbeta = torch.empty(c, dtype=torch.float32, device=dev)
    _check(_bn_bwd_finalize(slots.data_ptr(), c, which, float(count), _ptr(gamma), consts.data_ptr(), dgamma.data_ptr(),
                            dbeta.data_ptr(), coef.data_ptr(), 0, _stream()), "bn_bwd_finalize")
    return dgamma, dbeta, coef


def _bn_backward(dout, out, y, consts, gamma, beta, count, c, y2=None, consts2=None, gamma2=None, beta2=None,
                 want_dz=False, mask_mode=None, slots=None):
    """Shared backward: returns (dy, dgamma, dbeta, extra) where extra = dz | (dy2, dgamma2, dbeta2) | None.

    ``mask_mode``: 0 no ReLU, 1 ReLU mask from the stored output, 2 ReLU mask recomputed from ``y`` (plain
    BN+ReLU: saves reading the output tensor in both passes)."""
    dev = y.device
    M = y.numel() // c
    if mask_mode is None:
        mask_mode = 1 if out is not None else 0
    if slots is None:  # otherwise the kernel that produced dout has already accumulated the sums
        slots = ARENA.alloc((STAT_SLOTS, 3, c), dev)
        _check(_bn_bwd_reduce(dout.data_ptr(), _ptr(out), y.data_ptr(), consts.data_ptr(), _ptr(y2), _ptr(consts2),
                              slots.data_ptr(), M, c, mask_mode, _stream()), "bn_bwd_reduce")
    dgamma, dbeta, coef = _bn_param_grads(slots, 1, count, gamma, beta, consts, c, dev)
    dy = torch.empty_like(y)
    extra = None
    dz = dy2 = coef2 = None
    if y2 is not None:
        dgamma2, dbeta2, coef2 = _bn_param_grads(slots, 2, count, gamma2, beta2, consts2, c, dev)
        dy2 = torch.empty_like(y2)
        extra = (dy2, dgamma2, dbeta2)
    elif want_dz:
        dz = torch.empty_like(y)
        extra = dz
    _check(_bn_bwd_apply(dout.data_ptr(), _ptr(out), y.data_ptr(), coef.data_ptr(), consts.data_ptr(), dy.data_ptr(),
                         _ptr(dz), _ptr(y2), _ptr(coef2), _ptr(dy2), y.numel(), c, mask_mode, _stream()), "bn_bwd_apply")
    return dy, dgamma, dbeta, extra


class _BnAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, stats, gamma, beta, bn, relu: bool, stat_link: Optional[StatLink] = None):
        y = _as_nhwc(y)
        nb, c, h, w = y.shape
        count = nb * h * w
        consts = _bn_consts(stats, bn, count, y.device)
        out = torch.empty_like(y)
        _check(_bn_apply_fwd(y.data_ptr(), consts.data_ptr(), None, None, out.data_ptr(), None, y.numel(), c, int(relu),
                             _stream()), "bn_apply_fwd")
        ctx.save_for_backward(y, consts, gamma, beta)
        ctx.meta = (count, c, relu)
        ctx.stat_link = None
        if stat_link is not None and relu and bn.training and ctx.needs_input_grad[0]:
            stat_link.y, stat_link.consts = y, consts
            ctx.stat_link = stat_link
        return out

    @staticmethod
    def backward(ctx, dout):
        y, consts, gamma, beta = ctx.saved_tensors
        count, c, relu = ctx.meta
        dout = _as_nhwc(dout)
        slots = None
        if ctx.stat_link is not None:
            slots, ctx.stat_link.slots = ctx.stat_link.slots, None
            ctx.stat_link.y = ctx.stat_link.consts = None
        dy, dgamma, dbeta, _ = _bn_backward(dout, None, y, consts, gamma, beta, count, c, mask_mode=2 if relu else 0,
                                            slots=slots)
        return dy, None, dgamma, dbeta, None, None, None


class _BnAddRelu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, stats, gamma, beta, bn, ident, id_stats, id_gamma, id_beta, id_bn, link=None, stat_link=None):
        y = _as_nhwc(y)
        ident = _as_nhwc(ident)
        if ident.shape != y.shape:
            raise ValueError(f"shortcut shape {tuple(ident.shape)} != {tuple(y.shape)}")
        nb, c, h, w = y.shape
        count = nb * h * w
        consts = _bn_consts(stats, bn, count, y.device)
        iconsts = _bn_consts(id_stats, id_bn, count, y.device) if id_bn is not None else None
        out = torch.empty_like(y)
        # one mask bit per element replaces re-reading the whole output tensor in both backward passes
        bits = torch.empty(y.numel() // 8, dtype=torch.uint8, device=y.device) if ctx.needs_input_grad[0] else None
        _check(_bn_apply_fwd(y.data_ptr(), consts.data_ptr(), ident.data_ptr(), _ptr(iconsts), out.data_ptr(), _ptr(bits),
                             y.numel(), c, 1, _stream()), "bn_apply_fwd")
        ctx.projected = id_bn is not None
        ctx.link = link if (link is not None and link.armed and not ctx.projected) else None
        ctx.save_for_backward(y, bits, consts, gamma, beta, ident if ctx.projected else None, iconsts, id_gamma, id_beta)
        ctx.meta = (count, c)
        ctx.stat_link = None
        if stat_link is not None and bits is not None and bn.training:
            # offer (y, consts, mask bits) - and the shortcut BatchNorm's input and constants when the shortcut is
            # projected - to whichever kernel produces this output's gradient
            stat_link.y, stat_link.consts, stat_link.bits = y, consts, bits
            if ctx.projected:
                stat_link.y2, stat_link.consts2 = ident, iconsts
            ctx.stat_link = stat_link
        return out

    @staticmethod
    def backward(ctx, dout):
        y, bits, consts, gamma, beta, ident, iconsts, id_gamma, id_beta = ctx.saved_tensors
        count, c = ctx.meta
        dout = _as_nhwc(dout)
        slots = None
        sl = ctx.stat_link
        if sl is not None:
            # sums taken by the kernel that wrote this very gradient tensor (had autograd added another consumer's
            # gradient, ``dout`` would be a different tensor and the sums would not cover it)
            if sl.slots is not None and sl.grad_ptr == dout.data_ptr():
                slots = sl.slots
            sl.y = sl.consts = sl.bits = sl.y2 = sl.consts2 = sl.slots = None
        if ctx.projected:
            dy, dgamma, dbeta, (dy2, dgamma2, dbeta2) = _bn_backward(dout, bits, y, consts, gamma, beta, count, c, ident,
                                                                      iconsts, id_gamma, id_beta, mask_mode=3, slots=slots)
            return dy, None, dgamma, dbeta, None, dy2, None, dgamma2, dbeta2, None, None, None
        if ctx.link is not None:
            # the shortcut gradient dz = dout * mask is not materialised: the first convolution's data-gradient
            # kernel forms it from (dout, mask bits) while adding it in its epilogue
            dy, dgamma, dbeta, _ = _bn_backward(dout, bits, y, consts, gamma, beta, count, c, mask_mode=3, slots=slots)
            ctx.link.grad, ctx.link.bits = dout, bits
            return dy, None, dgamma, dbeta, None, None, None, None, None, None, None, None
        dy, dgamma, dbeta, dz = _bn_backward(dout, bits, y, consts, gamma, beta, count, c, want_dz=True, mask_mode=3,
                                             slots=slots)
        return dy, None, dgamma, dbeta, None, dz, None, None, None, None, None, None


class _BnReluMaxpool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, stats, gamma, beta, bn):
        y = _as_nhwc(y)
        nb, c, h, w = y.shape
        count = nb * h * w
        consts = _bn_consts(stats, bn, count, y.device)
        ho, wo = _conv_out(h, 3, 2, 1), _conv_out(w, 3, 2, 1)
        out = _empty_nhwc(nb, c, ho, wo, y.device)
        idx = torch.empty((nb, ho, wo, c), dtype=torch.uint8, device=y.device)
        # pre-normalisation value of every window's winner: all that BatchNorm's backward sums need besides dout
        ywin = torch.empty((nb, ho, wo, c), dtype=torch.bfloat16, device=y.device) if ctx.needs_input_grad[0] else None
        _check(_bn_relu_maxpool_fwd(y.data_ptr(), consts.data_ptr(), out.data_ptr(), idx.data_ptr(), _ptr(ywin), nb, h, w, c,
                                    _stream()), "bn_relu_maxpool_fwd")
        ctx.save_for_backward(y, idx, consts, gamma, beta, ywin)
        ctx.meta = (count, c, nb, h, w)
        return out

    @staticmethod
    def backward(ctx, dout):
        y, idx, consts, gamma, beta, ywin = ctx.saved_tensors
        count, c, nb, h, w = ctx.meta
        dout = _as_nhwc(dout)
        dev = y.device
        # The un-pooled gradient dz is never written out.  Its sums over input pixels equal sums over pooled outputs
        # taken at the winners (mask recomputed from the winner's value), a pass over two pooled-size tensors; the
        # apply pass gathers dz from (dout, idx).
        slots = ARENA.alloc((STAT_SLOTS, 3, c), dev)
        _check(_bn_bwd_reduce(dout.data_ptr(), None, ywin.data_ptr(), consts.data_ptr(), None, None, slots.data_ptr(),
                              dout.numel() // c, c, 2, _stream()), "bn_bwd_reduce")
        dgamma, dbeta, coef = _bn_param_grads(slots, 1, count, gamma, beta, consts, c, dev)
        dy = torch.empty_like(y)
        _check(_maxpool_bn_bwd_apply(dout.data_ptr(), idx.data_ptr(), y.data_ptr(), consts.data_ptr(), coef.data_ptr(),
                                     dy.data_ptr(), nb, h, w, c, _stream()), "maxpool_bn_bwd_apply")
        return dy, None, dgamma, dbeta, None


def bn_act(y, stats, bn, relu: bool = True):
    return _BnAct.apply(y, stats, bn.weight, bn.bias, bn, relu, None)


# 0: BatchNorm + ReLU always as a pass of its own; 1: applied inside the 1x1 consumer's kernels where that is measured
# to be no slower (consumers with at most 128 input channels: their GEMMs are bound by the output stores, so the
# transform of the operand tile in LDS hides; with 256+ input channels the launches are bound by LDS / MFMA and
# the transform costs more than the saved pass - bench/bn_fuse_bench.py); 2: wherever the kernels allow (saves the
# most memory).
FUSE_BN_INTO_1X1 = int(_os.environ.get("PDT_FUSE_BN_1X1", "1"))
# 1: BatchNorm + ReLU applied inside the 3x3 / 64 -> 64 consumer's shared-rows kernels instead of as a pass of its own.
# Measured on ResNet-50 at batch 4000 (same box, alternating runs): the pass it removes is 1.7 ms per step, the transform
# costs the forward kernel 0.36 ms and the weight-gradient kernel 0.5 ms, the step gains 0.6-0.9 % and 4.5 GiB less memory
# is held.  0: the stand-alone pass.
FUSE_BN_INTO_3X3 = int(_os.environ.get("PDT_FUSE_BN_3X3", "1"))
_CONV3X3_HALO = _os.environ.get("PDT_CONV3X3_HALO", "1") not in ("", "0")  # (the kernel library reads the same switch)


def _can_fuse_bn_into_conv(y_prev, bn, weight, stride: int, padding: int) -> bool:
    """BatchNorm + ReLU can be applied inside the consuming convolution's kernels (forward GEMM and weight gradient)
    when that convolution is 1x1 / stride 1 (no padding pixels, one tap), training-mode statistics are in use and the
    shapes land on the kernels that carry the transform: the 256x256 two-stage GEMM tile and a ring weight-gradient
    kernel."""
    if not FUSE_BN_INTO_1X1 or not WGRAD_RING or WGRAD_RING >= 10 or _TILE_WIDE >= 0:
        return False
    n, c, r, s = weight.shape
    if (r, s) == (3, 3):
        # 3x3 / stride 1 / 64 -> 64 on the kernels that keep one copy of the input rows in LDS: there the transform is
        # applied once per input element (the tap-table kernels would repeat it for every tap)
        nb, _, h, w = y_prev.shape
        M = nb * h * w
        return bool(FUSE_BN_INTO_3X3 and WGRAD_HALO and _CONV3X3_HALO and stride == 1 and padding == 1 and bn.training and c == 64
                    and n == 64 and 24 <= w <= 61 and M >= 64 * _num_cus() and _pick_tile(M, n, 9) in (14, 15))
    if (r, s) != (1, 1) or stride != 1 or padding != 0 or not bn.training or c % 64 or n % 64 or c > 2048:
        return False
    if FUSE_BN_INTO_1X1 == 1 and c > 128:
        return False
    nb, _, h, w = y_prev.shape
    M = nb * h * w
    if _pick_tile(M, n, c // 64) != 16:
        return False
    return (n >= 128 and c >= 128) or (n >= 256 and c < 128)  # the weight-gradient launch is a ring kernel


class _BnReluConv1x1(torch.autograd.Function):
    """conv(relu(bn(y_prev))) for a 1x1 - or, on the shared-rows kernels, a 3x3 / 64 -> 64 - convolution of stride 1
    without ever writing relu(bn(y_prev)): the forward GEMM and the weight-gradient
    kernel normalise their operand tiles in LDS, the data-gradient kernel takes BatchNorm's backward sums in its
    epilogue (mask recomputed from y_prev), and BatchNorm's backward apply pass produces the gradient of y_prev."""

    @staticmethod
    def forward(ctx, y_prev, stats_prev, gamma, beta, bn, weight):
        y_prev = _as_nhwc(y_prev)
        nb, c, h, w_ = y_prev.shape
        n = weight.shape[0]
        count = M = nb * h * w_
        consts = _bn_consts(stats_prev, bn, count, y_prev.device)
        y = _empty_nhwc(nb, n, h, w_, y_prev.device)
        _check_size(y)
        stats = ARENA.alloc((STAT_SLOTS + 1, 2, n), y_prev.device)
        r = weight.shape[2]  # 1 (no padding) or 3 (padding 1)
        pad = r // 2
        taps = [(i - pad, j - pad, (i * r + j) * c) for i in range(r) for j in range(r)]
        _launch_gather_gemm(y_prev, SHADOWS.fwd(weight), y, stats, None, M, n, r * r * c, n, c, taps, h, w_, 1, h, w_, h, w_, 1, 0, 0,
                            xf=consts, pivot=_stat_pivot(weight))
        ctx.save_for_backward(y_prev, consts, weight, gamma, beta)
        ctx.geom = (nb, c, h, w_, n)
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(stats)
        return y, stats

    @staticmethod
    def backward(ctx, dy, _dstats):
        y_prev, consts, weight, gamma, beta = ctx.saved_tensors
        nb, c, h, w_, n = ctx.geom
        if dy is None:
            return None, None, None, None, None, None
        dy = _as_nhwc(dy)
        count = M = nb * h * w_
        r = weight.shape[2]
        pad = r // 2
        dw = None
        if ctx.needs_input_grad[5]:
            sink = _grad_sink(weight, (n, r, r, c))
            dwp = sink if sink is not None else torch.zeros((n, r, r, c), dtype=torch.float32, device=dy.device)
            taps = [(i - pad, j - pad, (i * r + j) * c) for i in range(r) for j in range(r)]
            _launch_wgrad(dy, y_prev, dwp, M, n, n, c, taps, r * r * c, h, w_, 1, h, w_, xf=consts)
            if sink is None:
                dw = dwp.permute(0, 3, 1, 2)
        dyp = dgamma = dbeta = None
        if ctx.needs_input_grad[0]:
            wt, npad = SHADOWS.dgrad(weight, r * r)
            if npad != n:
                raise NotImplementedError("data gradient needs Cout % 64 == 0")
            da = _empty_nhwc(nb, c, h, w_, dy.device)
            slots = bn_stat = None
            if _dgrad_can_fuse_stats([(M, c)]):
                slots = ARENA.alloc((STAT_SLOTS, 3, c), dy.device)
                bn_stat = (y_prev, consts, None)
            dtaps = [(pad - i, pad - j, (i * r + j) * npad) for i in range(r) for j in range(r)]
            _launch_gather_gemm(dy, wt, da, slots, None, M, c, r * r * npad, c, n, dtaps, h, w_, 1, h, w_, h, w_, 1, 0, 0, bn_stat=bn_stat)
            dyp, dgamma, dbeta, _ = _bn_backward(da, None, y_prev, consts, gamma, beta, count, c, mask_mode=2, slots=slots)
        return dyp, None, dgamma, dbeta, None, dw


def bn_relu_conv2d(y_prev, stats_prev, bn, weight, stride: int = 1, padding: int = 0):
    if _can_fuse_bn_into_conv(y_prev, bn, weight, stride, padding):
        y, stats = _BnReluConv1x1.apply(y_prev, stats_prev, bn.weight, bn.bias, bn, weight)
    else:
        sl = StatLink()  # the convolution's data-gradient kernel also reduces for BatchNorm's backward
        a = _BnAct.apply(y_prev, stats_prev, bn.weight, bn.bias, bn, True, sl)
        y, stats = _ConvStats.apply(a, weight, stride, padding, bn.training, None, None, sl)
    if stats is not None:
        stats._pdt_pivot = _stat_pivot(weight)
    return y, stats


def bn_add_relu(y, stats, bn, identity, id_stats=None, id_bn=None, link=None):
    sl = StatLink()
    if id_bn is not None:
        out = _BnAddRelu.apply(y, stats, bn.weight, bn.bias, bn, identity, id_stats, id_bn.weight, id_bn.bias, id_bn, None, sl)
    else:
        out = _BnAddRelu.apply(y, stats, bn.weight, bn.bias, bn, identity, None, None, None, None, link, sl)
    if sl.y is not None:
        out._pdt_stat_link = sl  # picked up by the next block's first convolution (conv2d_stats)
    return out


def bn_relu_maxpool(y, stats, bn):
    return _BnReluMaxpool.apply(y, stats, bn.weight, bn.bias, bn)


class _GlobalAvgPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a):
        a = _as_nhwc(a)
        nb, c, h, w = a.shape
        out = torch.empty((nb, c), dtype=torch.bfloat16, device=a.device)
        _check(_avgpool_fwd(a.data_ptr(), out.data_ptr(), nb, h * w, c, _stream()), "avgpool_fwd")
        ctx.meta = (nb, c, h, w)
        return out

    @staticmethod
    def backward(ctx, g):
        nb, c, h, w = ctx.meta
        g = g.to(torch.bfloat16).contiguous()
        da = _empty_nhwc(nb, c, h, w, g.device)
        _check(_avgpool_bwd(g.data_ptr(), da.data_ptr(), nb, h * w, c, _stream()), "avgpool_bwd")
        return da


def global_avgpool(a):
    return _GlobalAvgPool.apply(a)


# ------------------------------------------------------------------------------------------------
# Linear
# ------------------------------------------------------------------------------------------------
_PADDED_GRADS: Dict[int, int] = {}  # data_ptr of a zero-padded gradient buffer -> its row stride


class _Linear(torch.autograd.Function):
    """out = x W^T (+ bias) (+ residual), the additions done in the GEMM epilogue."""

    @staticmethod
    def forward(ctx, x, weight, bias, residual=None, link=None):
        n, k = weight.shape
        if k % 64 != 0:
            raise ValueError("in_features must be a multiple of 64")
        lead = x.shape[:-1]
        x2 = x.reshape(-1, k)
        if x2.dtype != torch.bfloat16:
            x2 = x2.to(torch.bfloat16)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        M = x2.shape[0]
        npad = n if n % 64 == 0 else _round_up(n, 64)
        out = torch.empty((M, npad), dtype=torch.bfloat16, device=x.device)
        _check_size(out)
        _check_size(x2)
        wb = SHADOWS.fwd(weight)
        bias_f = bias.detach() if bias is not None else None
        addend = None
        if residual is not None:
            if npad != n or residual.numel() != M * n:
                raise ValueError("a fused residual needs out_features % 64 == 0 and the output's shape")
            addend = residual.reshape(M, n)
            if addend.dtype != torch.bfloat16:
                addend = addend.to(torch.bfloat16)
            if not addend.is_contiguous():
                addend = addend.contiguous()
        _launch_gather_gemm(x2, wb, out, None, bias_f, M, n, k, npad, k, [(0, 0, 0)], 1, 1, 1, 1, 1, 1, 1, 1, 0, 0,
                            addend=addend)
        ctx.save_for_backward(x2, weight)
        ctx.bias = bias
        ctx.meta = (lead, n, k, npad, bias is not None, residual is not None)
        # the skip gradient travels through the link only when the LayerNorm at the head of the branch is going
        # to run its backward (and so collect it)
        ctx.link = link if (link is not None and residual is not None and link.armed) else None
        ctx.res_shape = residual.shape if residual is not None else None
        res = out[:, :n] if npad != n else out
        return res.reshape(*lead, n) if len(lead) != 1 else res

    @staticmethod
    def backward(ctx, g):
        x2, weight = ctx.saved_tensors
        lead, n, k, npad, has_bias, has_res = ctx.meta
        M = x2.shape[0]
        g2 = g.reshape(M, n)
        if g2.dtype != torch.bfloat16:
            g2 = g2.to(torch.bfloat16)
        if npad == n:
            gp = g2 if g2.is_contiguous() else g2.contiguous()
        elif _PADDED_GRADS.pop(g2.data_ptr(), None) == npad and g2.stride() == (npad, 1):
            gp = torch.as_strided(g2, (M, npad), (npad, 1))  # produced zero-padded by the loss kernel
        else:
            gp = torch.zeros((M, npad), dtype=torch.bfloat16, device=g.device)
            gp[:, :n].copy_(g2)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            wt, npad_w = SHADOWS.dgrad(weight, 1)  # [k][npad]
            assert npad_w == npad
            dx = torch.empty((M, k), dtype=torch.bfloat16, device=g.device)
            _launch_gather_gemm(gp, wt, dx, None, None, M, k, npad, k, npad, [(0, 0, 0)], 1, 1, 1, 1, 1, 1, 1, 1, 0, 0)
            dx = dx.reshape(*lead, k)
        # bias gradient: summed by the weight-gradient kernel when it is a ring kernel, else by a column-sum pass
        want_db = has_bias and ctx.needs_input_grad[2]
        db_sink = db_buf = None
        if want_db:
            db_sink = _grad_sink(ctx.bias, (n,))
            db_buf = db_sink if db_sink is not None else torch.zeros(n, dtype=torch.float32, device=g.device)
        bias_done = False
        if ctx.needs_input_grad[1]:
            sink = _grad_sink(weight, (n, k))
            dw = sink if sink is not None else torch.zeros((n, k), dtype=torch.float32, device=g.device)
            bias_done = _launch_wgrad(gp, x2, dw, M, n, npad, k, [(0, 0, 0)], k, 1, 1, 1, 1, 1, dbias=db_buf)
            if sink is not None:
                dw = None
        if want_db:
            if not bias_done:
                _check(_colsum(gp.data_ptr(), db_buf.data_ptr(), M, n, npad, _stream()), "colsum")
            db = None if db_sink is not None else db_buf
        if ctx.link is not None:
            ctx.link.grad = gp.view(ctx.res_shape)  # npad == n here: gp is the contiguous bf16 gradient
            return dx, dw, db, None, None
        return dx, dw, db, (g if has_res else None), None


class _GeluMLP(torch.autograd.Function):
    """out = gelu(x W1^T + b1) W2^T + b2 (+ residual) as four GEMM launches forward+backward-data and two
    weight-gradient launches, with every element-wise step inside a GEMM epilogue: the first GEMM stores the GELU of
    its result and, in place of the result, the GELU's derivative (both from one evaluation of the logistic function),
    the second adds bias and residual, the data-gradient GEMM of the second layer multiplies by the stored derivative
    on the way out (no transcendentals in backward), and the weight-gradient kernels sum the bias gradients."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, residual=None, link=None):
        hd, k = w1.shape
        n = w2.shape[0]
        if k % 64 or hd % 64 or n % 64 or w2.shape[1] != hd:
            raise ValueError("fused MLP needs feature sizes that are multiples of 64")
        lead = x.shape[:-1]
        x2 = x.reshape(-1, k)
        if x2.dtype != torch.bfloat16:
            x2 = x2.to(torch.bfloat16)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        M = x2.shape[0]
        h = torch.empty((M, hd), dtype=torch.bfloat16, device=x.device)
        g = torch.empty_like(h)
        out = torch.empty((M, n), dtype=torch.bfloat16, device=x.device)
        _check_size(h)
        one = [(0, 0, 0)]
        _launch_gather_gemm(x2, SHADOWS.fwd(w1), h, None, b1.detach(), M, hd, k, hd, k, one, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0,
                            gelu_out=g)
        addend = None
        if residual is not None:
            if residual.numel() != M * n:
                raise ValueError("residual must have the output's shape")
            addend = residual.reshape(M, n)
            if addend.dtype != torch.bfloat16:
                addend = addend.to(torch.bfloat16)
            if not addend.is_contiguous():
                addend = addend.contiguous()
        _launch_gather_gemm(g, SHADOWS.fwd(w2), out, None, b2.detach(), M, n, hd, n, hd, one, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0,
                            addend=addend)
        ctx.save_for_backward(x2, h, g, w1, w2)
        ctx.biases = (b1, b2)
        ctx.meta = (lead, M, k, hd, n, residual is not None)
        ctx.link = link if (link is not None and residual is not None and link.armed) else None
        ctx.res_shape = residual.shape if residual is not None else None
        return out.reshape(*lead, n)

    @staticmethod
    def backward(ctx, gout):
        x2, h, g, w1, w2 = ctx.saved_tensors
        b1, b2 = ctx.biases
        lead, M, k, hd, n, has_res = ctx.meta
        dev = gout.device
        gp = gout.reshape(M, n)
        if gp.dtype != torch.bfloat16:
            gp = gp.to(torch.bfloat16)
        if not gp.is_contiguous():
            gp = gp.contiguous()
        one = [(0, 0, 0)]

        def weight_and_bias_grads(dy, inp, w, b, wi, bi, nn_, kk_):
            """Weight gradient (and the bias gradient, by the same kernel when it can) into sinks or fresh buffers."""
            dw = db = None
            want_b = ctx.needs_input_grad[bi]
            bsink = _grad_sink(b, (nn_,)) if want_b else None
            bbuf = (bsink if bsink is not None else torch.zeros(nn_, dtype=torch.float32, device=dev)) if want_b else None
            done = False
            if ctx.needs_input_grad[wi]:
                sink = _grad_sink(w, (nn_, kk_))
                dw = sink if sink is not None else torch.zeros((nn_, kk_), dtype=torch.float32, device=dev)
                done = _launch_wgrad(dy, inp, dw, M, nn_, nn_, kk_, one, kk_, 1, 1, 1, 1, 1, dbias=bbuf)
                if sink is not None:
                    dw = None
            if want_b:
                if not done:
                    _check(_colsum(dy.data_ptr(), bbuf.data_ptr(), M, nn_, nn_, _stream()), "colsum")
                db = None if bsink is not None else bbuf
            return dw, db

        # second layer: dh = (gout W2) * gelu'(pre-activation) (h holds that derivative), dW2 = gout^T g, db2 = colsum(gout)
        w2t, _ = SHADOWS.dgrad(w2, 1)  # [hd][n]
        dh = torch.empty_like(h)
        _launch_gather_gemm(gp, w2t, dh, None, None, M, hd, n, hd, n, one, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, gelu_of=h)
        # both weight gradients in one launch: the second problem's main loops cover the first one's atomic tail
        grouped = WGRAD_GROUP and not DETERMINISTIC  # (deterministic mode reads each launch's slabs right away)
        if grouped:
            _lib.pdt_wgrad_group_begin()
        try:
            dw2, db2 = weight_and_bias_grads(gp, g, w2, b2, 3, 4, n, hd)
            dw1, db1 = weight_and_bias_grads(dh, x2, w1, b1, 1, 2, hd, k)
        finally:
            if grouped:
                _check(_wgrad_group_end(_stream()), "wgrad (grouped)")
        # first layer's input gradient
        dx = None
        if ctx.needs_input_grad[0]:
            w1t, _ = SHADOWS.dgrad(w1, 1)  # [k][hd]
            dx = torch.empty((M, k), dtype=torch.bfloat16, device=dev)
            _launch_gather_gemm(dh, w1t, dx, None, None, M, k, hd, k, hd, one, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0)
            dx = dx.reshape(*lead, k)
        dres = None
        if ctx.link is not None:
            ctx.link.grad = gp.view(ctx.res_shape)
        elif has_res:
            dres = gout
        return dx, dw1, db1, dw2, db2, dres, None


def gelu_mlp(x, w1, b1, w2, b2, residual=None, link=None):
    return _GeluMLP.apply(x, w1, b1, w2, b2, residual, link)


def linear(x, weight, bias=None, activation: Optional[str] = None, residual=None, link=None):
    if activation is not None and residual is not None:
        raise ValueError("an activation and a residual cannot be fused into the same linear layer")
    out = _Linear.apply(x, weight, bias, residual, link)
    if activation == "gelu":
        return gelu(out)
    if activation is not None:
        raise ValueError("activation must be None or 'gelu'")
    return out


# ------------------------------------------------------------------------------------------------
# Softmax cross-entropy
# ------------------------------------------------------------------------------------------------
class _SoftmaxCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels):
        if logits.dim() != 2 or logits.stride(1) != 1:
            logits = logits.reshape(-1, logits.shape[-1]).contiguous()
        if logits.dtype not in (torch.bfloat16, torch.float32):
            logits = logits.float()
        rows, classes = logits.shape
        labels = labels.reshape(-1).to(torch.int64).contiguous()
        if labels.numel() != rows:
            raise ValueError("labels and logits disagree on the number of rows")
        dev = logits.device
        row_loss = torch.empty(rows, dtype=torch.float32, device=dev)
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        ld = logits.stride(0)
        ctx.fused = None
        if logits.dtype == torch.bfloat16 and 1024 < classes <= 65536 and ld % 8 == 0 and ld >= _round_up(classes, 8):
            # language-model vocabularies: the gradient is produced by the forward pass, which holds each row in
            # registers and so reads the logits once; the backward pass only scales it by the incoming gradient
            c8 = _round_up(classes, 8)
            pad_to = ld if ld - classes < 64 else c8
            dbuf = None
            if ctx.needs_input_grad[0]:
                dbuf = torch.empty((rows, pad_to), dtype=torch.bfloat16, device=dev)
            _check(_ce_fused(logits.data_ptr(), labels.data_ptr(), row_loss.data_ptr(), loss.data_ptr(), _ptr(dbuf), rows,
                             classes, ld, pad_to, pad_to, _stream()), "softmax_ce_fused")
            ctx.fused = [dbuf, classes, pad_to]
            return loss.reshape(())
        lse = torch.empty(rows, dtype=torch.float32, device=dev)
        dt = 1 if logits.dtype == torch.bfloat16 else 0
        _check(_ce_fwd(logits.data_ptr(), labels.data_ptr(), row_loss.data_ptr(), lse.data_ptr(), loss.data_ptr(), rows,
                       classes, logits.stride(0), dt, _stream()), "softmax_ce_fwd")
        ctx.save_for_backward(logits, labels, lse)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        if ctx.fused is not None:
            dbuf, classes, pad_to = ctx.fused
            if dbuf is None:
                raise RuntimeError("the fused cross-entropy gradient was already handed out (backward ran twice)")
            ctx.fused[0] = None
            g1 = g.reshape(1).to(torch.float32).contiguous()
            _check(_scale_by_scalar(dbuf.data_ptr(), g1.data_ptr(), dbuf.numel(), _stream()), "scale_by_scalar")
            if pad_to != classes:
                _PADDED_GRADS[dbuf.data_ptr()] = pad_to
                return dbuf[:, :classes], None
            return dbuf, None
        logits, labels, lse = ctx.saved_tensors
        rows, classes = logits.shape
        ld = logits.stride(0)
        pad_to = ld if ld >= classes and ld - classes < 64 else classes
        dbuf = torch.empty((rows, pad_to), dtype=logits.dtype, device=logits.device)
        g1 = g.reshape(1).to(torch.float32).contiguous()
        dt = 1 if logits.dtype == torch.bfloat16 else 0
        _check(_ce_bwd(logits.data_ptr(), labels.data_ptr(), lse.data_ptr(), g1.data_ptr(), dbuf.data_ptr(), rows, classes,
                       ld, pad_to, pad_to, dt, _stream()), "softmax_ce_bwd")
        if pad_to != classes:
            _PADDED_GRADS[dbuf.data_ptr()] = pad_to
            return dbuf[:, :classes], None
        return dbuf, None


def softmax_cross_entropy(logits, labels):
    return _SoftmaxCE.apply(logits, labels)


LM_HEAD_CHUNK_ROWS = int(_os.environ.get("PDT_LM_HEAD_CHUNK_ROWS", "32768"))


class _LMHeadLoss(torch.autograd.Function):
    """mean cross-entropy of ``x W^T`` against ``labels`` - a language model's output projection and loss - without
    ever holding the ``[rows, vocabulary]`` logits (or their gradient) of the whole batch.

    Rows are processed in chunks.  Per chunk: logits GEMM into a scratch buffer -> softmax-cross-entropy kernel that
    turns the buffer into the logits' gradient in place (each row sits in the registers of one workgroup) -> the
    data-gradient and weight-gradient GEMMs consume it.  So the forward pass already produces ``dx`` and ``dW`` for
    an incoming gradient of 1; the backward pass multiplies them by the incoming gradient (a device scalar: no
    host synchronisation) - ``dx`` in place, ``dW`` while adding it to the weight's gradient.  Peak memory is one
    chunk (32768 x 50304 bf16 = 3.3 GB, the largest that 32-bit element offsets allow) instead of two full-batch
    tensors (2 x 3.3 GB at 32 x 1024 tokens, impossible beyond
    32-bit element offsets from 43 sequences on), at the same number of bytes moved."""

    @staticmethod
    def forward(ctx, x, weight, labels):
        n, k = weight.shape
        if k % 64 != 0:
            raise ValueError("in_features must be a multiple of 64")
        lead = x.shape[:-1]
        x2 = x.reshape(-1, k)
        if x2.dtype != torch.bfloat16:
            x2 = x2.to(torch.bfloat16)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        M = x2.shape[0]
        labels = labels.reshape(-1).to(torch.int64).contiguous()
        if labels.numel() != M:
            raise ValueError("labels and activations disagree on the number of rows")
        if not (1024 < n <= 65536):
            raise ValueError("the fused language-model head handles vocabularies of 1025 to 65536 entries")
        dev = x.device
        npad = _round_up(n, 64)
        want_dx, want_dw = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        rows_c = min(M, max(256, LM_HEAD_CHUNK_ROWS))
        scratch = torch.empty((rows_c, npad), dtype=torch.bfloat16, device=dev)
        _check_size(scratch)
        row_loss = torch.empty(M, dtype=torch.float32, device=dev)
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        wb = SHADOWS.fwd(weight)
        dx = torch.empty((M, k), dtype=torch.bfloat16, device=dev) if want_dx else None
        dw = torch.zeros((n, k), dtype=torch.float32, device=dev) if want_dw else None
        wt = SHADOWS.dgrad(weight, 1)[0] if want_dx else None  # [k][npad]
        one = [(0, 0, 0)]
        for r0 in range(0, M, rows_c):
            rows = min(rows_c, M - r0)
            xc = x2[r0:r0 + rows]
            _launch_gather_gemm(xc, wb, scratch, None, None, rows, n, k, npad, k, one, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0)
            grad_buf = scratch if (want_dx or want_dw) else None
            _check(_ce_fused_chunk(scratch.data_ptr(), labels[r0:r0 + rows].data_ptr(), row_loss[r0:r0 + rows].data_ptr(),
                                   _ptr(grad_buf), rows, n, npad, npad, npad, M, _stream()), "softmax_ce_fused_chunk")
            if want_dx:
                _launch_gather_gemm(scratch, wt, dx[r0:r0 + rows], None, None, rows, k, npad, k, npad, one, 1, 1, 1, 1, 1, 1, 1, 1,
                                    0, 0)
            if want_dw:
                _launch_wgrad(scratch, xc, dw, rows, n, npad, k, one, k, 1, 1, 1, 1, 1)
        _check(_mean_f32(row_loss.data_ptr(), loss.data_ptr(), M, 1.0 / M, _stream()), "mean")
        ctx.grads = [dx, dw]
        ctx.weight = weight
        ctx.lead, ctx.k = lead, k
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        dx, dw = ctx.grads
        ctx.grads = [None, None]
        if dx is None and dw is None and (ctx.needs_input_grad[0] or ctx.needs_input_grad[1]):
            raise RuntimeError("the fused language-model head handed its gradients out already (backward ran twice)")
        g1 = g.reshape(1).to(torch.float32).contiguous()
        gx = gw = None
        if dx is not None:
            _check(_scale_by_scalar(dx.data_ptr(), g1.data_ptr(), dx.numel(), _stream()), "scale_by_scalar")
            gx = dx.reshape(*ctx.lead, ctx.k)
        if dw is not None:
            weight = ctx.weight
            sink = _grad_sink(weight, tuple(weight.shape))
            if sink is not None:
                _check(_axpy_dev(sink.data_ptr(), dw.data_ptr(), g1.data_ptr(), dw.numel(), _stream()), "axpy")
            else:
                gw = dw * g1
        return gx, gw, None


def lm_head_loss(x, weight, labels):
    return _LMHeadLoss.apply(x, weight, labels)


# ------------------------------------------------------------------------------------------------
# Multi-tensor optimizer step
# ------------------------------------------------------------------------------------------------
class MultiTensorPlan:
    """Device-resident descriptor and chunk tables for one parameter group."""

    def __init__(self, params: List[torch.Tensor], states: List[List[torch.Tensor]]):
        import numpy as np

        self.keys = [(p.data_ptr(), p.grad.data_ptr() if p.grad is not None else 0) for p in params]
        self.params = params
        dev = params[0].device
        chunk = _lib.pdt_opt_chunk_elems()
        assert _lib.pdt_opt_desc_bytes() == 48 and _lib.pdt_opt_chunk_bytes() == 16
        desc = np.zeros((len(params), 6), dtype=np.int64)
        chunks = []
        self.shadow_params = []
        for i, (p, st) in enumerate(zip(params, states)):
            if p.dtype != torch.float32 or p.grad is None or p.grad.dtype != torch.float32:
                raise TypeError("fused optimizer needs fp32 parameters and gradients")
            if not _same_dense_layout(p, p.grad) or any(not _same_dense_layout(p, s) for s in st):
                raise ValueError("parameter, gradient and state must share one dense layout")
            shadow = SHADOWS.optimizer_target(p)
            if shadow is not None:
                self.shadow_params.append(p)
            desc[i] = [p.data_ptr(), p.grad.data_ptr(), st[0].data_ptr(), st[1].data_ptr() if len(st) > 1 else 0,
                       shadow.data_ptr() if shadow is not None else 0, p.numel()]
            for start in range(0, p.numel(), chunk):
                chunks.append((i, 0, start))
        ch = np.zeros((len(chunks), 2), dtype=np.int64)
        for j, (t, _pad, start) in enumerate(chunks):
            ch[j, 0] = t  # int32 tensor index + int32 padding (little endian)
            ch[j, 1] = start
        self.n_chunks = len(chunks)
        self.desc = torch.from_numpy(desc).to(dev)
        self.chunks = torch.from_numpy(ch).to(dev)

    def matches(self, params: List[torch.Tensor]) -> bool:
        if len(params) != len(self.keys):
            return False
        for p, (pp, gp) in zip(params, self.keys):
            if p.data_ptr() != pp or p.grad is None or p.grad.data_ptr() != gp:
                return False
        return all(SHADOWS.optimizer_target(p) is not None for p in self.shadow_params)


def _same_dense_layout(a: torch.Tensor, b: torch.Tensor) -> bool:
    return a.shape == b.shape and a.stride() == b.stride() and a.numel() * a.element_size() == _span_bytes(a)


def _span_bytes(t: torch.Tensor) -> int:
    if t.numel() == 0:
        return 0
    return (sum((s - 1) * st for s, st in zip(t.shape, t.stride())) + 1) * t.element_size()


_MODE = {"adam": 0, "adamw": 1, "sgd": 2}


def _hyper_args(group):
    t = group["step"]
    b1, b2 = group["betas"]
    bc1 = 1.0 - b1 ** t
    bc2s = math.sqrt(1.0 - b2 ** t)
    return (group["lr"], b1, b2, group["eps"], group["weight_decay"], bc1, bc2s, group["momentum"], group["grad_scale"],
            1 if t == 1 else 0)


def new_hyper_block(device) -> torch.Tensor:
    """Device memory for one parameter group's hyper-parameters (see :func:`upload_hyper`)."""
    return torch.zeros(_lib.pdt_opt_hyper_bytes(), dtype=torch.uint8, device=device)


def upload_hyper(block: torch.Tensor, group) -> None:
    """Stream-ordered write of the group's current hyper-parameters (learning rate, bias corrections for
    ``group["step"]`` ...) into ``block``; a captured optimizer launch reads them from there."""
    _check(_opt_set_hyper(block.data_ptr(), _MODE[group["mode"]], *_hyper_args(group), _stream()), "opt_set_hyper")


def multi_tensor_step(plan: MultiTensorPlan, group, hyper_block: Optional[torch.Tensor] = None) -> None:
    if hyper_block is not None:
        _check(_mt_step_dev(plan.desc.data_ptr(), plan.chunks.data_ptr(), plan.n_chunks, hyper_block.data_ptr(), _stream()),
               "multi_tensor_step")
    else:
        lr, b1, b2, eps, wd, bc1, bc2s, mom, gs, first = _hyper_args(group)
        _check(
            _mt_step(plan.desc.data_ptr(), plan.chunks.data_ptr(), plan.n_chunks, _MODE[group["mode"]], lr, b1, b2, eps, wd,
                     bc1, bc2s, mom, gs, first, _stream()),
            "multi_tensor_step",
        )
    note_parameters_updated(plan)
    SHADOWS.refresh_dgrads(plan)
    ARENA.reset()


def note_parameters_updated(plan: MultiTensorPlan) -> None:
    """Host-side bookkeeping after the update kernel of ``plan`` ran (or was replayed from a hipGraph): every
    derived weight copy is stale except the bf16 copies the kernel itself rewrote."""
    SHADOWS.epoch += 1
    for p in plan.shadow_params:
        SHADOWS.mark_refreshed(p)


# ------------------------------------------------------------------------------------------------
# Transformer operators (LayerNorm, GELU, attention, embedding, bias+residual)
# ------------------------------------------------------------------------------------------------
def layer_norm(x, weight, bias, eps: float = 1e-5, link=None):
    from . import hip_transformer as ht

    return ht.layer_norm(x, weight, bias, eps, link)


def skip_link():
    from . import hip_transformer as ht

    return ht.GradLink()


def gelu(x):
    from . import hip_transformer as ht

    return ht.gelu(x)


def attention(q, k, v, causal: bool = False, scale: Optional[float] = None):
    from . import hip_transformer as ht

    return ht.attention(q, k, v, causal, scale)


def attention_qkv(qkv, heads: int, causal: bool = False):
    from . import hip_transformer as ht

    return ht.attention_qkv(qkv, heads, causal)


def embedding(ids, weight):
    from . import hip_transformer as ht

    return ht.embedding(ids, weight)


def add_bias_residual(x, bias, residual):
    from . import hip_transformer as ht

    return ht.add_bias_residual(x, bias, residual)
