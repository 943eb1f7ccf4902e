"""GPU transformer operators: LayerNorm, GELU, fused attention, embedding, bias+residual.

Thin ``autograd.Function`` wrappers over ``csrc/kernels/transformer.hip`` and
``csrc/kernels/attention.hip``.  Activations are bf16, parameters and their gradients fp32.
Not exercised by the reference (its model is a CNN, /root/reference/src/main.py:49).
"""

from __future__ import annotations

import ctypes
import math
from typing import Optional

import torch

from .hip import _check, _colsum, _grad_sink, _lib, _ptr, _stream

_vp, _i, _f, _ll = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_longlong


def _sig(name, argtypes):
    fn = getattr(_lib, name)
    fn.argtypes = argtypes
    fn.restype = ctypes.c_int
    return fn


_ln_fwd = _sig("pdt_layernorm_fwd", [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _vp])
_ln_bwd = _sig("pdt_layernorm_bwd", [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp])
_ln_bwd_parts = _sig("pdt_layernorm_bwd_parts", [_i, _i])
_gelu_fwd = _sig("pdt_gelu_fwd", [_vp, _vp, _ll, _vp])
_gelu_bwd = _sig("pdt_gelu_bwd", [_vp, _vp, _vp, _ll, _vp])
_bias_res = _sig("pdt_bias_residual", [_vp, _vp, _vp, _vp, _ll, _i, _vp])
_emb_fwd = _sig("pdt_embedding_fwd", [_vp, _vp, _vp, _ll, _i, _ll, _vp])
_emb_bwd = _sig("pdt_embedding_bwd", [_vp, _vp, _vp, _ll, _i, _ll, _vp])
_emb_pos_fwd = _sig("pdt_embedding_pos_fwd", [_vp, _vp, _vp, _vp, _ll, _i, _ll, _i, _vp])
_batch_rowsum = _sig("pdt_batch_rowsum", [_vp, _vp, _i, _i, _i, _i, _i, _vp])
_vit_asm_fwd = _sig("pdt_vit_assemble_fwd", [_vp, _vp, _vp, _vp, _i, _i, _i, _vp])
_vit_asm_bwd = _sig("pdt_vit_assemble_bwd", [_vp, _vp, _i, _i, _i, _vp])
_attn_fwd = _sig("pdt_attention_fwd", [_vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _vp])
_attn_bwd = _sig("pdt_attention_bwd", [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _vp])


def _bf16c(x: torch.Tensor) -> torch.Tensor:
    if x.dtype != torch.bfloat16:
        x = x.to(torch.bfloat16)
    return x if x.is_contiguous() else x.contiguous()


class GradLink:
    """Carries the gradient of a skip connection around a normalised branch to the LayerNorm at the branch's
    head: ``y = linear(f(layer_norm(x, link=L)), residual=x, link=L)``.  The linear's backward leaves the
    gradient of ``residual`` (its incoming gradient, unchanged) here instead of returning it, and the
    LayerNorm backward - which always runs later, being upstream of the linear - adds it to its own input
    gradient in the same kernel, so autograd never has to sum the two with a separate pass.  ``armed`` is set
    by the LayerNorm forward only when its backward is going to run."""

    __slots__ = ("grad", "armed")

    def __init__(self):
        self.grad = None
        self.armed = False


class _LayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, eps: float, link=None):
        ctx.link = link
        if link is not None:
            link.armed = bool(ctx.needs_input_grad[0])
            link.grad = None
        x = _bf16c(x)
        d = x.shape[-1]
        rows = x.numel() // d
        y = torch.empty_like(x)
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
        _check(_ln_fwd(x.data_ptr(), weight.data_ptr(), bias.data_ptr(), y.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                       rows, d, eps, _stream()), "layernorm_fwd")
        ctx.save_for_backward(x, weight, mean, rstd)
        ctx.bias = bias
        return y

    @staticmethod
    def backward(ctx, g):
        x, weight, mean, rstd = ctx.saved_tensors
        g = _bf16c(g)
        d = x.shape[-1]
        rows = x.numel() // d
        dx = torch.empty_like(x)
        # the kernel adds its column sums to its destinations: the parameters' .grad when that is allowed
        sw, sb = _grad_sink(weight, (d,)), _grad_sink(ctx.bias, (d,))
        if sw is not None and sb is not None:
            dg, db = sw, sb
        else:
            sw = sb = None
            dg, db = torch.zeros((2, d), dtype=torch.float32, device=x.device).unbind(0)
        addend = None
        if ctx.link is not None and ctx.link.grad is not None:
            addend = ctx.link.grad
            ctx.link.grad = None
            if addend.shape != x.shape or addend.dtype != torch.bfloat16 or not addend.is_contiguous():
                raise RuntimeError("the linked skip gradient does not match the LayerNorm input")
        partial = torch.empty((_ln_bwd_parts(rows, d), 2 * d), dtype=torch.float32, device=x.device)
        _check(_ln_bwd(g.data_ptr(), x.data_ptr(), weight.data_ptr(), mean.data_ptr(), rstd.data_ptr(), dx.data_ptr(),
                       dg.data_ptr(), db.data_ptr(), rows, d, _ptr(addend), partial.data_ptr(), _stream()), "layernorm_bwd")
        if sw is not None:
            return dx, None, None, None, None
        return dx, dg, db, None, None


LAYERNORM_MAX_WIDTH = 4096  # the kernels keep a row in the registers of one wave: 8 sixteen-byte vectors per lane


def layer_norm(x, weight, bias, eps: float = 1e-5, link=None):
    d = x.shape[-1]
    if d > LAYERNORM_MAX_WIDTH or d % 8:
        # wider (or oddly sized) rows than the register-resident kernels hold: PyTorch's operator, in fp32, with the
        # skip connection's gradient (normally added inside the backward kernel) left to autograd
        if link is not None:
            raise ValueError(f"layer_norm over {d} features cannot take a fused skip-link: the HIP kernels handle rows of "
                             f"up to {LAYERNORM_MAX_WIDTH} features (multiples of 8); call it without `link`")
        return torch.nn.functional.layer_norm(x.float(), (d,), weight.float(), None if bias is None else bias.float(),
                                              eps).to(x.dtype)
    return _LayerNorm.apply(x, weight, bias, eps, link)


class _Gelu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _bf16c(x)
        y = torch.empty_like(x)
        _check(_gelu_fwd(x.data_ptr(), y.data_ptr(), x.numel(), _stream()), "gelu_fwd")
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        g = _bf16c(g)
        dx = torch.empty_like(x)
        _check(_gelu_bwd(g.data_ptr(), x.data_ptr(), dx.data_ptr(), x.numel(), _stream()), "gelu_bwd")
        return dx


def gelu(x):
    return _Gelu.apply(x)


class _BiasResidual(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, bias, residual):
        x = _bf16c(x)
        d = x.shape[-1]
        res = _bf16c(residual) if residual is not None else None
        out = torch.empty_like(x)
        _check(_bias_res(x.data_ptr(), _ptr(bias), _ptr(res), out.data_ptr(), x.numel(), d, _stream()), "bias_residual")
        ctx.meta = (bias is not None, residual is not None, d)
        ctx.bias = bias
        return out

    @staticmethod
    def backward(ctx, g):
        has_bias, has_res, d = ctx.meta
        db = None
        if has_bias and ctx.needs_input_grad[1]:
            g2 = _bf16c(g)
            rows = g2.numel() // d
            sink = _grad_sink(ctx.bias, (d,))
            db = sink if sink is not None else torch.zeros(d, dtype=torch.float32, device=g.device)
            _check(_colsum(g2.data_ptr(), db.data_ptr(), rows, d, d, _stream()), "colsum")
            if sink is not None:
                db = None
        return g, db, (g if has_res else None)


def add_bias_residual(x, bias, residual):
    if bias is None and residual is None:
        return x
    return _BiasResidual.apply(x, bias, residual)


class _Embedding(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ids, weight):
        ids_c = ids.reshape(-1).to(torch.int64).contiguous()
        vocab, d = weight.shape
        w = weight if weight.is_contiguous() else weight.contiguous()
        out = torch.empty((ids_c.numel(), d), dtype=torch.bfloat16, device=weight.device)
        _check(_emb_fwd(ids_c.data_ptr(), w.data_ptr(), out.data_ptr(), ids_c.numel(), d, vocab, _stream()), "embedding_fwd")
        ctx.save_for_backward(ids_c)
        ctx.weight = weight
        ctx.meta = (vocab, d)
        return out.reshape(*ids.shape, d)

    @staticmethod
    def backward(ctx, g):
        (ids_c,) = ctx.saved_tensors
        vocab, d = ctx.meta
        g2 = _bf16c(g).reshape(-1, d)
        sink = _grad_sink(ctx.weight, (vocab, d))  # no [vocab, d] temporary and no dense add when allowed
        dw = sink if sink is not None else torch.zeros((vocab, d), dtype=torch.float32, device=g.device)
        _embedding_grad(ids_c, g2, dw, d, vocab)
        return None, (None if sink is not None else dw)


def embedding(ids, weight):
    return _Embedding.apply(ids, weight)


def _param_grad_target(param, shape, device):
    """(buffer to accumulate a parameter's fp32 gradient into, whether it is the parameter's own ``.grad``)."""
    sink = _grad_sink(param, shape)
    if sink is not None:
        return sink, True
    return torch.zeros(shape, dtype=torch.float32, device=device), False


class _TokenPositionEmbedding(torch.autograd.Function):
    """``wte[ids] + wpe[position]`` -> bf16 ``[B, S, D]`` in one kernel; the position table's gradient is a sum over
    the batch taken by a kernel that owns each (position, channel) - no broadcast add, no reduction pass."""

    @staticmethod
    def forward(ctx, ids, wte, wpe):
        b, s = ids.shape
        ids_c = ids.reshape(-1).to(torch.int64).contiguous()
        vocab, d = wte.shape
        if wpe.shape[0] < s or wpe.shape[1] != d or not wte.is_contiguous() or not wpe.is_contiguous():
            raise ValueError("position table too short or of a different width")
        out = torch.empty((b, s, d), dtype=torch.bfloat16, device=wte.device)
        _check(_emb_pos_fwd(ids_c.data_ptr(), wte.data_ptr(), wpe.data_ptr(), out.data_ptr(), ids_c.numel(), d, vocab, s,
                            _stream()), "embedding_pos_fwd")
        ctx.save_for_backward(ids_c)
        ctx.tables = (wte, wpe)
        ctx.meta = (b, s, d, vocab)
        return out

    @staticmethod
    def backward(ctx, g):
        (ids_c,) = ctx.saved_tensors
        wte, wpe = ctx.tables
        b, s, d, vocab = ctx.meta
        g2 = _bf16c(g).reshape(b, s, d)
        dwte = dwpe = None
        if ctx.needs_input_grad[1]:
            buf, own = _param_grad_target(wte, (vocab, d), g.device)
            _embedding_grad(ids_c, g2.reshape(-1, d), buf, d, vocab)
            dwte = None if own else buf
        if ctx.needs_input_grad[2]:
            buf, own = _param_grad_target(wpe, tuple(wpe.shape), g.device)
            _check(_batch_rowsum(g2.data_ptr(), buf.data_ptr(), b, s, d, 0, s, _stream()), "batch_rowsum")
            dwpe = None if own else buf
        return None, dwte, dwpe


def _embedding_grad(ids_c, g2, buf, d: int, vocab: int) -> None:
    """``buf[ids[i]] += g2[i]``.  Normally a scatter-add with fp32 atomics; in deterministic mode the tokens are sorted
    by id (stable) and each id's rows are summed in that order, so the result does not depend on arrival order."""
    from . import hip as _h

    if not _h.DETERMINISTIC:
        _check(_emb_bwd(ids_c.data_ptr(), g2.data_ptr(), buf.data_ptr(), ids_c.numel(), d, vocab, _stream()), "embedding_bwd")
        return
    ok = (ids_c >= 0) & (ids_c < vocab)
    ids_v, order = torch.sort(torch.where(ok, ids_c, torch.full_like(ids_c, vocab)), stable=True)
    n_ok = int(ok.sum())
    ids_v, order = ids_v[:n_ok], order[:n_ok]
    if n_ok == 0:
        return
    uniq, counts = torch.unique_consecutive(ids_v, return_counts=True)
    sums = torch.segment_reduce(g2.index_select(0, order).float(), "sum", lengths=counts, axis=0)
    buf.index_add_(0, uniq, sums)  # distinct rows: one adder per element


def token_position_embedding(ids, wte, wpe):
    return _TokenPositionEmbedding.apply(ids, wte, wpe)


class _VitAssemble(torch.autograd.Function):
    """``cat([cls, tok], 1) + pos`` -> bf16 ``[B, T + 1, D]`` in one kernel (class token and position table are fp32
    parameters); backward: one copy kernel for the patch tokens' gradient, batch sums for the two parameters."""

    @staticmethod
    def forward(ctx, tok, cls, pos):
        tok = _bf16c(tok)
        b, t, d = tok.shape
        if cls.numel() != d or pos.numel() != (t + 1) * d or not cls.is_contiguous() or not pos.is_contiguous():
            raise ValueError("class token / position table do not match the token grid")
        out = torch.empty((b, t + 1, d), dtype=torch.bfloat16, device=tok.device)
        _check(_vit_asm_fwd(tok.data_ptr(), cls.data_ptr(), pos.data_ptr(), out.data_ptr(), b, t, d, _stream()), "vit_assemble_fwd")
        ctx.params = (cls, pos)
        ctx.meta = (b, t, d)
        return out

    @staticmethod
    def backward(ctx, g):
        cls, pos = ctx.params
        b, t, d = ctx.meta
        g2 = _bf16c(g).reshape(b, t + 1, d)
        dtok = dcls = dpos = None
        if ctx.needs_input_grad[0]:
            dtok = torch.empty((b, t, d), dtype=torch.bfloat16, device=g.device)
            _check(_vit_asm_bwd(g2.data_ptr(), dtok.data_ptr(), b, t, d, _stream()), "vit_assemble_bwd")
        if ctx.needs_input_grad[1]:
            buf, own = _param_grad_target(cls, tuple(cls.shape), g.device)
            _check(_batch_rowsum(g2.data_ptr(), buf.data_ptr(), b, t + 1, d, 0, 1, _stream()), "batch_rowsum")
            dcls = None if own else buf
        if ctx.needs_input_grad[2]:
            buf, own = _param_grad_target(pos, tuple(pos.shape), g.device)
            _check(_batch_rowsum(g2.data_ptr(), buf.data_ptr(), b, t + 1, d, 0, t + 1, _stream()), "batch_rowsum")
            dpos = None if own else buf
        return dtok, dcls, dpos


def vit_assemble(tok, cls, pos):
    return _VitAssemble.apply(tok, cls, pos)


class _AttentionQKV(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv, heads: int, causal: bool):
        qkv = _bf16c(qkv)
        b, s, three_d = qkv.shape
        d = three_d // 3
        dh = d // heads
        if dh != 64 or heads * dh * 3 != three_d:
            raise ValueError("the attention kernel needs head_dim == 64")
        out = torch.empty((b, s, d), dtype=torch.bfloat16, device=qkv.device)
        lse = torch.empty((b, heads, s), dtype=torch.float32, device=qkv.device)
        scale = 1.0 / math.sqrt(dh)
        _check(_attn_fwd(qkv.data_ptr(), out.data_ptr(), lse.data_ptr(), b, s, heads, dh, scale, int(causal), _stream()),
               "attention_fwd")
        ctx.save_for_backward(qkv, out, lse)
        ctx.meta = (b, s, heads, dh, scale, causal)
        return out

    @staticmethod
    def backward(ctx, g):
        qkv, out, lse = ctx.saved_tensors
        b, s, heads, dh, scale, causal = ctx.meta
        g = _bf16c(g)
        dqkv = torch.empty_like(qkv)
        delta = torch.empty((b, heads, s), dtype=torch.float32, device=qkv.device)
        _check(_attn_bwd(qkv.data_ptr(), out.data_ptr(), g.data_ptr(), lse.data_ptr(), delta.data_ptr(), dqkv.data_ptr(), b, s,
                         heads, dh, scale, int(causal), _stream()), "attention_bwd")
        return dqkv, None, None


def attention_qkv(qkv, heads: int, causal: bool = False):
    return _AttentionQKV.apply(qkv, heads, causal)


def attention(q, k, v, causal: bool = False, scale: Optional[float] = None):
    """[B, H, S, 64] q/k/v -> [B, H, S, 64]; packs into the layout the fused kernel reads."""
    b, h, s, dh = q.shape
    if scale is not None and abs(scale - 1.0 / math.sqrt(dh)) > 1e-9:
        raise ValueError("the fused attention kernel uses scale = 1/sqrt(head_dim)")
    qkv = torch.stack([t.transpose(1, 2) for t in (q, k, v)], dim=2).reshape(b, s, 3 * h * dh)
    return attention_qkv(qkv, h, causal).view(b, s, h, dh).transpose(1, 2)
