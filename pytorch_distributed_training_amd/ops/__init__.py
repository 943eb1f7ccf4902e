"""Fused operators.

Every op has exactly one GPU implementation - a hand-written gfx950 HIP kernel wrapped in
an ``autograd.Function`` (``ops.hip``) - and a plain PyTorch definition (``ops.reference``)
that states its semantics, runs for CPU tensors and is the oracle in numerics tests.
Dispatch is by device only: a GPU tensor never takes the PyTorch composition, and a missing
kernel library is an error, not a fallback.
"""

from __future__ import annotations

from typing import Optional

import torch

from . import reference as _ref


import os as _os

# 0: projection and loss as two operators with full-batch logits (A/B measurements; caps the batch at 42 x 1024 tokens)
_LM_HEAD_FUSED = _os.environ.get("PDT_LM_HEAD_FUSED", "1") != "0"


def _hip():
    from . import hip  # imported lazily: loading it maps the kernel library

    return hip


def residual_link(x):
    """Token shared by the first convolution of a residual block and the other consumer of the block input
    ``x`` (the block tail, or the projection convolution of a down-sampling block): on GPU it lets the
    shortcut's gradient be added inside the first convolution's data-gradient kernel instead of in a separate
    pass.  ``None`` (no effect) for CPU tensors."""
    return _hip().residual_link() if x.is_cuda else None


def conv2d_stats(x, weight, stride: int = 1, padding: int = 0, want_stats: bool = True, link=None, link_out=None):
    """Convolution without bias.  Returns ``(y, stats)`` where ``stats`` carries the per-channel
    batch statistics of ``y`` gathered in the kernel epilogue (``None`` on CPU or when not wanted,
    e.g. in inference where BatchNorm uses its running statistics).

    ``link`` / ``link_out`` (see :func:`residual_link`): with ``link`` this convolution's data-gradient kernel
    adds whatever gradient the link carries; with ``link_out`` this convolution hands its input gradient to the
    link instead of returning it to autograd."""
    if x.is_cuda:
        return _hip().conv2d_stats(x, weight, stride, padding, want_stats, link, link_out)
    return _ref.conv2d_stats(x, weight, stride, padding)


def bn_act(y, stats, bn, relu: bool = True):
    """BatchNorm (+ReLU) of a raw convolution output."""
    if y.is_cuda:
        return _hip().bn_act(y, stats, bn, relu)
    return _ref.bn_act(y, stats, bn, relu)


def bn_relu_conv2d(y_prev, stats_prev, bn, weight, stride: int = 1, padding: int = 0):
    """conv(relu(bn(y_prev))) from a raw convolution output and its batch statistics.  On GPU the normalisation is
    applied inside the convolution's kernels for 1x1 consumers with at most 128 input channels (no normalised
    activation is written) and as one streaming pass in front of the convolution otherwise (``ops.hip.FUSE_BN_INTO_1X1``)."""
    if y_prev.is_cuda:
        return _hip().bn_relu_conv2d(y_prev, stats_prev, bn, weight, stride, padding)
    return _ref.bn_relu_conv2d(y_prev, stats_prev, bn, weight, stride, padding)


def bn_add_relu(y, stats, bn, identity, id_stats=None, id_bn=None, link=None):
    """relu(bn(y) + identity), or relu(bn(y) + id_bn(identity)) for a projected shortcut."""
    if y.is_cuda:
        return _hip().bn_add_relu(y, stats, bn, identity, id_stats, id_bn, link)
    return _ref.bn_add_relu(y, stats, bn, identity, id_stats, id_bn)


def bn_relu_maxpool(y, stats, bn):
    """Stem tail: maxpool3x3/2(relu(bn(y)))."""
    if y.is_cuda:
        return _hip().bn_relu_maxpool(y, stats, bn)
    return _ref.bn_relu_maxpool(y, stats, bn)


def global_avgpool(a):
    if a.is_cuda:
        return _hip().global_avgpool(a)
    return _ref.global_avgpool(a)


def skip_link(x):
    """Token shared by the LayerNorm at the head of a pre-norm residual branch and the linear layer that closes
    the branch with ``residual=x``: ``linear(f(layer_norm(x, ..., link=L)), ..., residual=x, link=L)``.  On GPU
    the skip connection's gradient is then added inside the LayerNorm backward kernel instead of by a separate
    element-wise pass.  Both ops must see the same ``x``.  ``None`` (no effect) for CPU tensors."""
    return _hip().skip_link() if x.is_cuda else None


def linear(x, weight, bias=None, activation: Optional[str] = None, residual=None, link=None):
    """act(x W^T + bias) or x W^T + bias + residual (the residual is added in the GEMM epilogue on GPU).
    ``link``: see :func:`skip_link`."""
    if x.is_cuda:
        return _hip().linear(x, weight, bias, activation, residual, link)
    return _ref.linear(x, weight, bias, activation, residual)


def gelu_mlp(x, w1, b1, w2, b2, residual=None, link=None):
    """gelu(x W1^T + b1) W2^T + b2 (+ residual): the transformer feed-forward block.  On GPU every element-wise
    step (bias, GELU, residual, GELU's derivative, bias gradients) runs inside a GEMM epilogue."""
    if x.is_cuda and b1 is not None and b2 is not None and not (w1.shape[0] % 64 or w1.shape[1] % 64 or w2.shape[0] % 64):
        return _hip().gelu_mlp(x, w1, b1, w2, b2, residual, link)
    return linear(linear(x, w1, b1, activation="gelu"), w2, b2, residual=residual, link=link)


def softmax_cross_entropy(logits, labels):
    if logits.is_cuda:
        return _hip().softmax_cross_entropy(logits, labels)
    return _ref.softmax_cross_entropy(logits, labels)


def lm_head_loss(x, weight, labels):
    """Mean cross-entropy of the logits ``x W^T`` (no bias) against ``labels``: a language model's tied output
    projection and loss.  On GPU the rows are processed in chunks through one scratch buffer, so the
    ``[rows, vocabulary]`` logits of the whole batch never exist (see ``ops.hip._LMHeadLoss``)."""
    if x.is_cuda and 1024 < weight.shape[0] <= 65536 and _LM_HEAD_FUSED:
        return _hip().lm_head_loss(x, weight, labels)
    logits = linear(x, weight, None)
    return softmax_cross_entropy(logits.reshape(-1, logits.shape[-1]), labels.reshape(-1))


def layer_norm(x, weight, bias, eps: float = 1e-5, link=None):
    if x.is_cuda:
        return _hip().layer_norm(x, weight, bias, eps, link)
    return _ref.layer_norm(x, weight, bias, eps)


def gelu(x):
    if x.is_cuda:
        return _hip().gelu(x)
    return _ref.gelu(x)


def attention(q, k, v, causal: bool = False, scale: Optional[float] = None):
    if q.is_cuda:
        return _hip().attention(q, k, v, causal, scale)
    return _ref.attention(q, k, v, causal, scale)


def attention_qkv(qkv, heads: int, causal: bool = False):
    """Self-attention on the packed QKV projection ``[B, S, 3*H*Dh]`` -> ``[B, S, H*Dh]``."""
    if qkv.is_cuda:
        return _hip().attention_qkv(qkv, heads, causal)
    return _ref.attention_qkv(qkv, heads, causal)


def embedding(ids, weight):
    if weight.is_cuda:
        return _hip().embedding(ids, weight)
    return _ref.embedding(ids, weight)


def token_position_embedding(ids, wte, wpe):
    """Input embedding of a language model: ``wte[ids] + wpe[position]`` for ``ids`` of shape ``[B, S]``."""
    if wte.is_cuda:
        from . import hip_transformer as ht

        return ht.token_position_embedding(ids, wte, wpe)
    return _ref.token_position_embedding(ids, wte, wpe)


def vit_assemble(tok, cls, pos):
    """Vision-transformer token sequence: class token in front of the patch tokens, plus the position table."""
    if tok.is_cuda:
        from . import hip_transformer as ht

        return ht.vit_assemble(tok, cls, pos)
    return _ref.vit_assemble(tok, cls, pos)


def add_bias_residual(x, bias, residual):
    if x.is_cuda:
        return _hip().add_bias_residual(x, bias, residual)
    return _ref.add_bias_residual(x, bias, residual)


def nonfinite_seen(device, clear: bool = True) -> bool:
    """Whether a GPU kernel met non-finite batch statistics since the last call (always False on the CPU path, where
    PyTorch's own operators propagate NaN / Inf into the loss)."""
    if torch.device(device).type == "cuda":
        return _hip().nonfinite_seen(device, clear)
    return False


def set_deterministic(on: bool) -> None:
    """Run-to-run bit-identical GPU results (see ``ops.hip.set_deterministic``; ``PDT_DETERMINISTIC=1`` does the same at
    import).  The CPU path is PyTorch's own operators and is not affected."""
    if torch.cuda.is_available():
        _hip().set_deterministic(on)


def prepare_module(module):
    """One-time per-model setup of the GPU path (no-op for CPU modules): lets weight-gradient kernels
    accumulate directly into ``.grad`` where that is safe."""
    p = next(module.parameters(), None)
    if p is not None and p.is_cuda:
        _hip().enable_grad_sink(module)
    return module


def to_device_layout(x, dtype=torch.bfloat16):
    """NCHW float batch -> compute dtype in the layout the GPU kernels read (NHWC / channels_last)."""
    if x.is_cuda:
        return _hip().to_device_layout(x, dtype)
    return x.to(dtype)


def image_batch_to_device_layout(x, dtype=torch.bfloat16, mean=None, std=None):
    """uint8 ``[N, H, W, 3]`` batch (the form images travel in) -> float batch the models read: normalised
    ``(x / 255 - mean) / std``, logical NCHW.  On GPU one kernel produces bf16 NHWC; on CPU plain PyTorch."""
    if x.is_cuda:
        if dtype != torch.bfloat16:
            raise ValueError("GPU kernels compute in bf16")
        return _hip().u8_to_device_layout(x, mean, std)
    out = x.permute(0, 3, 1, 2).to(torch.float32).contiguous().div_(255.0)  # plain NCHW, as ToTensor + default collation give
    if mean is not None or std is not None:
        m = torch.tensor(mean or (0.0, 0.0, 0.0), dtype=torch.float32).view(1, 3, 1, 1)
        s_ = torch.tensor(std or (1.0, 1.0, 1.0), dtype=torch.float32).view(1, 3, 1, 1)
        out = (out - m) / s_
    return out.to(dtype)
