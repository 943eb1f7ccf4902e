"""Plain PyTorch implementations of every fused op.

They define the semantics: the HIP kernels are tested against them, and they are what runs
for CPU tensors (the gloo plumbing configuration and unit tests).  They are ordinary
differentiable compositions, so autograd provides their backward passes.
"""

from __future__ import annotations

import math
from typing import Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F


def batch_norm(y: torch.Tensor, bn: nn.BatchNorm2d) -> torch.Tensor:
    """Training-mode batch statistics (and running-stat update) or eval-mode running statistics."""
    if bn.training and bn.track_running_stats and bn.num_batches_tracked is not None:
        bn.num_batches_tracked.add_(1)
    use_batch = bn.training or not bn.track_running_stats
    return F.batch_norm(
        y,
        bn.running_mean if bn.track_running_stats else None,
        bn.running_var if bn.track_running_stats else None,
        bn.weight,
        bn.bias,
        use_batch,
        bn.momentum if bn.momentum is not None else 0.1,
        bn.eps,
    )


def conv2d_stats(x, weight, stride: int = 1, padding: int = 0, want_stats: bool = True, link=None, link_out=None):
    return F.conv2d(x, weight, None, stride, padding), None


def bn_act(y, stats, bn, relu: bool = True):
    z = batch_norm(y, bn)
    return F.relu(z) if relu else z


def bn_relu_conv2d(y_prev, stats_prev, bn, weight, stride: int = 1, padding: int = 0):
    return F.conv2d(F.relu(batch_norm(y_prev, bn)), weight, None, stride, padding), None


def bn_add_relu(y, stats, bn, identity, id_stats=None, id_bn=None, link=None):
    z = batch_norm(y, bn)
    if id_bn is not None:
        identity = batch_norm(identity, id_bn)
    return F.relu(z + identity)


def bn_relu_maxpool(y, stats, bn):
    return F.max_pool2d(F.relu(batch_norm(y, bn)), kernel_size=3, stride=2, padding=1)


def global_avgpool(a: torch.Tensor) -> torch.Tensor:
    return a.mean(dim=(2, 3))


def linear(x, weight, bias=None, activation: Optional[str] = None, residual=None, link=None):
    out = F.linear(x, weight, bias)
    if residual is not None:
        out = out + residual
    if activation == "gelu":
        out = F.gelu(out, approximate="tanh")
    elif activation is not None:
        raise ValueError("activation must be None or 'gelu'")
    return out


def softmax_cross_entropy(logits: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
    """Mean softmax cross-entropy over all rows (role of ``nn.CrossEntropyLoss()`` at
    /root/reference/src/main.py:62,76)."""
    return F.cross_entropy(logits.float().reshape(-1, logits.shape[-1]), labels.reshape(-1))


def layer_norm(x, weight, bias, eps: float = 1e-5, link=None):
    return F.layer_norm(x, (x.shape[-1],), weight, bias, eps)


def gelu(x):
    return F.gelu(x, approximate="tanh")


def attention(q, k, v, causal: bool = False, scale: Optional[float] = None):
    """softmax(q k^T * scale [+ causal mask]) v for [B, H, S, D] tensors."""
    d = q.shape[-1]
    scale = scale if scale is not None else 1.0 / math.sqrt(d)
    s = torch.matmul(q.float(), k.float().transpose(-1, -2)) * scale
    if causal:
        sq, sk = s.shape[-2], s.shape[-1]
        mask = torch.ones(sq, sk, dtype=torch.bool, device=s.device).tril(sk - sq)
        s = s.masked_fill(~mask, float("-inf"))
    p = torch.softmax(s, dim=-1)
    return torch.matmul(p, v.float()).to(q.dtype)


def attention_qkv(qkv, heads: int, causal: bool = False):
    """Multi-head self-attention on the packed projection output ``[B, S, 3*H*Dh]`` -> ``[B, S, H*Dh]``."""
    b, s, three_d = qkv.shape
    d = three_d // 3
    dh = d // heads
    x = qkv.view(b, s, 3, heads, dh)
    q, k, v = (x[:, :, i].transpose(1, 2) for i in range(3))
    return attention(q, k, v, causal).transpose(1, 2).reshape(b, s, d)


def embedding(ids, weight):
    return F.embedding(ids, weight)


def token_position_embedding(ids, wte, wpe):
    """``wte[ids] + wpe[0..S)`` for ``ids`` of shape ``[B, S]``."""
    s_ = ids.shape[1]
    return F.embedding(ids, wte) + wpe[:s_].unsqueeze(0)


def vit_assemble(tok, cls, pos):
    """``cat([cls, tok], dim=1) + pos``: class token in front of the patch tokens, position table added."""
    b = tok.shape[0]
    return torch.cat([cls.to(tok.dtype).expand(b, -1, -1), tok], dim=1) + pos.to(tok.dtype)


def add_bias_residual(x, bias, residual):
    out = x
    if bias is not None:
        out = out + bias
    if residual is not None:
        out = out + residual
    return out
