"""Vision Transformer (ViT-B/16 and friends) on the fused operators.

Not present in the reference (its only model is ResNet-18,
/root/reference/src/main.py:49); it is the GEMM/attention-heavy data-parallel workload.
Architecture follows the standard ViT: 16x16 patch embedding, class token, learned position
embedding, pre-LayerNorm encoder blocks (MHA + GELU MLP), final LayerNorm, linear head on
the class token.  The patch embedding is expressed as a GEMM over unfolded patches so it
runs on the same MFMA kernel as every other linear layer.
"""

from __future__ import annotations

import math

import torch
import torch.nn as nn

from .. import ops


class MLP(nn.Module):
    def __init__(self, dim: int, hidden: int):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)

    def forward(self, x, residual, link=None):
        return ops.gelu_mlp(x, self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias, residual=residual, link=link)


class SelfAttention(nn.Module):
    def __init__(self, dim: int, heads: int, causal: bool = False):
        super().__init__()
        assert dim % heads == 0
        self.dim, self.heads, self.causal = dim, heads, causal
        self.qkv = nn.Linear(dim, 3 * dim)
        self.proj = nn.Linear(dim, dim)

    def forward(self, x, residual, link=None):
        qkv = ops.linear(x, self.qkv.weight, self.qkv.bias)  # [b, s, 3*d], consumed in place by the attention kernel
        o = ops.attention_qkv(qkv, self.heads, causal=self.causal)
        return ops.linear(o, self.proj.weight, self.proj.bias, residual=residual, link=link)


class EncoderBlock(nn.Module):
    def __init__(self, dim: int, heads: int, mlp_dim: int, causal: bool = False):
        super().__init__()
        self.ln_1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = SelfAttention(dim, heads, causal)
        self.ln_2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = MLP(dim, mlp_dim)

    def forward(self, x):
        # each branch's skip gradient is handed to the LayerNorm at its head (ops.skip_link)
        l1 = ops.skip_link(x)
        x = self.attn(ops.layer_norm(x, self.ln_1.weight, self.ln_1.bias, self.ln_1.eps, link=l1), x, l1)
        l2 = ops.skip_link(x)
        return self.mlp(ops.layer_norm(x, self.ln_2.weight, self.ln_2.bias, self.ln_2.eps, link=l2), x, l2)


class VisionTransformer(nn.Module):
    def __init__(self, image_size: int = 224, patch: int = 16, dim: int = 768, depth: int = 12, heads: int = 12,
                 mlp_dim: int = 3072, num_classes: int = 1000):
        super().__init__()
        assert image_size % patch == 0
        self.patch, self.dim = patch, dim
        self.n_tokens = (image_size // patch) ** 2 + 1
        self.patch_embed = nn.Linear(3 * patch * patch, dim)  # == Conv2d(3, dim, patch, patch) on unfolded patches
        self.class_token = nn.Parameter(torch.zeros(1, 1, dim))
        self.pos_embedding = nn.Parameter(torch.empty(1, self.n_tokens, dim).normal_(std=0.02))
        self.layers = nn.ModuleList([EncoderBlock(dim, heads, mlp_dim) for _ in range(depth)])
        self.ln = nn.LayerNorm(dim, eps=1e-6)
        self.head = nn.Linear(dim, num_classes)
        self.apply(self._init)
        nn.init.zeros_(self.head.weight)
        nn.init.zeros_(self.head.bias)

    @staticmethod
    def _init(m):
        if isinstance(m, nn.Linear):
            nn.init.trunc_normal_(m.weight, std=0.02)
            if m.bias is not None:
                nn.init.zeros_(m.bias)

    def _patches(self, x: torch.Tensor) -> torch.Tensor:
        b, c, hh, ww = x.shape
        p = self.patch
        x = x.reshape(b, c, hh // p, p, ww // p, p).permute(0, 2, 4, 1, 3, 5)  # b, gh, gw, c, p, p
        return x.reshape(b, (hh // p) * (ww // p), c * p * p)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        tok = ops.linear(self._patches(x), self.patch_embed.weight, self.patch_embed.bias)
        h = ops.vit_assemble(tok, self.class_token, self.pos_embedding)
        for blk in self.layers:
            h = blk(h)
        h = ops.layer_norm(h, self.ln.weight, self.ln.bias, self.ln.eps)
        return ops.linear(h[:, 0].contiguous(), self.head.weight, self.head.bias)


def vit_b_16(num_classes: int = 1000, image_size: int = 224) -> VisionTransformer:
    return VisionTransformer(image_size, 16, 768, 12, 12, 3072, num_classes)


def vit_tiny(num_classes: int = 10, image_size: int = 32) -> VisionTransformer:
    """Small configuration for tests."""
    return VisionTransformer(image_size, 8, 128, 2, 2, 256, num_classes)
