"""GPT-2 (medium: 24 layers, d=1024, 16 heads, vocab 50257, context 1024) on the fused operators.

Not present in the reference (/root/reference/src/main.py:49 builds a ResNet-18 only); it is
the large-gradient data-parallel workload (~355 M parameters, tied input/output
embeddings) that stresses the bucket reducer and the fused AdamW step.
The forward pass returns logits; with ``targets`` it returns the mean next-token
cross-entropy from the fused output-projection + loss operator, which works through the rows
in chunks and never materialises logits or probabilities for the whole batch.
"""

from __future__ import annotations

import math
from typing import Optional

import torch
import torch.nn as nn

from .. import ops
from .vit import EncoderBlock


class GPT2(nn.Module):
    def __init__(self, vocab_size: int = 50257, n_ctx: int = 1024, dim: int = 1024, depth: int = 24, heads: int = 16):
        super().__init__()
        self.vocab_size, self.n_ctx, self.dim = vocab_size, n_ctx, dim
        self.wte = nn.Embedding(vocab_size, dim)
        self.wpe = nn.Embedding(n_ctx, dim)
        self.h = nn.ModuleList([EncoderBlock(dim, heads, 4 * dim, causal=True) for _ in range(depth)])
        for blk in self.h:
            blk.ln_1.eps = blk.ln_2.eps = 1e-5
        self.ln_f = nn.LayerNorm(dim, eps=1e-5)
        self.apply(self._init)
        for blk in self.h:  # GPT-2 scales residual projections by 1/sqrt(2 * depth)
            nn.init.normal_(blk.attn.proj.weight, std=0.02 / math.sqrt(2 * depth))
            nn.init.normal_(blk.mlp.fc2.weight, std=0.02 / math.sqrt(2 * depth))

    @staticmethod
    def _init(m):
        if isinstance(m, nn.Linear):
            nn.init.normal_(m.weight, std=0.02)
            if m.bias is not None:
                nn.init.zeros_(m.bias)
        elif isinstance(m, nn.Embedding):
            nn.init.normal_(m.weight, std=0.02)

    def forward(self, idx: torch.Tensor, targets: Optional[torch.Tensor] = None) -> torch.Tensor:
        b, s = idx.shape
        if s > self.n_ctx:
            raise ValueError(f"sequence length {s} exceeds context {self.n_ctx}")
        x = ops.token_position_embedding(idx, self.wte.weight, self.wpe.weight)
        for blk in self.h:
            x = blk(x)
        x = ops.layer_norm(x, self.ln_f.weight, self.ln_f.bias, self.ln_f.eps)
        if targets is None:
            return ops.linear(x, self.wte.weight, None)  # tied output projection
        # projection + loss in one operator: the [b * s, vocabulary] logits are never held as a whole
        return ops.lm_head_loss(x, self.wte.weight, targets)


def gpt2_medium(vocab_size: int = 50257, n_ctx: int = 1024) -> GPT2:
    return GPT2(vocab_size, n_ctx, 1024, 24, 16)


def gpt2_small(vocab_size: int = 50257, n_ctx: int = 1024) -> GPT2:
    return GPT2(vocab_size, n_ctx, 768, 12, 12)


def gpt2_tiny(vocab_size: int = 512, n_ctx: int = 64) -> GPT2:
    """Small configuration for tests."""
    return GPT2(vocab_size, n_ctx, 128, 2, 2)
