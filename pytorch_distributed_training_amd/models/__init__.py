"""Model registry.  ``kind`` tells the trainer how to feed a model: ``image`` models map an
image batch to class logits; ``lm`` models map token ids (+targets) to a loss."""

from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Dict

import torch.nn as nn

from .gpt2 import GPT2, gpt2_medium, gpt2_small, gpt2_tiny
from .resnet import ResNet, resnet18, resnet34, resnet50, resnet101
from .vit import VisionTransformer, vit_b_16, vit_tiny


@dataclass(frozen=True)
class ModelSpec:
    name: str
    kind: str  # "image" | "lm"
    build: Callable[..., nn.Module]
    default_image_size: int = 224
    default_num_classes: int = 1000


MODELS: Dict[str, ModelSpec] = {
    "resnet18": ModelSpec("resnet18", "image", resnet18, 32, 10),
    "resnet34": ModelSpec("resnet34", "image", resnet34),
    "resnet50": ModelSpec("resnet50", "image", resnet50),
    "resnet101": ModelSpec("resnet101", "image", resnet101),
    "vit_b_16": ModelSpec("vit_b_16", "image", vit_b_16),
    "vit_tiny": ModelSpec("vit_tiny", "image", vit_tiny, 32, 10),
    "gpt2_medium": ModelSpec("gpt2_medium", "lm", gpt2_medium),
    "gpt2_small": ModelSpec("gpt2_small", "lm", gpt2_small),
    "gpt2_tiny": ModelSpec("gpt2_tiny", "lm", gpt2_tiny),
}


def build_model(name: str, num_classes: int = 0, image_size: int = 0, **kw) -> nn.Module:
    if name not in MODELS:
        raise KeyError(f"unknown model {name!r}; choose from {sorted(MODELS)}")
    spec = MODELS[name]
    if spec.kind == "lm":
        return spec.build(**kw)
    args = dict(num_classes=num_classes or spec.default_num_classes)
    if name.startswith("vit"):
        args["image_size"] = image_size or spec.default_image_size
    args.update(kw)
    return spec.build(**args)
