"""Whole-step hipGraph capture for launch-bound configurations.

A ResNet-18 step on CIFAR-sized images with the reference's batch of 32
(/root/reference/src/main.py:22,49,68-79) is a few hundred kernels of a few microseconds each: the
GPU waits for the host to launch them.  :class:`GraphedStep` records one complete optimisation step
(zero-grad, forward, loss, backward, gradient all-reduce launches, optimizer update) into a hipGraph
and replays it per batch with one host call.

What makes the step recordable:

* every kernel of this package is launched on ``torch.cuda.current_stream()``, which is the capture
  stream inside ``torch.cuda.graph``; scratch comes from the caching allocator's graph-private pool or
  from the step arena (fixed addresses), so replays touch the same memory;
* gradients live permanently in one flat buffer (or the reducer's buckets), so "zero the gradients" is
  a recorded memset and parameter gradients are accumulated in place by the kernels;
* the optimizer reads its hyper-parameters from device memory (``FusedOptimizer.set_capturable``), so
  the learning rate and Adam's bias corrections advance although the recorded launch never changes;
* the reducer's bucket launches are stream/event operations (record on the compute stream, wait +
  all-reduce on the communication stream, wait back), which a capture records as graph dependencies.

Inputs are copied into static device tensors before each replay; the loss is read from a static
tensor.  Batches whose shape differs from the captured one (the last, smaller batch of an epoch) must
be run eagerly by the caller.
"""

from __future__ import annotations

from typing import Callable, Sequence, Tuple

import torch


class GraphedStep:
    def __init__(self, body: Callable[..., torch.Tensor], optimizer, example: Sequence[torch.Tensor], warmup: int = 0):
        """``body(*inputs)`` runs one full eager step (including ``optimizer.step()`` and the gradient
        clearing) on device tensors and returns the detached loss.  ``example`` fixes shapes and dtypes.

        At least one eager step of the same shape must have run before the capture (here, ``warmup`` of them
        on the example batch, or earlier by the caller): it builds the optimizer's launch plan and the bf16
        weight copies with host-side work that cannot be recorded."""
        if not example or not all(t.is_cuda for t in example):
            raise ValueError("graph capture needs device-resident example inputs")
        self.optimizer = optimizer
        self.static_in: Tuple[torch.Tensor, ...] = tuple(t.clone() for t in example)
        self.signature = tuple((tuple(t.shape), t.dtype) for t in example)
        optimizer.set_capturable(True)
        # eager steps in capturable mode: allocate the hyper-parameter blocks, warm every cache the step
        # touches (bf16 weight copies, launch plans, kernel attributes) so the capture records steady state
        for _ in range(warmup):
            body(*self.static_in)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.static_loss = body(*self.static_in)
        self.replays = 0

    def matches(self, inputs: Sequence[torch.Tensor]) -> bool:
        return len(inputs) == len(self.signature) and all(
            t.is_cuda and tuple(t.shape) == s and t.dtype == d for t, (s, d) in zip(inputs, self.signature)
        )

    def __call__(self, *inputs: torch.Tensor) -> torch.Tensor:
        for dst, src in zip(self.static_in, inputs):
            if dst.data_ptr() != src.data_ptr():
                dst.copy_(src, non_blocking=True)
        self.optimizer.advance()
        self.graph.replay()
        self.optimizer.notify_replayed()
        self.replays += 1
        return self.static_loss
