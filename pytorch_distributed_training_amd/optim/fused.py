"""Multi-tensor fused optimizers: Adam (L2), AdamW (decoupled decay) and SGD-momentum.

The reference uses ``torch.optim.Adam(lr, weight_decay)`` - L2 decay folded into the
gradient, not AdamW (/root/reference/src/main.py:63,79) - so ``mode="adam"`` reproduces
exactly that update; ``adamw`` and ``sgd`` are the additional modes.

On GPU one kernel launch updates every parameter tensor of a group: a device-resident
descriptor table lists (param, grad, state, bf16 shadow) pointers per tensor and a chunk
table maps thread blocks to tensor slices (``csrc/kernels/optim.hip``).  The same launch
refreshes the bf16 compute copies of the weights that the convolution/GEMM kernels read,
so no separate cast pass exists.  CPU tensors take the ``torch._foreach`` implementation of
the same arithmetic below.
"""

from __future__ import annotations

import math
from typing import Dict, Iterable, List, Optional

import torch
from torch.optim import Optimizer

MODES = ("adam", "adamw", "sgd")


class FusedOptimizer(Optimizer):
    def __init__(
        self,
        params: Iterable,
        mode: str = "adam",
        lr: float = 1e-3,
        betas=(0.9, 0.999),
        eps: float = 1e-8,
        weight_decay: float = 0.0,
        momentum: float = 0.9,
        grad_scale: float = 1.0,
    ):
        if mode not in MODES:
            raise ValueError(f"mode must be one of {MODES}")
        if lr < 0 or eps < 0 or weight_decay < 0 or not (0 <= momentum < 1):
            raise ValueError("invalid hyper-parameter")
        defaults = dict(mode=mode, lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay,
                        momentum=momentum, grad_scale=grad_scale)
        super().__init__(params, defaults)
        self._plans: Dict[int, object] = {}  # per param-group GPU launch plans
        self._flat_grads: Dict[int, tuple] = {}  # per param-group (flat gradient buffer, view pointers)
        self._hyper_blocks: Dict[int, torch.Tensor] = {}  # capturable mode: device-resident hyper-parameters
        self.capturable = False

    # ------------------------------------------------------------------ state
    def _init_state(self, group) -> None:
        for p in group["params"]:
            if not p.requires_grad:
                continue
            st = self.state[p]
            if len(st) == 0:
                if group["mode"] == "sgd":
                    st["momentum_buffer"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                else:
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        group.setdefault("step", 0)

    # ------------------------------------------------------------------ hipGraph support
    def set_capturable(self, on: bool = True) -> None:
        """Make ``step()`` recordable in a hipGraph (GPU groups only).

        A recorded kernel launch replays with the argument values of capture time, which would freeze the
        learning rate and Adam's bias corrections.  In capturable mode those live in a small device block per
        group: :meth:`advance` (outside the graph, before every replay) bumps the step count and writes the
        current values with a stream-ordered kernel; ``step()`` inside a capture only records the update
        kernel, which reads the block.  Outside a capture ``step()`` does both, so eager use is unchanged."""
        self.capturable = bool(on)
        if self.capturable:
            from ..ops import hip

            for gi, group in enumerate(self.param_groups):
                params = [p for p in group["params"] if p.requires_grad]
                if params and params[0].is_cuda and gi not in self._hyper_blocks:
                    self._hyper_blocks[gi] = hip.new_hyper_block(params[0].device)

    def advance(self) -> None:
        """Capturable mode: count one step and publish its hyper-parameters to the device."""
        from ..ops import hip

        for gi, group in enumerate(self.param_groups):
            self._init_state(group)
            params = [p for p in group["params"] if p.grad is not None]
            if not params or not params[0].is_cuda:
                continue
            group["step"] += 1
            block = self._hyper_blocks.get(gi)
            if block is None:
                block = self._hyper_blocks[gi] = hip.new_hyper_block(params[0].device)
            hip.upload_hyper(block, group)

    def notify_replayed(self) -> None:
        """Capturable mode: a hipGraph that contains this optimizer's update kernels has just been replayed.
        The replay ran no host code, so redo the host-side bookkeeping of a step (which derived weight copies
        are fresh) for code that runs eagerly afterwards."""
        from ..ops import hip

        for plan in self._plans.values():
            hip.note_parameters_updated(plan)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        capturing = self.capturable and torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()
        if self.capturable and not capturing:
            self.advance()
        for gi, group in enumerate(self.param_groups):
            self._init_state(group)
            params = [p for p in group["params"] if p.grad is not None]
            if not params:
                continue
            if params[0].is_cuda:
                if not self.capturable:
                    group["step"] += 1
                elif capturing and self._plans.get(gi) is None:
                    raise RuntimeError("capturable optimizer: run at least one eager step before capturing "
                                       "(the launch plan is built with host-to-device copies)")
                self._step_hip(gi, group, params)
            else:
                group["step"] += 1
                self._step_foreach(group, params)
        return loss

    # ------------------------------------------------------------------ CPU arithmetic (the definition)
    def _step_foreach(self, group, params: List[torch.Tensor]) -> None:
        lr, wd, mode = group["lr"], group["weight_decay"], group["mode"]
        grads = [p.grad for p in params]
        if group["grad_scale"] != 1.0:
            grads = torch._foreach_mul(grads, group["grad_scale"])
        if mode == "sgd":
            if wd != 0:
                grads = torch._foreach_add(grads, params, alpha=wd)
            bufs = [self.state[p]["momentum_buffer"] for p in params]
            mu = group["momentum"]
            if mu != 0:
                if group["step"] == 1:
                    for b, g in zip(bufs, grads):
                        b.copy_(g)
                else:
                    torch._foreach_mul_(bufs, mu)
                    torch._foreach_add_(bufs, grads)
                torch._foreach_add_(params, bufs, alpha=-lr)
            else:
                torch._foreach_add_(params, grads, alpha=-lr)
            return
        b1, b2 = group["betas"]
        t = group["step"]
        if mode == "adam" and wd != 0:
            grads = torch._foreach_add(grads, params, alpha=wd)
        elif mode == "adamw" and wd != 0:
            torch._foreach_mul_(params, 1.0 - lr * wd)
        m = [self.state[p]["exp_avg"] for p in params]
        v = [self.state[p]["exp_avg_sq"] for p in params]
        torch._foreach_lerp_(m, grads, 1.0 - b1)
        torch._foreach_mul_(v, b2)
        torch._foreach_addcmul_(v, grads, grads, value=1.0 - b2)
        bc1 = 1.0 - b1 ** t
        bc2_sqrt = math.sqrt(1.0 - b2 ** t)
        denom = torch._foreach_sqrt(v)
        torch._foreach_div_(denom, bc2_sqrt)
        torch._foreach_add_(denom, group["eps"])
        torch._foreach_addcdiv_(params, m, denom, value=-lr / bc1)

    # ------------------------------------------------------------------ GPU: one launch per group
    def _step_hip(self, gi: int, group, params: List[torch.Tensor]) -> None:
        from ..ops import hip

        plan = self._plans.get(gi)
        if plan is None or not plan.matches(params):
            states = (
                [[self.state[p]["momentum_buffer"]] for p in params]
                if group["mode"] == "sgd"
                else [[self.state[p]["exp_avg"], self.state[p]["exp_avg_sq"]] for p in params]
            )
            plan = hip.MultiTensorPlan(params, states)
            self._plans[gi] = plan
        hip.multi_tensor_step(plan, group, self._hyper_blocks.get(gi) if self.capturable else None)

    def zero_grad(self, set_to_none: bool = False) -> None:
        """Zero the gradients in place (the default keeps ``.grad`` allocated: kernels accumulate into it).

        On GPU the gradients of a group are re-homed, once, as views of one flat buffer, so clearing them is a
        single memset instead of one fill launch per tensor.  Parameters managed by a gradient reducer already
        live in its buckets and are left alone (use the reducer's ``zero_grad``)."""
        if set_to_none:
            self._flat_grads.clear()
            super().zero_grad(set_to_none=True)
            return
        for gi, group in enumerate(self.param_groups):
            params = [p for p in group["params"] if p.requires_grad]
            if not params:
                continue
            if not params[0].is_cuda or any(hasattr(p, "_pdt_bucket_view") for p in params):
                for p in params:
                    if p.grad is not None:
                        p.grad.zero_()
                continue
            flat = self._flat_grads.get(gi)
            if flat is None or any(p.grad is None or p.grad.data_ptr() != ptr for p, ptr in zip(params, flat[1])):
                flat = self._build_flat_grads(params)
                self._flat_grads[gi] = flat
            flat[0].zero_()

    @staticmethod
    def _build_flat_grads(params: List[torch.Tensor]):
        offs, total = [], 0
        for p in params:
            offs.append(total)
            total += (p.numel() + 63) // 64 * 64
        buf = torch.zeros(total, dtype=params[0].dtype, device=params[0].device)
        ptrs = []
        for p, off in zip(params, offs):
            view = torch.as_strided(buf, p.shape, p.stride(), off) if p.numel() else buf[off:off].view(p.shape)
            if p.grad is not None:
                view.copy_(p.grad)
            p.grad = view
            ptrs.append(view.data_ptr())
        return buf, ptrs


def build_optimizer(params, name: str, lr: float, weight_decay: float, momentum: float = 0.9,
                    betas=(0.9, 0.999), eps: float = 1e-8) -> FusedOptimizer:
    return FusedOptimizer(params, mode=name, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, momentum=momentum)
