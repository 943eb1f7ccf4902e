"""MI355X-native data-parallel training framework.

One process per GPU, ``torch.distributed`` (RCCL over xGMI) for rendezvous, an
in-repo gradient-bucket reducer instead of stock DDP, and hand-written gfx950 HIP
kernels (MFMA implicit-GEMM convolution/linear, fused BatchNorm+ReLU, fused
softmax-cross-entropy, multi-tensor optimizers) for the hot path.

Capability parity target: the single-script trainer at
``/root/reference/src/main.py:18-84`` (click CLI, process-group init, DDP wrap,
train loop), extended with sharded sampling, evaluation, checkpointing and
throughput metering.
"""

__version__ = "0.1.0"

from . import utils  # noqa: F401
