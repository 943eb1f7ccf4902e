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

import os as _os

# Sharing device memory between the processes of a node (RCCL's peer-to-peer buffers over xGMI) goes through dmabuf
# handles; hosts whose driver offers only that need the legacy IPC mode off *before* the HIP runtime starts, otherwise
# communicator set-up with more than one rank fails in hipIpcGetMemHandle.  A value given by the user is kept.
_os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

from . import utils  # noqa: F401
