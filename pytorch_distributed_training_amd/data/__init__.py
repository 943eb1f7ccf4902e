from .datasets import CIFAR10, SyntheticImageDataset, SyntheticTokenDataset, build_dataset  # noqa: F401
from .loader import (  # noqa: F401
    BatchLoader,
    DevicePrefetcher,
    DeviceSyntheticLoader,
    build_loader,
    synthetic_image_batches,
    synthetic_token_batches,
)
from .sampler import ShardedSampler  # noqa: F401
