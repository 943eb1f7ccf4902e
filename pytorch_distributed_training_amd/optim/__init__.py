from .fused import FusedOptimizer, build_optimizer  # noqa: F401
