from .logging import get_logger, rank_zero_print, set_rank  # noqa: F401
from .meter import ThroughputMeter, AverageMeter  # noqa: F401
from .timers import StepTimer  # noqa: F401
