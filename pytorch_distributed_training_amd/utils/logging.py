"""Rank-aware logging.

The reference prints from every rank with bare ``print`` calls
(/root/reference/src/main.py:42,59,66,82,84).  Here the five contract lines are
still printed with ``print`` by the CLI, while everything else goes through a
logger that tags records with the rank and can be silenced on non-zero ranks.
"""

from __future__ import annotations

import logging
import os
import sys

_RANK = int(os.environ.get("RANK", "0"))


def set_rank(rank: int) -> None:
    global _RANK
    _RANK = int(rank)


def get_rank() -> int:
    return _RANK


class _RankFilter(logging.Filter):
    def filter(self, record: logging.LogRecord) -> bool:
        record.rank = _RANK
        return True


def get_logger(name: str = "pdt") -> logging.Logger:
    logger = logging.getLogger(name)
    if not logger.handlers:
        handler = logging.StreamHandler(sys.stderr)
        handler.setFormatter(
            logging.Formatter("[%(asctime)s r%(rank)s %(levelname).1s %(name)s] %(message)s", "%H:%M:%S")
        )
        handler.addFilter(_RankFilter())
        logger.addHandler(handler)
        logger.propagate = False
        logger.setLevel(os.environ.get("PDT_LOG_LEVEL", "INFO").upper())
    return logger


def rank_zero_print(*args, **kwargs) -> None:
    """``print`` on rank 0 only (extra, non-contract output)."""
    if _RANK == 0:
        print(*args, **kwargs)
        sys.stdout.flush()
