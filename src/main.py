"""Entry point at the same path and with the same flags as the reference script
(/root/reference/src/main.py:18-26,87-88):

    python src/main.py [--use-cpu] [--batch-size N] ...
    torchrun --nproc-per-node=N src/main.py --distributed ...
"""

import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from pytorch_distributed_training_amd.cli import main  # noqa: E402

if __name__ == "__main__":
    main()
