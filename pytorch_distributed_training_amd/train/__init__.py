from .checkpoint import load_checkpoint, load_model_state, portable_state_dict, save_checkpoint  # noqa: F401
from .trainer import TrainConfig, Trainer, TrainResult, run_training  # noqa: F401
