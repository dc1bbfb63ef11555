"""High-level training interfaces (``src/rl8/trainers/``)."""

from ._base import GenericTrainerBase
from ._trainers import RecurrentTrainer, Trainer
from .config import TrainConfig

__all__ = ["GenericTrainerBase", "RecurrentTrainer", "TrainConfig", "Trainer"]
