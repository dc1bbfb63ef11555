"""The two concrete trainers (``src/rl8/trainers/_feedforward.py:7-14``,
``src/rl8/trainers/_recurrent.py:7-14``)."""

from ..algorithms import Algorithm, RecurrentAlgorithm
from ._base import GenericTrainerBase


class Trainer(GenericTrainerBase[Algorithm]):
    """Training loop for feed-forward policies."""


class RecurrentTrainer(GenericTrainerBase[RecurrentAlgorithm]):
    """Training loop for recurrent policies."""
