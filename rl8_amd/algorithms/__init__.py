"""PPO algorithms (data collection and policy updates)."""

from ._feedforward import Algorithm, AlgorithmConfig
from ._recurrent import RecurrentAlgorithm, RecurrentAlgorithmConfig

__all__ = ["Algorithm", "AlgorithmConfig", "RecurrentAlgorithm", "RecurrentAlgorithmConfig"]
