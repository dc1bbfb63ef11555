"""PPO algorithms (data collection and policy updates)."""

from ._feedforward import Algorithm, AlgorithmConfig

__all__ = ["Algorithm", "AlgorithmConfig"]
