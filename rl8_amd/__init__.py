"""rl8_amd: the rl8 PPO hot path on AMD MI355X (gfx950).

A from-scratch drop-in for the ``Env`` / ``Model`` / ``Distribution`` /
``Algorithm.collect()`` / ``Algorithm.step()`` surface of theOGognf/rl8, with
the per-timestep env step + sampling + bookkeeping, the GAE reverse scan, the
PPO loss (forward and backward), the rollout statistics and the minibatch gather
written as HIP kernels behind a C ABI (``include/rl8_amd.h``). The policy / value
networks run on PyTorch-ROCm.

Importing the package never needs a GPU; running anything does.
"""

from .algorithms import (
    Algorithm,
    AlgorithmConfig,
    RecurrentAlgorithm,
    RecurrentAlgorithmConfig,
)
from .env import Env
from .trainers import RecurrentTrainer, TrainConfig, Trainer

__all__ = [
    "Algorithm",
    "AlgorithmConfig",
    "Env",
    "RecurrentAlgorithm",
    "RecurrentAlgorithmConfig",
    "RecurrentTrainer",
    "TrainConfig",
    "Trainer",
]
__version__ = "0.1.0"
