"""The reference's example environments (``examples/*/env.py``) as batched HIP kernels."""

from .cartpole import CartPole, CartPoleConfig
from .mountain_car import MountainCar, MountainCarConfig
from .pendulum import Pendulum, PendulumConfig

__all__ = ["CartPole", "CartPoleConfig", "MountainCar", "MountainCarConfig", "Pendulum", "PendulumConfig"]
