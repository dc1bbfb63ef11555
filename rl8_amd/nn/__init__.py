"""Functional PPO ops backed by the gfx950 kernels."""

from .functional import generalized_advantage_estimate, ppo_losses

__all__ = ["generalized_advantage_estimate", "ppo_losses"]
