"""MountainCar as a batched HIP kernel.

Same task as the reference's ``examples/mountain_car/env.py`` (``step`` :12-38,
``MountainCarConfig`` :41-62, ``MountainCar`` :65-121): three discrete actions
(push left / none / right), observation ``(position, velocity)``, reward = minus
the distance from the goal, or +1 at the goal moving forward, ``max_horizon = 512``.
State is struct-of-arrays ``[2, num_envs]``.

The reference fuses its tensor ops with ``@torch.compile``; here the physics is
``rl8_mountain_car_step_f32`` (one launch), and inside ``Algorithm.collect()`` the
sampler, the physics and the buffer bookkeeping are one launch per timestep
(``rl8_rollout_step_mountain_car_f32``).

"""

from __future__ import annotations

from dataclasses import asdict, dataclass
from typing import Any

import torch

from .. import hip
from ..data import DataKeys, Device
from ..distributions import Categorical as CategoricalDistribution
from ..env import Env, default_seed
from ..specs import Categorical, Unbounded
from ..tensordict import TensorDict


@dataclass
class MountainCarConfig:
    #: Force applied to the car.
    force_mag: float = 0.001
    #: Car must be at or past this position for max reward.
    goal_position: float = 0.5
    #: Car must be moving at least this fast past the position for max reward.
    goal_velocity: float = 0.0
    #: Gravity pulling the car down the hill.
    gravity: float = 0.0025
    #: Car max position.
    max_position: float = 0.6
    #: Car max speed.
    max_speed: float = 0.07
    #: Car min position.
    min_position: float = -1.2

    def to_abi(self) -> hip.MountainCarCfg:
        return hip.MountainCarCfg(
            self.force_mag, self.goal_position, self.goal_velocity, self.gravity, self.max_position,
            self.max_speed, self.min_position,
        )


class MountainCar(Env):
    """Batched MountainCar with a continuous, shaped reward."""

    max_horizon = 512

    #: ``[2, num_envs]`` rows ``position, velocity``.
    state: torch.Tensor

    #: Distributions the fused per-timestep kernel implements for this env.
    fused_distributions = (CategoricalDistribution,)

    def __init__(
        self,
        num_envs: int,
        /,
        horizon: None | int = None,
        *,
        device: Device = "cpu",
    ) -> None:
        super().__init__(num_envs, horizon, device=device)
        self.observation_spec = Unbounded(2, device=device, dtype=torch.float32)
        self.action_spec = Categorical(3, shape=torch.Size([1]), device=device)
        self.seed = default_seed()
        self.reset_count = 0
        self._config = MountainCarConfig()
        self._abi_config = self._config.to_abi()

    @property
    def config(self) -> dict[str, Any]:
        return asdict(self._config)

    def reset(self, *, config: dict[str, Any] | None = None) -> torch.Tensor:
        self._config = MountainCarConfig(**(config or {}))
        self._abi_config = self._config.to_abi()
        self.state = torch.empty(2, self.num_envs, dtype=torch.float32, device=self.device)
        obs = torch.empty(self.num_envs, 2, dtype=torch.float32, device=self.device)
        hip.mountain_car_reset(self.state, self.seed, self.reset_count, self.env_offset, obs)
        self.reset_count += 1
        return obs

    def step(self, action: torch.Tensor) -> TensorDict:
        if action.dtype != torch.int64:
            action = action.to(torch.int64)
        obs = torch.empty(self.num_envs, 2, dtype=torch.float32, device=self.device)
        reward = torch.empty(self.num_envs, 1, dtype=torch.float32, device=self.device)
        hip.mountain_car_step(self.state, action.contiguous(), self._abi_config, obs, reward)
        return TensorDict(
            {DataKeys.OBS: obs, DataKeys.REWARDS: reward},
            batch_size=self.num_envs,
            device=self.device,
        )

    def fused_rollout_step(self, *, squashed: bool, features: torch.Tensor, features2: Any, **kw: Any) -> None:
        del squashed, features2
        hip.rollout_step_mountain_car(logits=features, state=self.state, cfg=self._abi_config, **kw)
