"""Pendulum as a batched HIP kernel.

Same task as the reference's ``examples/pendulum/env.py`` (``step`` :12-39,
``PendulumConfig`` :42-60, ``Pendulum`` :63-127): one continuous action (torque,
clipped to ``+-max_torque``), observation ``(cos th, sin th, thdot)``, reward =
minus the quadratic cost of angle, speed and torque, ``max_horizon = 512``. State
is struct-of-arrays ``[2, num_envs]`` rows ``th, thdot``.

The physics is ``rl8_pendulum_step_f32`` (one launch); inside
``Algorithm.collect()`` the ``Normal`` / ``SquashedNormal`` sampler, the physics
and the buffer bookkeeping are one launch per timestep
(``rl8_rollout_step_pendulum_f32``).

"""

from __future__ import annotations

from dataclasses import asdict, dataclass
from typing import Any

import torch

from .. import hip
from ..data import DataKeys, Device
from ..distributions import Normal, SquashedNormal
from ..env import Env, default_seed
from ..specs import Unbounded
from ..tensordict import TensorDict


@dataclass
class PendulumConfig:
    #: Timestep between step calls.
    dt: float = 0.05
    #: Gravity.
    g: float = 10.0
    #: Pendulum length.
    l: float = 1.0  # noqa: E741
    #: System mass.
    m: float = 1.0
    #: Pendulum max angular speed.
    max_speed: float = 8.0
    #: Max torque that can be applied to the pendulum.
    max_torque: float = 2.0

    def to_abi(self) -> hip.PendulumCfg:
        # the two physics coefficients, in double, as the reference forms them
        # before they meet the fp32 tensors (examples/pendulum/env.py:32)
        return hip.PendulumCfg(
            self.dt, 3 * self.g / (2 * self.l), 3.0 / (self.m * self.l**2), self.max_speed, self.max_torque
        )


class Pendulum(Env):
    """Batched Pendulum swing-up."""

    max_horizon = 512

    #: ``[2, num_envs]`` rows ``th, thdot``.
    state: torch.Tensor

    #: Distributions the fused per-timestep kernel implements for this env.
    fused_distributions = (Normal, SquashedNormal)

    def __init__(
        self,
        num_envs: int,
        /,
        horizon: None | int = None,
        *,
        device: Device = "cpu",
    ) -> None:
        super().__init__(num_envs, horizon, device=device)
        self.action_spec = Unbounded(device=device, dtype=torch.float32, shape=torch.Size([1]))
        self.observation_spec = Unbounded(3, device=device, dtype=torch.float32)
        self.seed = default_seed()
        self.reset_count = 0
        self._config = PendulumConfig()
        self._abi_config = self._config.to_abi()

    @property
    def config(self) -> dict[str, Any]:
        return asdict(self._config)

    def reset(self, *, config: dict[str, Any] | None = None) -> torch.Tensor:
        self._config = PendulumConfig(**(config or {}))
        self._abi_config = self._config.to_abi()
        self.state = torch.empty(2, self.num_envs, dtype=torch.float32, device=self.device)
        obs = torch.empty(self.num_envs, 3, dtype=torch.float32, device=self.device)
        hip.pendulum_reset(self.state, self.seed, self.reset_count, self.env_offset, obs)
        self.reset_count += 1
        return obs

    def step(self, action: torch.Tensor) -> TensorDict:
        if action.dtype != torch.float32:
            action = action.to(torch.float32)
        obs = torch.empty(self.num_envs, 3, dtype=torch.float32, device=self.device)
        reward = torch.empty(self.num_envs, 1, dtype=torch.float32, device=self.device)
        hip.pendulum_step(self.state, action.contiguous(), self._abi_config, obs, reward)
        return TensorDict(
            {DataKeys.OBS: obs, DataKeys.REWARDS: reward},
            batch_size=self.num_envs,
            device=self.device,
        )

    def fused_rollout_step(self, *, squashed: bool, features: torch.Tensor, features2: Any, **kw: Any) -> None:
        hip.rollout_step_pendulum(
            squashed=squashed, mean=features, log_std=features2, state=self.state, cfg=self._abi_config, **kw
        )
