"""CartPole as a batched HIP kernel.

Same task as the reference's ``examples/cartpole/env.py`` (``step`` :12-64,
``CartPoleConfig`` :67-98, ``CartPole`` :101-150): three discrete actions (push
left / none / right), observation ``(x, x_dot, cos theta, sin theta,
theta_dot)``, reward = minus the L1 distance from upright-and-still,
``max_horizon = 128``. State is struct-of-arrays ``[4, num_envs]``.

The reference fuses its ~40 tensor ops with ``@torch.compile``; here the physics
is ``rl8_cartpole_step_f32`` (one launch), and inside ``Algorithm.collect()`` the
sampler, the physics and the buffer bookkeeping are one launch per timestep
(``rl8_rollout_step_cartpole_f32``).

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
class CartPoleConfig:
    #: Cart mass.
    cart_mass: float = 1.0
    #: Force magnitude applied to the cart.
    force_mag: float = 5.0
    #: Gravity.
    gravity: float = 9.8
    #: ``"euler"`` or anything else for semi-implicit Euler.
    kinematics_integrator: str = "euler"
    #: Pole (half-)length.
    length: float = 0.5
    #: Pole mass.
    pole_mass: float = 0.1
    #: Pole mass * pole length (derived).
    pole_mass_length: float = 0.05
    #: Pole mass + cart mass (derived).
    total_mass: float = 1.1
    #: Timestep.
    tau: float = 0.02

    def __post_init__(self) -> None:
        self.pole_mass_length = self.pole_mass * self.length
        self.total_mass = self.cart_mass + self.pole_mass

    def to_abi(self) -> hip.CartPoleCfg:
        return hip.CartPoleCfg(
            self.force_mag, self.gravity, self.length, self.pole_mass, self.pole_mass_length,
            self.total_mass, self.tau, 0 if self.kinematics_integrator == "euler" else 1,
        )


class CartPole(Env):
    """Batched CartPole with a continuous, shaped reward."""

    max_horizon = 128

    #: ``[4, num_envs]`` rows ``x, x_dot, theta, theta_dot``.
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
        self.observation_spec = Unbounded(5, device=device, dtype=torch.float32)
        self.action_spec = Categorical(3, shape=torch.Size([1]), device=device)
        self.seed = default_seed()
        self.reset_count = 0
        self._config = CartPoleConfig()
        self._abi_config = self._config.to_abi()

    @property
    def config(self) -> dict[str, Any]:
        return asdict(self._config)

    def reset(self, *, config: dict[str, Any] | None = None) -> torch.Tensor:
        self._config = CartPoleConfig(**(config or {}))
        self._abi_config = self._config.to_abi()
        self.state = torch.empty(4, self.num_envs, dtype=torch.float32, device=self.device)
        obs = torch.empty(self.num_envs, 5, dtype=torch.float32, device=self.device)
        hip.cartpole_reset(self.state, 0.01, self.seed, self.reset_count, self.env_offset, obs)
        self.reset_count += 1
        return obs

    def step(self, action: torch.Tensor) -> TensorDict:
        if action.dtype != torch.int64:
            action = action.to(torch.int64)
        obs = torch.empty(self.num_envs, 5, dtype=torch.float32, device=self.device)
        reward = torch.empty(self.num_envs, 1, dtype=torch.float32, device=self.device)
        hip.cartpole_step(self.state, action.contiguous(), self._abi_config, obs, reward)
        return TensorDict(
            {DataKeys.OBS: obs, DataKeys.REWARDS: reward},
            batch_size=self.num_envs,
            device=self.device,
        )

    def fused_rollout_step(self, *, squashed: bool, features: torch.Tensor, features2: Any, **kw: Any) -> None:
        del squashed, features2
        hip.rollout_step_cartpole(logits=features, state=self.state, cfg=self._abi_config, **kw)
