"""Environment protocol and the built-in dummy environments, stepped by HIP
kernels.

API and semantics follow the reference's ``src/rl8/env.py``: ``Env`` :16-128,
``EnvFactory`` :131-151, ``DummyEnv.reset`` :197-203, ``ContinuousDummyEnv.step``
:224-230, ``DiscreteDummyEnv.step`` :253-259. An ``Env`` holds ``num_envs``
independent simulations as ``[num_envs, ...]`` device tensors; ``step`` takes an
action tensor and returns a tensordict with ``"obs"`` and ``"rewards"``.

What differs is the machinery: ``step`` is one launch of
``rl8_dummy_env_step_f32`` (the reference issues ~5 eager ops), ``reset`` draws
from the build's counter-based Philox stream (``include/rl8_philox.h``) instead
of torch's generator, and environments that implement the optional
``fused_rollout_step`` hook let ``Algorithm.collect()`` do sampling + step +
buffer bookkeeping in a single launch per timestep.

"""

from __future__ import annotations

from abc import ABC, abstractmethod
from typing import Any, ClassVar, Generic, Protocol, TypeVar

import torch

from . import hip
from .data import DataKeys, Device
from .specs import Categorical, TensorSpec, Unbounded
from .tensordict import TensorDict

_ObservationSpec = TypeVar("_ObservationSpec", bound=TensorSpec)
_ActionSpec = TypeVar("_ActionSpec", bound=TensorSpec)


def default_seed() -> int:
    """Seed of the build's noise stream: follows ``torch.manual_seed`` so that
    seeding a run the reference's way seeds this one too."""
    return int(torch.initial_seed()) & 0xFFFFFFFFFFFFFFFF


class Env(ABC):
    """Protocol of IsaacGym-like, massively parallel environments.

    Subclasses define ``observation_spec`` / ``action_spec`` and implement
    :meth:`reset` and :meth:`step`.

    Args:
        num_envs: Number of parallel, independent environments simulated by
            one instance.
        horizon: Number of steps expected before a reset (``None``: may never
            reset).
        device: Device all environment data lives on.

    """

    action_spec: TensorSpec
    device: Device
    horizon: None | int
    #: Optional cap on ``horizon``; validated at construction.
    max_horizon: ClassVar[int]
    #: Optional cap on ``num_envs``; validated at construction.
    max_num_envs: ClassVar[int]
    num_envs: int
    observation_spec: TensorSpec

    #: Global index of this instance's first environment when environments are
    #: sharded across GPUs (keys the noise stream so that shards draw what the
    #: unsharded run would).
    env_offset: int = 0

    def __init__(
        self,
        num_envs: int,
        /,
        horizon: None | int = None,
        *,
        device: Device = "cpu",
    ) -> None:
        if hasattr(self, "max_horizon") and horizon is not None:
            if not (horizon <= self.max_horizon):
                raise ValueError(
                    f"{self.__class__.__name__} `horizon` must be <= {self.max_horizon}."
                )
        if hasattr(self, "max_num_envs"):
            if not (num_envs <= self.max_num_envs):
                raise ValueError(
                    f"{self.__class__.__name__} `num_envs` must be <= {self.max_num_envs}."
                )
        self.num_envs = num_envs
        self.horizon = horizon
        self.device = device

    @abstractmethod
    def reset(self, *, config: None | dict[str, Any] = None) -> torch.Tensor | TensorDict:
        """Reset every environment and return the initial observation."""

    @abstractmethod
    def step(self, action: torch.Tensor | TensorDict) -> TensorDict:
        """Apply ``action``; return a tensordict with ``"obs"`` and ``"rewards"``."""


class EnvFactory(Protocol):
    """Anything that builds an :class:`Env` from ``(num_envs, horizon, device=)``."""

    max_horizon: ClassVar[int]
    max_num_envs: ClassVar[int]

    def __call__(
        self,
        num_envs: int,
        /,
        horizon: None | int = None,
        *,
        device: Device = "cpu",
    ) -> Env:
        ...


class GenericEnv(Env, Generic[_ObservationSpec, _ActionSpec]):
    """``Env`` with statically typed specs."""

    observation_spec: _ObservationSpec
    action_spec: _ActionSpec


class DummyEnv(GenericEnv[Unbounded, _ActionSpec]):
    """A point on a line; the action nudges it; the reward is minus its distance
    from the origin. Action space and step are defined by subclasses."""

    #: Initial states are uniform in ``[-bounds, bounds]``.
    bounds: float
    #: Current positions, ``[num_envs, 1]`` float32. The observation IS the state
    #: (``step`` returns this tensor, as the reference does).
    state: torch.Tensor

    def __init__(
        self,
        num_envs: int,
        /,
        horizon: None | int = None,
        *,
        device: Device = "cpu",
    ) -> None:
        super().__init__(num_envs, horizon, device=device)
        self.observation_spec = Unbounded(1, device=self.device)
        self.bounds = 100.0
        self.seed = default_seed()
        self.reset_count = 0

    def reset(self, *, config: None | dict[str, Any] = None) -> torch.Tensor:
        config = config or {}
        self.bounds = config.get("bounds", self.bounds)
        self.state = torch.empty(self.num_envs, 1, dtype=torch.float32, device=self.device)
        hip.dummy_env_reset(
            self.state, float(self.bounds), self.seed, self.reset_count, self.env_offset
        )
        self.reset_count += 1
        return self.state

    def _step(self, action: torch.Tensor) -> TensorDict:
        rewards = torch.empty_like(self.state)
        hip.dummy_env_step(self.state, action.contiguous(), rewards)
        return TensorDict(
            {DataKeys.OBS: self.state, DataKeys.REWARDS: rewards},
            batch_size=self.num_envs,
            device=self.device,
        )

    # -- optional hook used by Algorithm.collect() -------------------------
    def fused_rollout_step(self, **kw: Any) -> None:
        """Sampler + step + buffer bookkeeping in one launch. ``kw`` are the
        arguments of :func:`rl8_amd.hip.rollout_step_dummy` minus ``state`` and
        ``discrete``."""
        hip.rollout_step_dummy(
            discrete=isinstance(self.action_spec, Categorical), state=self.state, **kw
        )


class ContinuousDummyEnv(DummyEnv[Unbounded]):
    """Actions move the state by any amount."""

    def __init__(
        self,
        num_envs: int,
        /,
        horizon: None | int = None,
        *,
        device: Device = "cpu",
    ) -> None:
        super().__init__(num_envs, horizon, device=device)
        self.action_spec = Unbounded(shape=torch.Size([1]), device=device)

    def step(self, action: torch.Tensor) -> TensorDict:
        if action.dtype != torch.float32:
            action = action.to(torch.float32)
        return self._step(action)


class DiscreteDummyEnv(DummyEnv[Categorical]):
    """Actions move the state one unit left (0) or right (1)."""

    def __init__(
        self,
        num_envs: int,
        /,
        horizon: None | int = None,
        *,
        device: Device = "cpu",
    ) -> None:
        super().__init__(num_envs, horizon, device=device)
        self.action_spec = Categorical(2, shape=torch.Size([1]), device=device)

    def step(self, action: torch.Tensor) -> TensorDict:
        if action.dtype != torch.int64:
            action = action.to(torch.int64)
        return self._step(action)
