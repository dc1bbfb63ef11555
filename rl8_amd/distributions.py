"""Action distributions over model features.

API follows the reference's ``src/rl8/distributions.py``: ``Distribution`` ABC
:18-95 (``sample``, ``deterministic_sample``, ``logp``, ``entropy``,
``default_dist_cls``), ``Categorical`` :125-132, ``Normal`` :135-144,
``SquashedNormal`` :147-170.

Sampling runs in the HIP sampler kernels (``rl8_categorical_sample_logp_f32`` /
``rl8_normal_sample_logp_f32``), which return the action AND its log-probability
from one launch and draw noise from the build's Philox stream (or take injected
noise, which is how parity with the reference's recorded draws is tested).
``logp`` / ``entropy`` remain differentiable tensor expressions for callers that
compose their own losses; the training path does not use them --
:func:`rl8_amd.nn.functional.ppo_losses` recognises the three built-in
distributions and runs the fused forward+backward kernel instead.

"""

from __future__ import annotations

import math
from abc import ABC, abstractmethod
from typing import Any

import torch

from . import hip
from ._utils import assert_1d_spec
from .specs import Categorical as Discrete
from .specs import TensorSpec, Unbounded
from .tensordict import TensorDict


class NoiseStream:
    """Address of the next draw in the build's Philox stream: ``(seed, step)``
    plus the global row offset of the first sample (env sharding). ``step``
    advances by one per sampling call so that no two calls reuse noise."""

    def __init__(self, seed: None | int = None) -> None:
        self._seed = seed
        self.step = 0
        self.row_offset = 0

    @property
    def seed(self) -> int:
        if self._seed is None:
            self._seed = int(torch.initial_seed()) & 0xFFFFFFFFFFFFFFFF
        return self._seed

    def reseed(self, seed: int) -> None:
        self._seed = seed & 0xFFFFFFFFFFFFFFFF
        self.step = 0

    def next_step(self) -> int:
        s = self.step
        self.step += 1
        return s


#: Stream used by ``Distribution.sample()`` when the caller gives none.
default_noise = NoiseStream()


class Distribution(ABC):
    """Probability distribution over actions, parameterised by model features.

    Args:
        features: Output of the model's forward pass.
        model: The model (some distributions need its components).

    """

    features: TensorDict
    model: Any

    #: Injected noise for the next ``sample()`` (tests / parity runs).
    noise: None | torch.Tensor = None
    #: Philox address for the next ``sample()``; ``None`` -> ``default_noise``.
    noise_stream: None | NoiseStream = None

    def __init__(self, features: TensorDict, model: Any, /) -> None:
        super().__init__()
        self.features = features
        self.model = model

    @staticmethod
    def default_dist_cls(action_spec: TensorSpec, /) -> type["Distribution"]:
        """``Categorical`` for discrete specs, ``Normal`` for unbounded ones."""
        assert_1d_spec(action_spec)
        if isinstance(action_spec, Discrete):
            return Categorical
        if isinstance(action_spec, Unbounded):
            return Normal
        raise TypeError(f"Action spec {action_spec} has no default distribution support.")

    @abstractmethod
    def deterministic_sample(self) -> torch.Tensor | TensorDict:
        """The distribution's mode."""

    @abstractmethod
    def entropy(self) -> torch.Tensor:
        """Entropy, summed over action dims, shape ``[B, 1]``."""

    @abstractmethod
    def logp(self, samples: torch.Tensor | TensorDict) -> torch.Tensor:
        """Log-probability of ``samples``, summed over action dims, ``[B, 1]``."""

    @abstractmethod
    def sample(self) -> torch.Tensor | TensorDict:
        """A stochastic sample."""

    def sample_with_logp(self, *, deterministic: bool = False) -> tuple[Any, torch.Tensor]:
        """Sample and its log-probability. Built-in distributions get both from
        one kernel launch; custom ones fall back to ``sample`` then ``logp``."""
        actions = self.deterministic_sample() if deterministic else self.sample()
        return actions, self.logp(actions)

    def _address(self) -> tuple[int, int, int]:
        stream = self.noise_stream or default_noise
        return stream.seed, stream.next_step(), stream.row_offset


class Categorical(Distribution):
    """Categorical over ``features["logits"]`` of shape ``[B, A, K]``."""

    def __init__(self, features: TensorDict, model: Any, /) -> None:
        super().__init__(features, model)
        self.logits = features["logits"]

    def _normalised(self) -> torch.Tensor:
        return self.logits - self.logits.logsumexp(dim=-1, keepdim=True)

    def sample_with_logp(self, *, deterministic: bool = False) -> tuple[torch.Tensor, torch.Tensor]:
        seed, step, row_offset = self._address()
        return hip.categorical_sample_logp(
            self.logits.contiguous(),
            None if deterministic else self.noise,
            seed=seed,
            step=step,
            row_offset=row_offset,
            deterministic=deterministic,
        )

    def sample(self) -> torch.Tensor:
        return self.sample_with_logp()[0]

    def deterministic_sample(self) -> torch.Tensor:
        return self.sample_with_logp(deterministic=True)[0]

    def logp(self, samples: torch.Tensor) -> torch.Tensor:
        nl = self._normalised()
        return nl.gather(-1, samples.long().unsqueeze(-1)).squeeze(-1).sum(-1, keepdim=True)

    def entropy(self) -> torch.Tensor:
        nl = self._normalised()
        return -(nl * nl.exp()).sum(-1).sum(-1, keepdim=True)


class Normal(Distribution):
    """Diagonal normal over ``features["mean"]`` / ``features["log_std"]``,
    each ``[B, A]``."""

    squashed = False

    def __init__(self, features: TensorDict, model: Any) -> None:
        super().__init__(features, model)
        self.mean = features["mean"]
        self.log_std = features["log_std"]

    def sample_with_logp(self, *, deterministic: bool = False) -> tuple[torch.Tensor, torch.Tensor]:
        seed, step, row_offset = self._address()
        return hip.normal_sample_logp(
            self.mean.contiguous(),
            self.log_std.contiguous(),
            None if deterministic else self.noise,
            squashed=self.squashed,
            seed=seed,
            step=step,
            row_offset=row_offset,
            deterministic=deterministic,
        )

    def sample(self) -> torch.Tensor:
        return self.sample_with_logp()[0]

    def deterministic_sample(self) -> torch.Tensor:
        return self.sample_with_logp(deterministic=True)[0]

    def _log_prob(self, value: torch.Tensor) -> torch.Tensor:
        scale = torch.exp(self.log_std)
        var = scale**2
        return (
            -((value - self.mean) ** 2) / (2 * var)
            - scale.log()
            - math.log(math.sqrt(2 * math.pi))
        )

    def logp(self, samples: torch.Tensor) -> torch.Tensor:
        return self._log_prob(samples).sum(-1, keepdim=True)

    def entropy(self) -> torch.Tensor:
        return (0.5 + 0.5 * math.log(2 * math.pi) + self.log_std).sum(-1, keepdim=True)


class SquashedNormal(Normal):
    """Normal squashed through ``tanh`` so samples lie in ``[-1, 1]``."""

    squashed = True

    def entropy(self) -> torch.Tensor:
        raise NotImplementedError(
            f"Entropy isn't defined for {self.__class__.__name__}. Set the"
            " entropy coefficient to `0` to avoid this error during training."
        )

    def logp(self, samples: torch.Tensor) -> torch.Tensor:
        eps = torch.finfo(samples.dtype).eps
        clipped = samples.clamp(min=-1 + eps, max=1 - eps)
        inverted = 0.5 * (clipped.log1p() - (-clipped).log1p())
        logp = torch.clamp(self._log_prob(inverted), min=-100, max=100).sum(-1, keepdim=True)
        return logp - torch.sum(torch.log(1 - samples**2 + eps), dim=-1, keepdim=True)
