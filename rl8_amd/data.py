"""Names and records shared across the PPO path: buffer keys, hyperparameters,
algorithm state and the stat dictionaries ``collect()`` / ``step()`` return.

These are part of the drop-in contract: key strings, field names, defaults,
validation rules and error types follow the reference's ``src/rl8/data.py``
(``DataKeys`` :12-76, ``AlgorithmHparams`` :79-270, ``AlgorithmState`` :329-353,
``CollectStats`` / ``StepStats`` :367-430).

"""

from __future__ import annotations

from dataclasses import dataclass
from typing import Literal, TypedDict, Union

import torch

Device = Union[str, torch.device]


class DataKeys:
    """String identifiers of batch elements."""

    OBS = "obs"
    REWARDS = "rewards"
    RETURNS = "returns"
    FEATURES = "features"
    ACTIONS = "actions"
    LOGP = "logp"
    VALUES = "values"
    INPUTS = "inputs"
    PADDING_MASK = "padding_mask"
    VIEWS = "views"
    ADVANTAGES = "advantages"
    STATES = "states"
    HIDDEN_STATES = "hidden_states"
    CELL_STATES = "cell_states"
    REVERSED_DISCOUNTED_RETURNS = "reversed_discounted_returns"


def _require(condition: bool, message: str) -> None:
    if not condition:
        raise ValueError(message)


@dataclass(frozen=True, kw_only=True)
class AlgorithmHparams:
    """Feed-forward PPO hyperparameters, fixed for the lifetime of an
    algorithm and validated on construction."""

    accumulate_grads: bool
    clip_param: float
    device: Device
    dual_clip_param: None | float
    enable_amp: bool
    gae_lambda: float
    gamma: float
    horizon: int
    horizons_per_env_reset: int
    max_grad_norm: float
    normalize_advantages: bool
    normalize_rewards: bool
    num_envs: int
    num_sgd_iters: int
    sgd_minibatch_size: int
    shuffle_minibatches: bool
    target_kl_div: None | float
    vf_clip_param: float
    vf_coeff: float

    def __post_init__(self) -> None:
        _require(0 < self.clip_param < 1, "`clip_param` must be in (0, 1).")
        _require(
            self.dual_clip_param is None or self.dual_clip_param > 1,
            "`dual_clip_param` must be `None` or > 1.",
        )
        _require(
            not (str(self.device) == "cpu" and self.enable_amp),
            "`enable_amp` may only be used with CUDA devices.",
        )
        _require(0 < self.gae_lambda <= 1, "`gae_lambda` must be in (0, 1].")
        _require(0 < self.gamma <= 1, "`gamma` must be in (0, 1].")
        _require(self.horizon > 0, "`horizon` must be > 0.")
        _require(self.horizons_per_env_reset != 0, "`horizons_per_env_reset` must be nonzero.")
        _require(self.max_grad_norm > 0, "`max_grad_norm` must be > 0.")
        _require(self.num_sgd_iters > 0, "`num_sgd_iters` must be > 0.")
        _require(self.sgd_minibatch_size > 0, "`sgd_minibatch_size` must be > 0.")
        _require(
            not (self.target_kl_div is not None and self.accumulate_grads),
            "Early-stopping using `target_kl_div` is not compatible with gradient"
            " accumulation.",
        )
        _require(
            not (self.target_kl_div is not None and self.enable_amp),
            "Early-stopping using `target_kl_div` is not compatible with AMP.",
        )
        _require(
            self.target_kl_div is None or self.target_kl_div > 0,
            "`target_kl_div` must be > 0.",
        )
        _require(self.vf_clip_param > 0, "`vf_clip_param` must be > 0.")
        _require(self.vf_coeff > 0, "`vf_coeff` must be > 0.")
        _require(
            not (self.accumulate_grads and self.num_minibatches == 1),
            "`accumulate_grads` is `True` but there's only one minibatch during"
            " training, making gradient accumulation irrelevant. Update"
            " `sgd_minibatch_size` or disable `accumulate_grads`.",
        )

    @property
    def device_type(self) -> Literal["cpu", "cuda"]:
        return "cuda" if str(self.device) != "cpu" else "cpu"

    @property
    def num_minibatches(self) -> int:
        return (self.num_envs * self.horizon) // self.sgd_minibatch_size

    def validate(self) -> "AlgorithmHparams":
        """Checks that need the final sizes."""
        _require(
            (self.num_envs * self.horizon) % self.sgd_minibatch_size == 0,
            "`sgd_minibatch_size` must be a factor of `num_envs * horizon`.",
        )
        return self


@dataclass(frozen=True, kw_only=True)
class RecurrentAlgorithmHparams(AlgorithmHparams):
    """Recurrent PPO hyperparameters (truncated BPTT over ``seq_len``)."""

    seq_len: int
    seqs_per_state_reset: int

    def __post_init__(self) -> None:
        super().__post_init__()
        _require(self.seq_len > 0, "`seq_len` must be > 0.")
        _require(self.horizon % self.seq_len == 0, "`seq_len` must be a factor of `horizon`.")
        _require(self.seqs_per_state_reset != 0, "`seqs_per_state_reset` must be nonzero.")
        _require(
            (self.horizon * self.horizons_per_env_reset)
            % (self.seq_len * self.seqs_per_state_reset)
            == 0,
            "`seq_len * seqs_per_state_reset` must be a factor of `horizon *"
            " horizons_per_env_reset`. As an example, if `horizon=8`,"
            " `horizons_per_env_reset=1`, and `seq_len=2`, then"
            " `seqs_per_state_reset` can be 1, 2, or 4.",
        )

    @property
    def num_minibatches(self) -> int:
        return (self.num_envs * (self.horizon // self.seq_len)) // self.sgd_minibatch_size

    def validate(self) -> "RecurrentAlgorithmHparams":
        _require(
            (self.num_envs * (self.horizon // self.seq_len)) % self.sgd_minibatch_size == 0,
            "`sgd_minibatch_size` must be a factor of `num_envs * (horizon //"
            " seq_len)`.",
        )
        return self


@dataclass(kw_only=True)
class AlgorithmState:
    """Mutable feed-forward PPO state."""

    #: ``collect()`` has run since the last ``step()``.
    buffered: bool = False
    #: Number of horizons collected so far (drives the env-reset cadence).
    horizons: int = 0
    #: std of the reversed discounted returns of the last ``collect()``.
    reward_scale: float = 1.0


@dataclass(kw_only=True)
class RecurrentAlgorithmState(AlgorithmState):
    #: Number of recurrent sequences collected so far.
    seqs: int = 0


TrainerState = TypedDict(
    "TrainerState",
    {"algorithm/collects": int, "algorithm/steps": int, "env/steps": int},
)

CollectStats = TypedDict(
    "CollectStats",
    {
        "env/resets": int,
        "env/steps": int,
        "profiling/collect_ms": float,
        "returns/min": float,
        "returns/max": float,
        "returns/mean": float,
        "returns/std": float,
        "rewards/min": float,
        "rewards/max": float,
        "rewards/mean": float,
        "rewards/std": float,
    },
    total=False,
)

MemoryStats = TypedDict(
    "MemoryStats",
    {"memory/free": int, "memory/total": int, "memory/percent": float},
    total=False,
)

StepStats = TypedDict(
    "StepStats",
    {
        "coefficients/entropy": float,
        "coefficients/vf": float,
        "losses/entropy": float,
        "losses/policy": float,
        "losses/vf": float,
        "losses/total": float,
        "monitors/kl_div": float,
        "profiling/step_ms": float,
    },
    total=False,
)


class TrainStats(CollectStats, MemoryStats, StepStats, TrainerState):
    ...


#: ``eval/``-prefixed collect statistics returned by ``Trainer.eval``.
EvalCollectStats = dict[str, float]

#: Any key of :class:`TrainStats` (what a stop condition monitors).
TrainStatKey = str


__all__ = [
    "AlgorithmHparams",
    "AlgorithmState",
    "CollectStats",
    "DataKeys",
    "Device",
    "EvalCollectStats",
    "MemoryStats",
    "RecurrentAlgorithmHparams",
    "RecurrentAlgorithmState",
    "StepStats",
    "TrainStatKey",
    "TrainStats",
    "TrainerState",
]
