"""The trainer loop shared by the feed-forward and the recurrent trainers.

Mirrors ``GenericTrainerBase`` of the reference (``src/rl8/trainers/_base.py``:
``__init__`` :35-42, ``eval`` :44-101, ``run`` :103-177, ``step`` :179-201): a
trainer owns an algorithm and three running totals, ``step()`` is one
``collect()`` + ``step()`` of the algorithm, ``eval()`` collects
``horizons_per_env_reset`` deterministic horizons, ``run()`` alternates them
until a stop condition fires. The checks (and when they raise) are the
reference's; metric logging goes through :mod:`rl8_amd.tracking`.

"""

from __future__ import annotations

from collections import defaultdict
from typing import Any, Generic, Sequence, TypeVar

from .._utils import reduce_stats
from ..conditions import Condition
from ..data import EvalCollectStats, TrainerState, TrainStats
from ..tracking import Tracker, default_tracker

_Algorithm = TypeVar("_Algorithm")

_NO_EVAL_CONFIG = (
    "An eval environment config was provided even though the environment is not expected to use"
    " the config because `horizons_per_env_reset` is < 0 (indicating the environment is reset"
    " just once at the beginning of training). Either 1) do not provide an eval environment"
    " config, or 2) set `horizons_per_env_reset` > 0."
)


class GenericTrainerBase(Generic[_Algorithm]):
    #: The PPO algorithm (environment, policy, buffer, hyperparameters).
    algorithm: _Algorithm

    #: ``algorithm/collects``, ``algorithm/steps``, ``env/steps`` so far.
    state: TrainerState

    #: Where params and metrics are logged.
    tracker: Tracker

    def __init__(self, algorithm: _Algorithm, /, *, tracker: None | Tracker = None) -> None:
        self.algorithm = algorithm
        self.state = {"algorithm/collects": 0, "algorithm/steps": 0, "env/steps": 0}
        self.tracker = tracker if tracker is not None else default_tracker()
        self.tracker.log_params(self.algorithm.params)  # type: ignore[attr-defined]

    # -- evaluation ------------------------------------------------------------
    def eval(self, *, env_config: None | dict[str, Any] = None, deterministic: bool = True) -> EvalCollectStats:
        """Collect ``horizons_per_env_reset`` horizons without learning and log
        their reduced statistics under ``eval/``.

        Raises:
            ValueError: an ``env_config`` was given but the environment is only
                reset once (``horizons_per_env_reset < 0``) and has been already.
            RuntimeError: called between two environment resets; training and
                evaluation share the rollout buffer.
        """
        algo: Any = self.algorithm
        per_reset = algo.horizons_per_env_reset
        collects = self.state["algorithm/collects"]
        if env_config and per_reset < 0 and collects:
            raise ValueError(_NO_EVAL_CONFIG)
        if per_reset > 0 and collects % per_reset:
            raise RuntimeError(
                f"{type(self).eval.__qualname__} can only be called every `horizons_per_env_reset`. This is"
                " necessary because algorithms share the same buffer when collecting experiences for"
                " training and for evaluation."
            )
        per_key: dict[str, list[float]] = defaultdict(list)
        for _ in range(max(1, per_reset)):
            for key, value in algo.collect(env_config=env_config, deterministic=deterministic).items():
                per_key[key].append(value)
            self.state["algorithm/collects"] += 1
        eval_stats = {f"eval/{key}": value for key, value in reduce_stats(per_key).items()}
        self.tracker.log_metrics(eval_stats, step=self.state["env/steps"])
        return eval_stats  # type: ignore[return-value]

    # -- training --------------------------------------------------------------
    def step(self, *, env_config: None | dict[str, Any] = None) -> TrainStats:
        """One ``collect()`` + one ``step()`` of the algorithm; returns (and
        logs) memory, collect, step and trainer-state statistics together."""
        algo: Any = self.algorithm
        train_stats: dict[str, Any] = dict(algo.memory_stats())
        collect_stats = algo.collect(env_config=env_config)
        train_stats.update(collect_stats)
        train_stats.update(algo.step())
        self.state["algorithm/collects"] += 1
        self.state["algorithm/steps"] += 1
        self.state["env/steps"] += collect_stats["env/steps"]
        train_stats.update(self.state)
        self.tracker.log_metrics(train_stats, step=self.state["env/steps"])
        return train_stats  # type: ignore[return-value]

    def run(
        self,
        *,
        env_config: None | dict[str, Any] = None,
        eval_env_config: None | dict[str, Any] = None,
        steps_per_eval: None | int = None,
        stop_conditions: None | Sequence[Condition] = None,
    ) -> TrainStats:
        """Train until one of ``stop_conditions`` is true (forever without any),
        evaluating every ``steps_per_eval`` trainer steps.

        Raises:
            ValueError: ``steps_per_eval`` is not a multiple of the algorithm's
                ``horizons_per_env_reset``, or an eval config was given for an
                environment that is reset only once.
        """
        per_reset = self.algorithm.horizons_per_env_reset  # type: ignore[attr-defined]
        if steps_per_eval and per_reset < 0 and eval_env_config:
            raise ValueError(_NO_EVAL_CONFIG)
        if steps_per_eval and per_reset > 0 and steps_per_eval % per_reset:
            raise ValueError(
                f"{type(self).eval.__qualname__} can only be called every `horizons_per_env_reset`. This is"
                " necessary because algorithms share the same buffer for collecting experiences during"
                " training and for evaluation. Set `steps_per_eval` to a factor of"
                " `horizons_per_env_reset` to avoid this error."
            )
        eval_env_config = eval_env_config or env_config
        conditions = list(stop_conditions or [])
        train_stats = self.step(env_config=env_config)
        while not any([condition(train_stats) for condition in conditions]):
            if steps_per_eval and self.state["algorithm/steps"] % steps_per_eval == 0:
                self.eval(env_config=eval_env_config)
            train_stats = self.step(env_config=env_config)
        return train_stats
