"""Stop conditions for :meth:`rl8_amd.trainers.Trainer.run`.

Mirrors the reference's ``src/rl8/conditions.py`` (``Condition`` :12-25, ``And``
:28-44, ``HitsLowerBound`` :47-71, ``HitsUpperBound`` :74-98, ``Plateaus``
:101-153, ``StopsDecreasing`` :156-193, ``StopsIncreasing`` :196-231): a
condition is any callable ``train_stats -> bool``; training stops as soon as one
of the conditions handed to ``run`` returns ``True``.

"""

from __future__ import annotations

from typing import Protocol, Sequence

from .data import TrainStatKey, TrainStats


class Condition(Protocol):
    """Anything callable on the latest train stats that says "stop now"."""

    def __call__(self, train_stats: TrainStats, /) -> bool: ...


class And:
    """True when every one of ``conditions`` is true. All of them are evaluated on
    every call (stateful conditions keep counting)."""

    conditions: list[Condition]

    def __init__(self, conditions: Sequence[Condition], /) -> None:
        self.conditions = list(conditions)

    def __call__(self, train_stats: TrainStats, /) -> bool:
        results = [condition(train_stats) for condition in self.conditions]
        return all(results)


class _Bound:
    key: TrainStatKey

    def __init__(self, key: TrainStatKey, bound: float, /) -> None:
        self.key = key
        self._bound = bound


class HitsLowerBound(_Bound):
    """True once ``train_stats[key] <= lower_bound``."""

    @property
    def lower_bound(self) -> float:
        return self._bound

    def __call__(self, train_stats: TrainStats, /) -> bool:
        return train_stats[self.key] <= self._bound  # type: ignore[literal-required]


class HitsUpperBound(_Bound):
    """True once ``train_stats[key] >= upper_bound``."""

    @property
    def upper_bound(self) -> float:
        return self._bound

    def __call__(self, train_stats: TrainStats, /) -> bool:
        return train_stats[self.key] >= self._bound  # type: ignore[literal-required]


class _Patience:
    """Counts consecutive calls that did not make progress; true at ``patience``."""

    key: TrainStatKey
    patience: int
    #: Consecutive calls without progress so far.
    losses: int

    def __init__(self, key: TrainStatKey, /, *, patience: int = 5) -> None:
        self.key = key
        self.patience = patience
        self.losses = 0

    def _progressed(self, value: float) -> bool:
        raise NotImplementedError

    def __call__(self, train_stats: TrainStats, /) -> bool:
        value = train_stats[self.key]  # type: ignore[literal-required]
        self.losses = 0 if self._progressed(value) else self.losses + 1
        return self.losses >= self.patience


class Plateaus(_Patience):
    """True after ``patience`` consecutive calls whose value stayed within
    ``rtol`` (relative to the previous value) of the previous value."""

    rtol: float
    #: Value seen by the previous call.
    old_value: float

    def __init__(self, key: TrainStatKey, /, *, patience: int = 5, rtol: float = 1e-3) -> None:
        super().__init__(key, patience=patience)
        self.rtol = rtol
        self.old_value = 0

    def _progressed(self, value: float) -> bool:
        moved = abs(value - self.old_value) > self.rtol * abs(self.old_value)
        self.old_value = value
        return moved


class StopsDecreasing(_Patience):
    """True after ``patience`` consecutive calls that set no new minimum."""

    #: Smallest value seen.
    min_: float

    def __init__(self, key: TrainStatKey, /, *, patience: int = 5) -> None:
        super().__init__(key, patience=patience)
        self.min_ = float("inf")

    def _progressed(self, value: float) -> bool:
        if value < self.min_:
            self.min_ = value
            return True
        return False


class StopsIncreasing(_Patience):
    """True after ``patience`` consecutive calls that set no new maximum."""

    #: Largest value seen.
    max_: float

    def __init__(self, key: TrainStatKey, /, *, patience: int = 5) -> None:
        super().__init__(key, patience=patience)
        self.max_ = float("-inf")

    def _progressed(self, value: float) -> bool:
        if value > self.max_:
            self.max_ = value
            return True
        return False
