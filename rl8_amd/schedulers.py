"""Host-side schedules of the learning rate and the entropy coefficient, keyed
on the number of environment transitions.

Scalars only -- nothing here touches the device. Behaviour follows the
reference's ``src/rl8/schedulers.py`` (``EntropyScheduler`` :121-171,
``LRScheduler`` :174-232) because ``Algorithm.step()`` calls both after every
update (``src/rl8/algorithms/_feedforward.py:599-600``).

"""

from __future__ import annotations

from typing import Literal, Protocol, Sequence

import numpy as np
import torch.optim as optim

ScheduleKind = Literal["interp", "step"]
Schedule = Sequence[Sequence[float]]


class Scheduler(Protocol):
    def step(self, count: int, /) -> float:
        ...


class ConstantScheduler:
    """Always the same value."""

    def __init__(self, value: float, /) -> None:
        self.value = value

    def step(self, _: int, /) -> float:
        return self.value


def _check_schedule(owner: str, schedule: Schedule) -> None:
    if schedule[0][0]:
        raise ValueError(
            f"{owner} `schedule` arg's first step value (i.e., `schedule[0][0]`)"
            " must be `0` to indicate the scheduler's initial value."
        )


class InterpScheduler:
    """Piecewise-linear interpolation between ``(count, value)`` knots."""

    def __init__(self, schedule: Schedule, /) -> None:
        _check_schedule(type(self).__name__, schedule)
        self.x = [knot[0] for knot in schedule]
        self.y = [knot[1] for knot in schedule]

    def step(self, count: int, /) -> float:
        return float(np.interp(count, self.x, self.y))


class StepScheduler:
    """Jump to a knot's value once ``count`` reaches it, and hold."""

    def __init__(self, schedule: Schedule, /) -> None:
        _check_schedule(type(self).__name__, schedule)
        self.schedule = schedule

    def step(self, count: int, /) -> float:
        value = 0.0
        for threshold, v in self.schedule:
            if count >= threshold:
                value = v
        return value


def _make(owner: str, schedule: Schedule, kind: str) -> Scheduler:
    if kind == "interp":
        return InterpScheduler(schedule)
    if kind == "step":
        return StepScheduler(schedule)
    raise ValueError(f"{owner} only supports kinds `interp` and `step`.")


class EntropyScheduler:
    """Entropy-coefficient schedule; constant when no schedule is given."""

    def __init__(
        self,
        coeff: float,
        /,
        *,
        schedule: None | Schedule = None,
        kind: ScheduleKind = "step",
    ) -> None:
        self.scheduler: Scheduler = (
            ConstantScheduler(coeff)
            if schedule is None
            else _make("Entropy scheduler", schedule, kind)
        )
        self.coeff = self.step(0)

    def step(self, count: int, /) -> float:
        self.coeff = self.scheduler.step(count)
        return self.coeff


class LRScheduler:
    """Learning-rate schedule; leaves the optimizer alone when no schedule is
    given (``coeff`` then reads 0, as in the reference)."""

    def __init__(
        self,
        optimizer: optim.Optimizer,
        /,
        *,
        schedule: None | Schedule = None,
        kind: ScheduleKind = "step",
    ) -> None:
        self.optimizer = optimizer
        self.scheduler: Scheduler = (
            ConstantScheduler(0.0)
            if schedule is None
            else _make("Learning rate scheduler", schedule, kind)
        )
        self.coeff = self.step(0)

    def step(self, count: int, /) -> float:
        self.coeff = self.scheduler.step(count)
        if not isinstance(self.scheduler, ConstantScheduler):
            for group in self.optimizer.param_groups:
                group["lr"] = self.coeff
        return self.coeff
