"""Host-side helpers of the PPO path: minibatch index generation (``Batcher``),
running loss averages (``StatTracker``), timing and memory probes, spec checks.

The heavy part of the reference's ``Batcher`` (``src/rl8/_utils.py:175-225``) --
one index-gather per buffer leaf per minibatch -- runs as one HIP launch here
(``rl8_gather_minibatch``); index generation stays on the host/torch side.

"""

from __future__ import annotations

import time
from contextlib import contextmanager
from typing import Any, Callable, Generator, Iterable, Literal

import psutil
import torch

from .data import MemoryStats
from .specs import Composite, TensorSpec
from .tensordict import TensorDict


def assert_1d_spec(spec: TensorSpec, /) -> None:
    """Default models and distributions only take 1D specs."""
    assert spec.ndim == 1, (
        f"{spec} is not compatible with default models and distributions. "
        "Default models and distributions do not support tensor specs "
        "that aren't 1D. Tensor specs must have shape ``[N]`` "
        "(where ``N`` is the number of independent elements) to be "
        "compatible with default models and distributions."
    )


def assert_nd_spec(spec: TensorSpec, /) -> None:
    """Every leaf spec must have a non-empty shape."""
    if isinstance(spec, Composite):
        for k in spec:
            assert_nd_spec(spec[k])
        return
    assert spec.ndim >= 1, (
        f"{spec} is not a valid spec. Models and distributions must have specs"
        " that have a non-empty shape. Tensor specs must have shape ``[N,"
        " ...]`` (where ``N`` is the number of independent elements and"
        " ``...`` is any number of additional dimensions)."
    )


def memory_stats(device_type: Literal["cuda", "cpu"], /) -> MemoryStats:
    """Free / total memory of the host or of the current HIP device."""
    if device_type == "cpu":
        vm = psutil.virtual_memory()
        free, total = vm.free, vm.total
    else:
        free, total = torch.cuda.mem_get_info()
    return {
        "memory/free": free,
        "memory/total": total,
        "memory/percent": 100 * (total - free) / total,
    }


@contextmanager
def profile_ms() -> Generator[Callable[[], float], None, None]:
    """Wall-clock milliseconds spent inside the ``with`` block so far."""
    start = time.perf_counter_ns()
    yield lambda: (time.perf_counter_ns() - start) / 1e6


def reduce_stats(x: dict[str, list[float]], /) -> dict[str, float]:
    """Collapse lists of per-collect stats by the operation their key names."""
    out = {}
    for k, v in x.items():
        op = k.split("/")[-1]
        if op == "min":
            out[k] = min(v)
        elif op == "max":
            out[k] = max(v)
        elif op == "mean":
            out[k] = sum(v) / len(v)
        elif op == "std":
            out[k] = (sum(s**2 for s in v) / len(v)) ** 0.5
        else:
            out[k] = sum(v)
    return out


class Batcher:
    """Iterates a 1D tensordict in (optionally shuffled) chunks.

    A fresh permutation is drawn every time iteration starts, as in the
    reference (``__iter__`` is re-entered by each ``enumerate(batcher)``,
    ``src/rl8/algorithms/_feedforward.py:513``). ``permutation_fn`` lets a
    caller (tests injecting the reference's recorded permutations) supply the
    index order.

    """

    def __init__(
        self,
        batch: TensorDict,
        /,
        *,
        batch_size: None | int = None,
        shuffle: bool = False,
        permutation_fn: None | Callable[[int], torch.Tensor] = None,
    ) -> None:
        self.batch = batch
        self.batch_size = batch_size or self.batch.size(0)
        self.shuffle = shuffle
        self.permutation_fn = permutation_fn
        self.indices: tuple[torch.Tensor, ...] = ()
        self.idx = 0

    def draw_indices(self) -> tuple[torch.Tensor, ...]:
        size = self.batch.size(0)
        device = self.batch.device
        if self.shuffle:
            if self.permutation_fn is not None:
                indices = self.permutation_fn(size).to(device)
            else:
                indices = torch.randperm(size, device=device)
        else:
            indices = torch.arange(size, device=device)
        return torch.split(indices, self.batch_size)

    def __iter__(self) -> "Batcher":
        self.idx = 0
        self.indices = self.draw_indices()
        return self

    def __next__(self) -> TensorDict:
        if self.idx < len(self.indices):
            out = self.batch[self.indices[self.idx], ...]
            self.idx += 1
            return out
        raise StopIteration


class CumulativeAverage:
    """Running mean.

    >>> ca = CumulativeAverage()
    >>> ca.update(0.0)
    0.0
    >>> ca.update(2.0)
    1.0

    """

    def __init__(self) -> None:
        self.avg = 0.0
        self.n = 0

    def update(self, value: float, /) -> float:
        self.avg = (value + self.n * self.avg) / (self.n + 1)
        self.n += 1
        return self.avg


class StatTracker:
    """Running means of named values; ``sum_keys`` are first summed across the
    minibatches of one optimizer step and enter their mean only on
    ``reduce=True`` (``src/rl8/_utils.py:259-313``)."""

    def __init__(self, keys: Iterable[str], *, sum_keys: None | Iterable[str] = None) -> None:
        self.cumulative_averages = {k: CumulativeAverage() for k in keys}
        self.sums: dict[str, float] = {k: 0 for k in (sum_keys or [])}

    def items(self) -> dict[str, float]:
        return {k: ca.avg for k, ca in self.cumulative_averages.items()}

    def update(self, data: dict[str, float], /, *, reduce: bool = False) -> None:
        for k in self.sums:
            self.sums[k] += data[k]
        for k in self.cumulative_averages.keys() - self.sums.keys():
            self.cumulative_averages[k].update(data[k])
        if reduce:
            for k in self.sums:
                self.cumulative_averages[k].update(self.sums[k])
                self.sums[k] = 0.0


