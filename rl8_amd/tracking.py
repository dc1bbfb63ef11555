"""Experiment tracking behind the trainers.

The reference logs straight to MLflow (``src/rl8/trainers/_base.py:42,100,199``,
``src/rl8/__main__.py:78-101``). MLflow is optional here: :func:`default_tracker`
returns an :class:`MLflowTracker` when the package is importable and an
in-memory :class:`MemoryTracker` otherwise; :class:`JSONLTracker` appends one
line per call to a file. All of them take the same two calls the reference
makes -- ``log_params`` once, ``log_metrics(stats, step=env_steps)`` per
iteration.

"""

from __future__ import annotations

import json
import os
from typing import Any, Mapping, Protocol


class Tracker(Protocol):
    def log_params(self, params: Mapping[str, Any], /) -> None: ...

    def log_metrics(self, metrics: Mapping[str, float], /, *, step: int) -> None: ...


class MemoryTracker:
    """Keeps everything in two lists (tests, notebooks, no-MLflow installs)."""

    def __init__(self) -> None:
        self.params: dict[str, Any] = {}
        self.metrics: list[tuple[int, dict[str, float]]] = []

    def log_params(self, params: Mapping[str, Any], /) -> None:
        self.params.update(params)

    def log_metrics(self, metrics: Mapping[str, float], /, *, step: int) -> None:
        self.metrics.append((step, dict(metrics)))


class JSONLTracker(MemoryTracker):
    """:class:`MemoryTracker` that also appends each call as a JSON line."""

    def __init__(self, path: str | os.PathLike[str]) -> None:
        super().__init__()
        self.path = os.fspath(path)

    def _append(self, record: dict[str, Any]) -> None:
        with open(self.path, "a") as f:
            f.write(json.dumps(record, default=str) + "\n")

    def log_params(self, params: Mapping[str, Any], /) -> None:
        super().log_params(params)
        self._append({"params": dict(params)})

    def log_metrics(self, metrics: Mapping[str, float], /, *, step: int) -> None:
        super().log_metrics(metrics, step=step)
        self._append({"step": step, "metrics": dict(metrics)})


class MLflowTracker:
    """Forwards to the active MLflow run, exactly the reference's calls."""

    def __init__(self) -> None:
        import mlflow  # noqa: F401  (raises ImportError when absent)

    def log_params(self, params: Mapping[str, Any], /) -> None:
        import mlflow

        mlflow.log_params(dict(params))

    def log_metrics(self, metrics: Mapping[str, float], /, *, step: int) -> None:
        import mlflow

        mlflow.log_metrics(dict(metrics), step=step)


def default_tracker() -> Tracker:
    try:
        return MLflowTracker()
    except ImportError:
        return MemoryTracker()
