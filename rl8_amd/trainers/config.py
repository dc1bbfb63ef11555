"""Build a trainer from a JSON / YAML file (``src/rl8/trainers/config.py``:
``_import`` :16-25, ``TrainConfig`` :28-144)."""

from __future__ import annotations

import importlib
import json
import pathlib
from dataclasses import dataclass, field
from typing import Any

from ..algorithms import AlgorithmConfig, RecurrentAlgorithmConfig
from ..env import EnvFactory
from ._trainers import RecurrentTrainer, Trainer

#: ``algorithm_config`` entries given as dotted paths in a file.
IMPORTED_FIELDS = ("model_cls", "distribution_cls", "optimizer_cls")


def import_object(name: str) -> Any:
    """``"package.module.Attr"`` -> the object. The longest importable module
    prefix is imported, the rest is attribute access."""
    parts = name.split(".")
    for cut in range(len(parts), 0, -1):
        try:
            obj: Any = importlib.import_module(".".join(parts[:cut]))
        except (ModuleNotFoundError, ValueError):
            continue
        try:
            for attr in parts[cut:]:
                obj = getattr(obj, attr)
        except AttributeError as e:
            raise ImportError(f"Could not dynamically import {name}.") from e
        return obj
    raise ImportError(f"Could not dynamically import {name}.")


@dataclass
class TrainConfig:
    """Environment class + algorithm config (+ ``recurrent``) -> trainer.

    ::

        # config.yaml
        env_cls: rl8_amd.env.DiscreteDummyEnv
        algorithm_config:
            horizon: 8
            gamma: 1

        TrainConfig.from_file("config.yaml").build().run()
    """

    #: Environment class (or factory) the algorithm is built with.
    env_cls: EnvFactory

    #: Keyword arguments of ``AlgorithmConfig`` / ``RecurrentAlgorithmConfig``.
    algorithm_config: dict[str, Any] = field(default_factory=dict)

    #: Build the recurrent variant.
    recurrent: bool = False

    def build(self) -> Trainer | RecurrentTrainer:
        if self.recurrent:
            return RecurrentTrainer(RecurrentAlgorithmConfig(**self.algorithm_config).build(self.env_cls))
        return Trainer(AlgorithmConfig(**self.algorithm_config).build(self.env_cls))

    @classmethod
    def from_file(cls, path: str | pathlib.Path) -> "TrainConfig":
        """Read a ``.json`` or ``.yaml`` file. ``env_cls`` (required) and the
        ``model_cls`` / ``distribution_cls`` / ``optimizer_cls`` entries of
        ``algorithm_config`` are dotted paths into installed packages.

        Raises:
            ValueError: neither ``.json`` nor ``.yaml``.
            RuntimeError: no ``env_cls`` in the file.
            ImportError: a dotted path does not resolve.
        """
        p = pathlib.Path(path)
        if p.suffix == ".json":
            data = json.loads(p.read_text())
        elif p.suffix == ".yaml":
            import yaml

            data = yaml.safe_load(p.read_text())
        else:
            raise ValueError("Config must be a JSON or YAML file")
        if "env_cls" not in data:
            raise RuntimeError(f"{cls.__name__} config {path} must contain `env_cls`")
        data["env_cls"] = import_object(data["env_cls"])
        algorithm_config = data.get("algorithm_config") or {}
        for key in IMPORTED_FIELDS:
            if key in algorithm_config:
                algorithm_config[key] = import_object(algorithm_config[key])
        if "algorithm_config" in data:
            data["algorithm_config"] = algorithm_config
        return cls(**data)
