"""N3 host logic that needs no GPU: stop conditions (the reference's
tests/test_conditions.py cases), the trackers, TrainConfig.from_file parsing."""

import json

import pytest

from rl8_amd.conditions import And, HitsLowerBound, HitsUpperBound, Plateaus, StopsDecreasing, StopsIncreasing
from rl8_amd.tracking import JSONLTracker, MemoryTracker
from rl8_amd.trainers.config import TrainConfig, import_object


def test_and():
    both = And([HitsUpperBound("returns/mean", 100.0), HitsUpperBound("counting/step_calls", 50.0)])
    assert not both({"returns/mean": 100.0, "counting/step_calls": 1})
    assert both({"returns/mean": 100.0, "counting/step_calls": 50})


def test_and_evaluates_every_condition_each_call():
    plateau = Plateaus("x", patience=2, rtol=0.5)
    both = And([HitsUpperBound("x", 10.0), plateau])
    both({"x": 1.0})
    both({"x": 1.1})  # the first condition is false, the plateau counter still advances
    assert plateau.losses == 1


def test_hits_bounds():
    lower = HitsLowerBound("returns/mean", -100.0)
    assert not lower({"returns/mean": 1}) and lower({"returns/mean": -200.0}) and lower({"returns/mean": -100.0})
    upper = HitsUpperBound("returns/mean", 100.0)
    assert not upper({"returns/mean": 1}) and upper({"returns/mean": 200.0}) and upper({"returns/mean": 100.0})
    assert lower.lower_bound == -100.0 and upper.upper_bound == 100.0


def test_plateaus():
    plateaus = Plateaus("returns/mean", patience=2, rtol=2e-1)
    assert not plateaus({"returns/mean": 1})
    assert not plateaus({"returns/mean": 0.9})
    assert plateaus({"returns/mean": 1})
    assert not plateaus({"returns/mean": 5})  # a jump resets the count
    assert plateaus.losses == 0 and plateaus.old_value == 5


def test_stops_decreasing():
    cond = StopsDecreasing("returns/mean", patience=2)
    assert not cond({"returns/mean": 1})
    assert not cond({"returns/mean": 1.1})
    assert cond({"returns/mean": 1.2})
    assert not cond({"returns/mean": 0.5}) and cond.min_ == 0.5


def test_stops_increasing():
    cond = StopsIncreasing("returns/mean", patience=2)
    assert not cond({"returns/mean": 1})
    assert not cond({"returns/mean": 0.9})
    assert cond({"returns/mean": 0.8})
    assert not cond({"returns/mean": 3}) and cond.max_ == 3


def test_trackers(tmp_path):
    t = MemoryTracker()
    t.log_params({"a": 1})
    t.log_metrics({"m": 2.0}, step=10)
    assert t.params == {"a": 1} and t.metrics == [(10, {"m": 2.0})]
    path = tmp_path / "metrics.jsonl"
    j = JSONLTracker(path)
    j.log_params({"device": "cuda"})
    j.log_metrics({"returns/mean": -1.5}, step=64)
    lines = [json.loads(line) for line in path.read_text().splitlines()]
    assert lines == [{"params": {"device": "cuda"}}, {"step": 64, "metrics": {"returns/mean": -1.5}}]


def test_import_object():
    assert import_object("rl8_amd.env.DiscreteDummyEnv").__name__ == "DiscreteDummyEnv"
    assert import_object("torch.optim.SGD").__name__ == "SGD"
    for bad in ("rl8_amd.env.NoSuchEnv", "no_such_package.Thing", ""):
        with pytest.raises(ImportError):
            import_object(bad)


def test_train_config_from_file(tmp_path):
    from rl8_amd.distributions import SquashedNormal
    from rl8_amd.env import ContinuousDummyEnv

    y = tmp_path / "config.yaml"
    y.write_text(
        "env_cls: rl8_amd.env.ContinuousDummyEnv\n"
        "algorithm_config:\n"
        "    horizon: 8\n"
        "    gamma: 1\n"
        "    distribution_cls: rl8_amd.distributions.SquashedNormal\n"
        "    optimizer_cls: torch.optim.SGD\n"
        "recurrent: true\n"
    )
    config = TrainConfig.from_file(y)
    assert config.env_cls is ContinuousDummyEnv and config.recurrent
    assert config.algorithm_config["horizon"] == 8 and config.algorithm_config["gamma"] == 1
    assert config.algorithm_config["distribution_cls"] is SquashedNormal
    assert config.algorithm_config["optimizer_cls"].__name__ == "SGD"
    j = tmp_path / "config.json"
    j.write_text(json.dumps({"env_cls": "rl8_amd.envs.Pendulum"}))
    config = TrainConfig.from_file(j)
    assert config.env_cls.__name__ == "Pendulum" and config.algorithm_config == {} and not config.recurrent
    bad = tmp_path / "config.toml"
    bad.write_text("")
    with pytest.raises(ValueError, match="JSON or YAML"):
        TrainConfig.from_file(bad)
    j.write_text(json.dumps({"algorithm_config": {}}))
    with pytest.raises(RuntimeError, match="env_cls"):
        TrainConfig.from_file(j)
    j.write_text(json.dumps({"env_cls": "rl8_amd.env.Nope"}))
    with pytest.raises(ImportError):
        TrainConfig.from_file(j)


def test_cli_parser():
    from rl8_amd.__main__ import build_parser, qualified_name
    from rl8_amd.env import DiscreteDummyEnv

    args = build_parser().parse_args(["train", "-f", "c.yaml", "--max-steps", "3", "--steps-per-eval", "1"])
    assert args.command == "train" and args.max_steps == 3 and args.steps_per_eval == 1 and args.save is None
    assert qualified_name(DiscreteDummyEnv) == "rl8_amd.env.DiscreteDummyEnv"
    with pytest.raises(SystemExit):
        build_parser().parse_args([])
