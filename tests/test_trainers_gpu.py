"""N3 on the GPU: the reference's tests/test_trainers.py behaviours for both
trainers, TrainConfig.build, the CLI end to end."""

import json
import pickle

import pytest
import torch

pytestmark = pytest.mark.gpu

from rl8_amd import AlgorithmConfig, RecurrentAlgorithmConfig, RecurrentTrainer, TrainConfig, Trainer  # noqa: E402
from rl8_amd.conditions import HitsUpperBound  # noqa: E402
from rl8_amd.env import DiscreteDummyEnv  # noqa: E402
from rl8_amd.tracking import MemoryTracker  # noqa: E402

NUM_ENVS = 64
HORIZON = 32
HORIZONS_PER_ENV_RESET = 2
KINDS = [(AlgorithmConfig, Trainer), (RecurrentAlgorithmConfig, RecurrentTrainer)]


def build(config_cls, trainer_cls, **kw):
    algo = config_cls(num_envs=NUM_ENVS, horizon=HORIZON, horizons_per_env_reset=HORIZONS_PER_ENV_RESET, **kw)
    trainer = trainer_cls(algo.build(DiscreteDummyEnv), tracker=MemoryTracker())
    assert trainer.state == {"algorithm/collects": 0, "algorithm/steps": 0, "env/steps": 0}
    return trainer


@pytest.mark.parametrize("config_cls,trainer_cls", KINDS)
def test_trainer_eval(config_cls, trainer_cls):
    trainer = build(config_cls, trainer_cls)
    stats = trainer.eval()
    assert trainer.state["algorithm/collects"] == HORIZONS_PER_ENV_RESET
    assert trainer.state["algorithm/steps"] == 0
    assert all(k.startswith("eval/") for k in stats) and "eval/returns/mean" in stats
    assert stats["eval/env/steps"] == HORIZONS_PER_ENV_RESET * NUM_ENVS * HORIZON
    assert trainer.tracker.metrics[-1] == (0, stats)
    assert trainer.tracker.params["num_envs"] == NUM_ENVS


@pytest.mark.parametrize("config_cls,trainer_cls", KINDS)
def test_trainer_eval_runtime_error(config_cls, trainer_cls):
    trainer = build(config_cls, trainer_cls)
    trainer.step()
    with pytest.raises(RuntimeError, match="horizons_per_env_reset"):
        trainer.eval()


@pytest.mark.parametrize("config_cls,trainer_cls", KINDS)
def test_trainer_step(config_cls, trainer_cls):
    trainer = build(config_cls, trainer_cls)
    stats = trainer.step()
    assert trainer.state["algorithm/collects"] == 1
    assert trainer.state["algorithm/steps"] == 1
    assert trainer.state["env/steps"] == NUM_ENVS * HORIZON
    for key in ("memory/free", "returns/mean", "losses/total", "algorithm/steps", "env/steps"):
        assert key in stats
    assert trainer.tracker.metrics[-1] == (NUM_ENVS * HORIZON, stats)


@pytest.mark.parametrize("config_cls,trainer_cls", KINDS)
def test_trainer_run(config_cls, trainer_cls):
    trainer = build(config_cls, trainer_cls)
    trainer.run(
        steps_per_eval=HORIZONS_PER_ENV_RESET,
        stop_conditions=[HitsUpperBound("algorithm/collects", 2 * HORIZONS_PER_ENV_RESET + 1)],
    )
    assert trainer.state["algorithm/collects"] == 2 * HORIZONS_PER_ENV_RESET + 1
    assert trainer.state["algorithm/steps"] == HORIZONS_PER_ENV_RESET + 1


@pytest.mark.parametrize("config_cls,trainer_cls", KINDS)
def test_trainer_run_value_error(config_cls, trainer_cls):
    trainer = build(config_cls, trainer_cls)
    with pytest.raises(ValueError, match="steps_per_eval"):
        trainer.run(steps_per_eval=1)


def test_eval_config_for_an_env_that_is_reset_once():
    algo = AlgorithmConfig(num_envs=NUM_ENVS, horizon=HORIZON, horizons_per_env_reset=-1).build(DiscreteDummyEnv)
    trainer = Trainer(algo, tracker=MemoryTracker())
    with pytest.raises(ValueError, match="eval environment config"):
        trainer.run(steps_per_eval=1, eval_env_config={"bounds": 10.0})
    trainer.step()
    with pytest.raises(ValueError, match="eval environment config"):
        trainer.eval(env_config={"bounds": 10.0})
    trainer.eval()  # without a config it is allowed at any time


def test_train_config_build_and_cli(tmp_path):
    trainer = TrainConfig(DiscreteDummyEnv, {"num_envs": 32, "horizon": 8}).build()
    assert isinstance(trainer, Trainer) and trainer.algorithm.hparams.horizon == 8
    trainer = TrainConfig(DiscreteDummyEnv, {"num_envs": 32, "horizon": 8, "seq_len": 4, "seqs_per_state_reset": 2},
                          recurrent=True).build()
    assert isinstance(trainer, RecurrentTrainer)

    from rl8_amd.__main__ import main

    config = tmp_path / "config.json"
    config.write_text(json.dumps({"env_cls": "rl8_amd.envs.CartPole",
                                  "algorithm_config": {"num_envs": 256, "horizon": 16}}))
    metrics = tmp_path / "metrics.jsonl"
    save = tmp_path / "out"
    assert main(["train", "-f", str(config), "--max-steps", "3", "--save", str(save), "--metrics", str(metrics)]) == 0
    lines = [json.loads(line) for line in metrics.read_text().splitlines()]
    assert "params" in lines[0] and [rec["metrics"]["algorithm/steps"] for rec in lines[1:]] == [1, 2, 3]
    with open(save / "policy.pkl", "rb") as f:
        policy = pickle.load(f)
    obs = torch.zeros(4, 1, 5, device=policy.device)
    from rl8_amd.data import DataKeys
    from rl8_amd.tensordict import TensorDict

    out = policy.sample(TensorDict({DataKeys.OBS: obs}, batch_size=[4, 1]), deterministic=True, return_actions=True)
    assert out[DataKeys.ACTIONS].shape == (4, 1)
