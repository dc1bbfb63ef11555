"""``python -m rl8_amd train -f config.yaml`` (``src/rl8/__main__.py:21-99``).

Builds a trainer from a config file, trains for ``--max-steps`` trainer steps and
optionally pickles the policy. With MLflow installed the run is logged there as
the reference does; without it metrics go to ``--metrics`` (JSON lines) or stay in
memory.

"""

from __future__ import annotations

import argparse
import pathlib
from typing import Any

from .conditions import HitsUpperBound
from .tracking import JSONLTracker
from .trainers import TrainConfig


def qualified_name(obj: Any) -> str:
    module = getattr(obj, "__module__", None)
    name = getattr(obj, "__qualname__", getattr(obj, "__name__", repr(obj)))
    return name if module in (None, "builtins") else f"{module}.{name}"


def build_parser() -> argparse.ArgumentParser:
    parser = argparse.ArgumentParser(prog="rl8_amd")
    sub = parser.add_subparsers(dest="command", required=True)
    train = sub.add_parser("train", help="Train a policy from a config file with the trainer interface.")
    train.add_argument("-f", "--file", type=pathlib.Path, required=True, help="Train config (.json / .yaml).")
    train.add_argument("--experiment-name", default=None,
                       help="MLflow experiment name; defaults to the environment's qualified name.")
    train.add_argument("--max-steps", type=int, default=100, help="Trainer steps before stopping.")
    train.add_argument("--save", default=None, help="Directory to save the trained policy to.")
    train.add_argument("--steps-per-eval", type=int, default=None, help="Trainer steps between evaluations.")
    train.add_argument("--metrics", default=None, help="Append metrics to this JSON-lines file (no MLflow).")
    return parser


def train(args: argparse.Namespace) -> int:
    config = TrainConfig.from_file(args.file)
    experiment_name = args.experiment_name or qualified_name(config.env_cls)
    try:
        import mlflow
    except ImportError:
        mlflow = None
    if mlflow is not None:
        experiment = mlflow.set_experiment(experiment_name)
        print(f"Logging runs under MLflow experiment {experiment.name}")
        mlflow.start_run()
    trainer = config.build()
    if mlflow is None and args.metrics:
        trainer.tracker = JSONLTracker(args.metrics)
        trainer.tracker.log_params(trainer.algorithm.params)
    stats = trainer.run(
        steps_per_eval=args.steps_per_eval,
        stop_conditions=[HitsUpperBound("algorithm/steps", args.max_steps)],
    )
    print(f"{experiment_name}: {stats['algorithm/steps']} steps, {stats['env/steps']} env steps,"
          f" returns/mean {stats['returns/mean']:.4f}")
    if args.save:
        out = pathlib.Path(args.save)
        out.mkdir(exist_ok=True)
        trainer.algorithm.policy.model.eval()
        trainer.algorithm.policy.save(out / "policy.pkl")
        print(f"Saved the policy to {out / 'policy.pkl'}")
    if mlflow is not None:
        mlflow.end_run()
    return 0


def main(argv: None | list[str] = None) -> int:
    args = build_parser().parse_args(argv)
    if args.command == "train":
        return train(args)
    return 2


if __name__ == "__main__":
    raise SystemExit(main())
