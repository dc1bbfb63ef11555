"""Which kernels run inside one feed-forward collect() + step() of the headline configuration (2^20 envs x 32), torch's
own (copies, fills, reductions) beside the library's, by total device time.

    python tools/diag/feedforward_aten_ops.py [--num-envs 1048576] [--horizon 32] [--env discrete|continuous|cartpole|mountain_car|pendulum] [--minibatches K]
"""
from __future__ import annotations

import argparse
import sys
from pathlib import Path

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))

from rl8_amd import AlgorithmConfig  # noqa: E402
from rl8_amd.env import ContinuousDummyEnv, DiscreteDummyEnv  # noqa: E402


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--num-envs", type=int, default=1 << 20)
    ap.add_argument("--horizon", type=int, default=32)
    ap.add_argument("--env", default="discrete")
    ap.add_argument("--minibatches", type=int, default=1, help="shuffled minibatches per SGD iteration")
    args = ap.parse_args()
    torch.manual_seed(0)
    if args.env == "cartpole":
        from rl8_amd.envs.cartpole import CartPole as env
    elif args.env == "mountain_car":
        from rl8_amd.envs import MountainCar as env
    elif args.env == "pendulum":
        from rl8_amd.envs import Pendulum as env
    else:
        env = DiscreteDummyEnv if args.env == "discrete" else ContinuousDummyEnv
    size = None if args.minibatches == 1 else args.num_envs * args.horizon // args.minibatches
    algo = AlgorithmConfig(num_envs=args.num_envs, horizon=args.horizon, sgd_minibatch_size=size).build(env)
    algo.collect()
    algo.step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        algo.collect()
        algo.step()
        torch.cuda.synchronize()
    print(prof.key_averages(group_by_input_shape=True).table(sort_by="cuda_time_total", row_limit=45, max_name_column_width=50,
                                                             max_shapes_column_width=60))


if __name__ == "__main__":
    main()
