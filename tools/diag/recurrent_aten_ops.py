"""Which torch (aten) kernels run inside one recurrent collect() + step(), with shapes and callers.

    python tools/diag/recurrent_aten_ops.py [--num-envs 8192] [--horizon 256]

The HIP kernels of the library are listed by rocprofv3; this names what is left around them (copies, fills,
reductions) so that each can be traced to the host line that issues it.
"""
from __future__ import annotations

import argparse
import sys
from pathlib import Path

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))

from rl8_amd import RecurrentAlgorithmConfig  # noqa: E402
from rl8_amd.env import DiscreteDummyEnv  # noqa: E402


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--num-envs", type=int, default=8192)
    ap.add_argument("--horizon", type=int, default=256)
    args = ap.parse_args()
    torch.manual_seed(0)
    algo = RecurrentAlgorithmConfig(num_envs=args.num_envs, horizon=args.horizon).build(DiscreteDummyEnv)
    algo.collect()
    algo.step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
        algo.collect()
        algo.step()
        torch.cuda.synchronize()
    print(prof.key_averages(group_by_input_shape=True).table(sort_by="cuda_time_total", row_limit=40, max_name_column_width=50,
                                                             max_shapes_column_width=60))
    print(prof.key_averages(group_by_stack_n=6).table(sort_by="cuda_time_total", row_limit=30, max_name_column_width=40,
                                                      max_src_column_width=90))


if __name__ == "__main__":
    main()
