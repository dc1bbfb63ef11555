"""rollout_stats on rows nobody has touched since they were written.

The microbench replays the kernel over the same 268 MB, which the 256 MB
Infinity Cache partly holds; inside collect() the rows are as cold as HBM
gets.  Here 1 GiB is read, or written, between launches and each launch is timed on
its own with events: after a write the cache is full of dirty lines whose
write-back is charged to whoever evicts them, which is the state collect()
leaves behind the last rollout step.

    python tools/diag/stats_cold.py [--envs 1048576] [--horizon 32] [--rounds 20]
"""
from __future__ import annotations

import argparse
import statistics
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))

from rl8_amd import hip  # noqa: E402


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=1 << 20)
    ap.add_argument("--horizon", type=int, default=32)
    ap.add_argument("--rounds", type=int, default=20)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    n, h = args.envs, args.horizon
    rewards = torch.randn(h + 1, n, device=dev)
    rdr = torch.randn(h + 1, n, device=dev)
    evict = torch.empty(1 << 28, dtype=torch.float32, device=dev)
    r3, d3 = rewards.T.unsqueeze(-1), rdr.T.unsqueeze(-1)
    flat = rewards.view(-1)[: n * h]
    both = torch.stack([rewards[:h], rdr[:h]])
    ops = {
        "rollout_stats": lambda: hip.rollout_stats(r3, d3),
        "torch.sum(134 MB)": lambda: flat.sum(),
        "torch.sum(268 MB)": lambda: both.sum(),
    }
    for name, op in ops.items():
        run(name, op, evict, args.rounds, 8.0 * n * h / 1e9 if "134" not in name else 4.0 * n * h / 1e9)


def run(name, op, evict, rounds, gb) -> None:
    for cold in ("warm", "cold after 1 GiB read", "cold after 1 GiB written"):
        times = []
        for i in range(rounds + 3):
            if cold.endswith("read"):
                evict.sum()
            elif cold.endswith("written"):
                evict.fill_(float(i))
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            op()
            b.record()
            torch.cuda.synchronize()
            if i >= 3:
                times.append(a.elapsed_time(b) * 1e3)
        med = statistics.median(times)
        print(f"{name:20s} {cold:25s}: median {med:7.1f} us  min {min(times):7.1f} us  "
              f"{gb / (med * 1e-6) / 1e3:5.2f} TB/s")


if __name__ == "__main__":
    main()
