"""Times the UNMODIFIED reference (imported from /root/reference through the
container stubs of tests/golden/_stubs) on BASELINE configs[0]: DiscreteDummyEnv,
num_envs=8192, horizon=32, all AlgorithmConfig defaults, device="cpu".

Build container only: the reference cannot travel to the GPU box, so this number
is a labelled side note (`reference_real`, host named) next to bench.py's on-box
`cpu_baseline` -- SURVEY 8(d)(2). Protocol: 3 warm-up iterations, then >= 10 timed
`collect(); step()` iterations; median and best reported.

    python tools/time_reference.py [--iters 10] [--threads 8] > profiles/r02_reference_real.json
"""
import argparse
import json
import os
import platform
import statistics
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
REFERENCE = "/root/reference"
sys.path[:0] = [os.path.join(REPO, "tests", "golden", "_stubs"), REPO, os.path.join(REFERENCE, "src"), REFERENCE]

import torch  # noqa: E402

p = argparse.ArgumentParser()
p.add_argument("--iters", type=int, default=10)
p.add_argument("--warmup", type=int, default=3)
p.add_argument("--threads", type=int, default=os.cpu_count())
p.add_argument("--num-envs", type=int, default=8192)
p.add_argument("--horizon", type=int, default=32)
args = p.parse_args()
torch.set_num_threads(args.threads)

from rl8 import AlgorithmConfig  # noqa: E402
from rl8.env import DiscreteDummyEnv  # noqa: E402

torch.manual_seed(0)
algo = AlgorithmConfig(num_envs=args.num_envs, horizon=args.horizon, device="cpu").build(DiscreteDummyEnv)
for _ in range(args.warmup):
    algo.collect()
    algo.step()
times, collect_ms, step_ms = [], [], []
for _ in range(args.iters):
    t0 = time.perf_counter()
    c = algo.collect()
    s = algo.step()
    times.append(time.perf_counter() - t0)
    collect_ms.append(c["profiling/collect_ms"])
    step_ms.append(s["profiling/step_ms"])
n = args.num_envs * args.horizon
cpu = ""
try:
    cpu = next(line.split(":", 1)[1].strip() for line in open("/proc/cpuinfo") if line.startswith("model name"))
except Exception:  # noqa: BLE001
    pass
print(json.dumps({
    "kind": "reference_real",
    "what": "unmodified theOGognf/rl8 (src/rl8) through tests/golden/_stubs, AlgorithmConfig defaults, device=cpu",
    "workload": f"DiscreteDummyEnv num_envs={args.num_envs} horizon={args.horizon} (BASELINE configs[0])",
    "host": f"build container, {platform.machine()}, {cpu}",
    "os_cpu_count": os.cpu_count(),
    "torch_threads": torch.get_num_threads(),
    "torch": torch.__version__,
    "iters": args.iters,
    "warmup": args.warmup,
    "transitions_per_sec_median": n / statistics.median(times),
    "transitions_per_sec_best": n / min(times),
    "policy_updates_per_sec_median": 1.0 / statistics.median(times),
    "ms_per_iteration_median": statistics.median(times) * 1e3,
    "collect_ms_median": statistics.median(collect_ms),
    "step_ms_median": statistics.median(step_ms),
}, indent=1))
