import os, sys, torch, numpy as np
sys.path.insert(0, "/root/repo")
from rl8_amd import AlgorithmConfig
from rl8_amd.envs import Pendulum
res = {}
for planes in ("f16", "bf16", "f16!"):
    for name in ("RL8_WGRAD_PLANES", "RL8_WGRAD_GATE_PLANES"):
        if planes == "f16": os.environ.pop(name, None)
        else: os.environ[name] = planes
    out = []
    for seed in range(8):
        torch.manual_seed(seed)
        algo = AlgorithmConfig(horizon=128, num_envs=4096, horizons_per_env_reset=4).build(Pendulum)
        for _ in range(41):
            last = algo.collect()["returns/mean"]; algo.step()
        out.append(round(last, 1))
    res[planes] = out
    print(planes, out, "mean", round(float(np.mean(out)), 1), "std", round(float(np.std(out)), 1), flush=True)
