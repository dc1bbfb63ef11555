"""Iteration-1 StepStats of the minibatched discrete config (64 envs x 32 steps, 8 minibatches
x 4 SGD iterations per update) under the four arithmetic-equivalent tower evaluations, over
several seeds: how far 64 Adam steps carry rounding-level differences, per mode, measured
against the fp32-MFMA run of the same seed."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from rl8_amd import AlgorithmConfig  # noqa: E402
from rl8_amd.env import DiscreteDummyEnv  # noqa: E402
from rl8_amd.nn import fused_mlp  # noqa: E402

KEYS = ("losses/total", "losses/vf", "losses/policy", "monitors/kl_div")


def run(mode, seed):
    fused_mlp.ENABLED = mode != "eager"
    fused_mlp.FORWARD_GEMM = fused_mlp.BACKWARD_GEMM = mode if mode != "eager" else "f16"
    torch.manual_seed(seed)
    algo = AlgorithmConfig(num_envs=64, horizon=32, sgd_minibatch_size=256, entropy_coeff=1e-2, dual_clip_param=5.0,
                           horizons_per_env_reset=2).build(DiscreteDummyEnv)
    out = []
    shuffle = torch.Generator().manual_seed(1000 + seed)
    for _ in range(2):
        # (the same minibatch permutations for every mode: CUDA's randperm stream is shared with other draws)
        algo.injected_permutations = [torch.randperm(64 * 32, generator=shuffle) for _ in range(4)]
        algo.collect()
        actions = algo.buffer["actions"].clone()
        stats = algo.step()
        out.append(({k: float(stats[k]) for k in KEYS}, actions))
    return out


if os.environ.get("DRIFT_SELF_CHECK"):  # the same mode twice: must be bit-identical
    for mode in ("f32", "f16"):
        a, b = run(mode, 0), run(mode, 0)
        print(mode, "it0", a[0][0], b[0][0], "it1", a[1][0], b[1][0], torch.equal(a[1][1], b[1][1]))
    sys.exit(0)

MODES = ("f32", "eager", "f16")
pairs = {}
for seed in range(8):
    runs = {mode: run(mode, seed) for mode in MODES}
    for i, a in enumerate(MODES):
        for b in MODES[i + 1:]:
            rec = pairs.setdefault(f"{a} vs {b}", {"it0": [], "it1": [], "it1_actions_equal": []})
            for it in range(2):
                x, y = runs[a][it][0]["losses/total"], runs[b][it][0]["losses/total"]
                rec[f"it{it}"].append(abs(x - y) / abs(y))
            rec["it1_actions_equal"].append(bool(torch.equal(runs[a][1][1], runs[b][1][1])))
summary = {}
for name, rec in pairs.items():
    summary[name] = {"losses/total relative difference, median over 8 seeds": {it: sorted(rec[it])[4] for it in ("it0", "it1")},
                     "max": {it: max(rec[it]) for it in ("it0", "it1")}, "it1_actions_equal": sum(rec["it1_actions_equal"])}
print(json.dumps(summary, indent=1))
