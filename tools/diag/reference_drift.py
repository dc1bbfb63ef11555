"""How far the REFERENCE's own StepStats move when only the GEMM summation order
changes (MKL thread count 1 / 2 / 4 / 8): the drift band that applies to any
implementation after tens of Adam steps. Build container only (imports the reference
through tests/golden/_stubs)."""
import json, os, sys
HERE = os.path.dirname(os.path.abspath(__file__)); REPO = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [os.path.join(REPO, "tests", "golden", "_stubs"), REPO, "/root/reference/src", "/root/reference"]
import torch
from rl8 import AlgorithmConfig
from rl8.env import DiscreteDummyEnv, ContinuousDummyEnv
from rl8.distributions import SquashedNormal

CASES = {
    "ff_discrete": (DiscreteDummyEnv, {}),
    "ff_discrete_minibatch": (DiscreteDummyEnv, dict(sgd_minibatch_size=256, entropy_coeff=1e-2, dual_clip_param=5.0, horizons_per_env_reset=2)),
    "ff_continuous_squashed": (ContinuousDummyEnv, dict(distribution_cls=SquashedNormal)),
    "ff_continuous_normal": (ContinuousDummyEnv, dict(entropy_coeff=1e-2)),
}
out = {}
for name, (env, kw) in CASES.items():
    runs = {}
    for threads in (1, 2, 4, 8):
        torch.set_num_threads(threads)
        torch.manual_seed(42)
        algo = AlgorithmConfig(num_envs=64, horizon=32, device="cpu", **kw).build(env)
        stats = []
        for it in range(2):
            algo.collect()
            s = algo.step()
            stats.append({k: v for k, v in s.items() if k.startswith(("losses", "monitors"))})
        runs[threads] = stats
    spread = {}
    for it in range(2):
        for k in runs[1][it]:
            vals = [runs[t][it][k] for t in runs]
            ref = max(abs(v) for v in vals) or 1.0
            spread[f"it{it} {k}"] = {"min": min(vals), "max": max(vals), "rel_spread": (max(vals) - min(vals)) / ref}
    out[name] = spread
    print(name, {k: f"{v['rel_spread']:.2e}" for k, v in spread.items()}, flush=True)
json.dump(out, open(os.path.join(REPO, "profiles", "r02_reference_drift.json"), "w"), indent=1)
