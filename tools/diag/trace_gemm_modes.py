"""How far the iteration-1 StepStats of the minibatched discrete trace move with the
GEMM used for the towers (eager rocBLAS / fp32 MFMA / bf16 planes / fp16 planes):
the spread between arithmetic-equivalent trajectories is the floor any tolerance on
those averages has to respect.  Run from the repo root on a GPU box."""

import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import test_algorithm_gpu as t  # noqa: E402
from rl8_amd.env import DiscreteDummyEnv  # noqa: E402
from rl8_amd.nn import fused_mlp  # noqa: E402

g = dict(np.load(os.path.join(ROOT, "tests", "golden", "trace_ff_discrete_minibatch.npz"), allow_pickle=True))
keys = [str(k) for k in g["step_stat_keys"]]
result = {"golden": {f"it{it}": dict(zip(keys, map(float, g[f"it{it}_step_stats"]))) for it in range(2)}}
for mode in ("eager", "f32", "f16"):  # (round 2 also ran "split", the bf16-plane forward / data gradient removed in round 3)
    fused_mlp.ENABLED = mode != "eager"
    fused_mlp.FORWARD_GEMM = mode if mode != "eager" else "f16"
    fused_mlp.BACKWARD_GEMM = mode if mode != "eager" else "f16"
    algo = t.build_from_trace(g, DiscreteDummyEnv, sgd_minibatch_size=256, entropy_coeff=1e-2, dual_clip_param=5.0,
                              horizons_per_env_reset=2)
    out = {}
    for it in range(2):
        t.inject(algo, g, it)
        algo.collect()
        stats = algo.step()
        out[f"it{it}"] = {k: float(stats[k]) for k in keys}
        sd = algo.policy.model.state_dict()
        out[f"it{it}_weight_max_abs_dev"] = max(float(np.abs(v.cpu().numpy() - g[f"it{it}_final_{k}"]).max())
                                                for k, v in sd.items())
    result[mode] = out
print(json.dumps(result, indent=1))
