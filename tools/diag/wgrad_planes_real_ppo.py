"""What real PPO gradients do to the fp16-plane weight gradient.

A real run of the headline config (DiscreteDummyEnv, 2^20 x 32, defaults).  At iterations 0, 5 and 20 the first
backward of each tower (policy: two-way categorical, value) is intercepted and dW2 is formed from the SAME saved state
(x, dOut, gate bits, weights) three ways -- fp16 planes unguarded (RL8_WGRAD_GATE_PLANES=f16!), exact bf16 planes
(=bf16), and the shipped default (fp16 under the guard) -- and compared, on 16 x 16 sampled entries, with an fp64
accumulation over all 2^25 rows: error relative to the entry's own sum of |terms| and relative to the tensor's largest
entry.  Also the statistics the guard looks at (the spread of |dOut| over binades) and, for the sampled columns, the
ratio bound / mean |term| that says how many of the planes' 22 bits a typical term keeps.

    python tools/diag/wgrad_planes_real_ppo.py [--num-envs N] [--out profiles/r04_wgrad_planes_real_ppo.json]
"""

import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

import torch  # noqa: E402

from rl8_amd import AlgorithmConfig, hip  # noqa: E402
from rl8_amd.env import DiscreteDummyEnv  # noqa: E402

p = argparse.ArgumentParser()
p.add_argument("--num-envs", type=int, default=1 << 20)
p.add_argument("--horizon", type=int, default=32)
p.add_argument("--iterations", default="0,5,20")
p.add_argument("--out", default="profiles/r04_wgrad_planes_real_ppo.json")
args = p.parse_args()
WATCH = sorted(int(v) for v in args.iterations.split(","))

torch.manual_seed(0)
algo = AlgorithmConfig(num_envs=args.num_envs, horizon=args.horizon).build(DiscreteDummyEnv)
real_backward = hip.mlp_tower_backward
captured = {}
report = {"config": f"DiscreteDummyEnv num_envs={args.num_envs} horizon={args.horizon} defaults", "iterations": {}}


def fp64_entries(x, dout, gate, w1, b1, w3, J, I):
    """dW2[J, I] and the entries' sums of |terms| in fp64 over all rows, plus mean |term| of the columns I."""
    m = x.shape[0]
    w3e = (w3[0] - w3[1]).double() if w3.shape[0] == 2 else w3[0].double()
    got = torch.zeros(len(J), len(I), dtype=torch.float64, device=x.device)
    size = torch.zeros_like(got)
    col_abs = torch.zeros(len(I), dtype=torch.float64, device=x.device)
    for lo in range(0, m, 1 << 22):
        hi = min(m, lo + (1 << 22))
        d = dout[lo:hi, 0].double()
        h = torch.relu(x[lo:hi].double() @ w1[I].double().T + b1[I].double())          # [rows, |I|]
        words = gate[lo:hi][:, (J >> 5)]                                                # [rows, |J|]
        g = ((words >> (J & 31)) & 1).double()
        t = d[:, None] * h
        got += g.T @ t
        size += g.T @ t.abs()
        col_abs += t.abs().sum(0)
    return got * w3e[J][:, None], size * w3e[J].abs()[:, None], col_abs / m


def dout_stats(dout):
    a = dout[:, 0].abs()
    top = float(a.max())
    nz = a > 0
    out = {"rows": int(a.numel()), "zero_share": float((~nz).float().mean()), "max": top,
           "mean_nonzero": float(a[nz].mean()) if bool(nz.any()) else 0.0}
    for bits in (8, 12, 17, 20):
        out[f"share_of_nonzero_below_2^-{bits}_of_max"] = float((nz & (a < top * 2.0 ** -bits)).sum() / nz.sum().clamp(min=1))
    return out


def analyse(tag, it, x, dout, gate, w1, b1, w2, b2, w3, kwargs):
    g = torch.Generator(device="cpu").manual_seed(17 + it)
    J = torch.randperm(256, generator=g)[:16].to(x.device)
    I = torch.randperm(256, generator=g)[:16].to(x.device)
    want, size, col_mean = fp64_entries(x, dout, gate, w1, b1, w3, J, I)
    bound = float(dout.abs().max()) * (b1[I].abs().double() + float(x.abs().max()) * w1[I, 0].abs().double())
    out = {"dout": dout_stats(dout),
           "bound_over_mean_term_log2": [round(float(v), 2) for v in torch.log2(bound / col_mean.clamp(min=1e-300))]}
    calls0, fires0 = hip.wgrad_guard_counts()
    grads = {}
    for mode, env in (("f16_unguarded", "f16!"), ("bf16_exact", "bf16"), ("shipped_guarded", "")):
        os.environ["RL8_WGRAD_GATE_PLANES"] = env
        grads[mode] = real_backward(*[t.clone() if torch.is_tensor(t) else t for t in (x, None, None, dout)],
                                    *kwargs["rest"], **kwargs["kw"])["w2"].double()
    os.environ.pop("RL8_WGRAD_GATE_PLANES", None)
    calls1, fires1 = hip.wgrad_guard_counts()
    out["guard"] = {"consulted": calls1 - calls0, "sent_to_bf16": fires1 - fires0}
    top = float(grads["bf16_exact"].abs().max())
    for mode, gw in grads.items():
        sub = gw[J][:, I]
        rel = ((sub - want).abs() / size.clamp(min=1e-300)).flatten()
        out[mode] = {
            "entrywise_err_over_sum_abs_terms": {"max": float(rel.max()), "p90": float(rel.quantile(0.9)),
                                                 "median": float(rel.median())},
            "max_err_over_largest_entry": float((sub - want).abs().max()) / top,
        }
    out["f16_vs_bf16_whole_tensor_max_diff_over_largest_entry"] = float(
        (grads["f16_unguarded"] - grads["bf16_exact"]).abs().max()) / top
    out["shipped_equals"] = ("f16" if torch.equal(grads["shipped_guarded"], grads["f16_unguarded"]) else
                             "bf16" if torch.equal(grads["shipped_guarded"], grads["bf16_exact"]) else "neither")
    report["iterations"].setdefault(str(it), {})[tag] = out
    print(it, tag, json.dumps(out), flush=True)


current = {"it": -1}


def spy(x, h1, h2, dout, w2t_packed, w3, w1=None, b1=None, **kw):
    it = current["it"]
    tag = "policy_tower" if w3.shape[0] == 2 else "value_tower"
    if it in WATCH and (it, tag) not in captured and kw.get("gate2") is not None and h2 is None:
        captured[(it, tag)] = True
        analyse(tag, it, x, dout, kw["gate2"], w1, b1, kw["w2"], kw["b2"], w3, {"rest": (w2t_packed, w3, w1, b1), "kw": kw})
    return real_backward(x, h1, h2, dout, w2t_packed, w3, w1, b1, **kw)


hip.mlp_tower_backward = spy
for it in range(max(WATCH) + 1):
    current["it"] = it
    algo.collect()
    stats = algo.step()
    if it in WATCH:
        report["iterations"].setdefault(str(it), {})["losses"] = {k: v for k, v in stats.items() if k.startswith(("losses", "monitors"))}
calls, fires = hip.wgrad_guard_counts()
report["guard_lifetime"] = {"consulted": calls, "sent_to_bf16": fires}
os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
json.dump(report, open(args.out, "w"), indent=1)
print("wrote", args.out)
