"""Diagnostic (GPU): actual errors of the Normal / SquashedNormal sampler log-probs
and loss gradients against the reference's golden vectors, and of the sampler's
log-prob against an fp64 evaluation of the reference formula ON THE KERNEL'S OWN
ACTION (separates evaluation error from the conditioning of log(1 - s^2 + eps))."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rl8_amd import hip

G = os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden")
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
host = lambda t: t.detach().cpu().numpy()
EPS = float(np.finfo(np.float32).eps)


def logp64(mean, log_std, a, squashed):
    mean, log_std, a = (x.astype(np.float64) for x in (mean, log_std, a))
    std = np.exp(log_std.astype(np.float32).astype(np.float64))  # fp32 exp rounding is part of the model's scale
    std = np.exp(log_std)
    if squashed:
        c = np.clip(a, -1 + EPS, 1 - EPS)
        u = 0.5 * (np.log1p(c) - np.log1p(-c))
    else:
        u = a
    lp = -((u - mean) ** 2) / (2 * std ** 2) - np.log(std) - 0.5 * np.log(2 * np.pi)
    if squashed:
        lp = np.clip(lp, -100, 100).sum(-1, keepdims=True) - np.log(1 - a ** 2 + EPS).sum(-1, keepdims=True)
    else:
        lp = lp.sum(-1, keepdims=True)
    return lp


def stats(name, got, want):
    d = np.abs(got.astype(np.float64) - want.astype(np.float64))
    rel = d / np.maximum(np.abs(want), 1e-30)
    mix = d / (1.0 + np.abs(want))
    print(f"{name:58s} max_abs={d.max():.3e} max_rel={rel.max():.3e} max |d|/(1+|w|)={mix.max():.3e}")


g = dict(np.load(os.path.join(G, "samplers.npz")))
for kind in ("normal", "squashed"):
    for adim in (1, 3):
        p = f"{kind}{adim}"
        a, lp = hip.normal_sample_logp(dev(g[f"{p}_mean"]), dev(g[f"{p}_log_std"]), dev(g[f"{p}_eps"]), squashed=kind == "squashed")
        a, lp = host(a), host(lp)
        ulp = np.abs(a - g[f"{p}_actions"]) / np.spacing(np.abs(g[f"{p}_actions"]).astype(np.float32))
        print(f"{p}: action max ulp diff = {ulp.max():.2f}, fraction differing = {(ulp > 0).mean():.4f}")
        stats(f"{p} logp vs reference golden", lp, g[f"{p}_logp"])
        stats(f"{p} logp vs fp64 formula on OWN action", lp, logp64(g[f"{p}_mean"], g[f"{p}_log_std"], a, kind == "squashed"))
        stats(f"{p} REFERENCE logp vs fp64 formula on ITS action", g[f"{p}_logp"], logp64(g[f"{p}_mean"], g[f"{p}_log_std"], g[f"{p}_actions"], kind == "squashed"))
        if kind == "squashed":
            cond = np.abs(2 * a / (1 - a.astype(np.float64) ** 2 + EPS)) * np.spacing(np.abs(a))
            print(f"   conditioning: 1 ulp of the action moves log(1-s^2+eps) by up to {cond.max():.3e}")

g = dict(np.load(os.path.join(G, "ppo_losses.npz")))
worst = {}
for case in g["cases"]:
    case = str(case)
    if case.startswith("cat"):
        continue
    m = g[f"{case}_values"].shape[0]
    c, d, e, vc, vf = (float(x) for x in g[f"{case}_hparams"])
    hp = hip.ppo_hparams(grad_scale=1.0 / m, clip_param=c, dual_clip_param=d or None, entropy_coeff=e, vf_clip_param=vc, vf_coeff=vf)
    sums, gm, gl, gv = hip.ppo_loss_normal(dev(g[f"{case}_feat_mean"]), dev(g[f"{case}_feat_log_std"]), dev(g[f"{case}_values"]),
                                           dev(g[f"{case}_actions"]), dev(g[f"{case}_logp_old"]), dev(g[f"{case}_advantages"]),
                                           dev(g[f"{case}_returns"]), hp, squashed=case.startswith("squashed"))
    for nm, got, want in (("grad_mean", gm, g[f"{case}_grad_mean"]), ("grad_log_std", gl, g[f"{case}_grad_log_std"])):
        got = host(got).astype(np.float64)
        dd = np.abs(got - want)
        rel = dd / np.maximum(np.abs(want), 1e-30)
        scale = np.abs(want).max()
        i = np.unravel_index(np.argmax(np.where(dd > 1e-9, rel, 0)), rel.shape)
        print(f"{case:14s} {nm:12s} max_abs={dd.max():.3e} max_abs/max|g|={dd.max()/scale:.3e} worst_rel={rel[i]:.3e} at want={want[i]:.3e} (action {g[f'{case}_actions'][i]:.9f})")
