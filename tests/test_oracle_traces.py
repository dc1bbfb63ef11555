"""Pins the oracle's CPU restatement of Algorithm.collect()/step()
(oracle/ppo_cpu.py) to end-to-end traces of the real reference: same initial
weights, reset state, per-timestep noise and permutations in -> buffer,
CollectStats, StepStats and final weights out."""

import numpy as np
import pytest
import torch

from oracle.ppo_cpu import OraclePPO, load_reference_weights

CASES = [
    ("trace_ff_discrete.npz", dict(env="discrete")),
    ("trace_ff_discrete_minibatch.npz",
     dict(env="discrete", sgd_minibatch_size=256, entropy_coeff=1e-2, dual_clip_param=5.0, horizons_per_env_reset=2)),
    ("trace_ff_continuous_squashed.npz", dict(env="continuous", distribution="squashed")),
    ("trace_ff_continuous_normal.npz", dict(env="continuous", entropy_coeff=1e-2)),
]


@pytest.mark.parametrize("name,kw", CASES)
def test_oracle_driver_reproduces_reference_trace(golden, name, kw):
    torch.set_num_threads(8)
    g = golden(name)
    algo = OraclePPO(num_envs=64, horizon=32, **kw)
    load_reference_weights(algo.model, g)
    discrete = kw["env"] == "discrete"
    for it in range(2):
        noise = g[f"it{it}_cat_q"] if discrete else g[f"it{it}_normal_eps"]
        reset_state = g.get(f"it{it}_reset_state")
        stats = algo.collect(noise=noise, reset_state=reset_state)
        if discrete:
            assert np.array_equal(algo.buf["actions"][:, :32], g[f"it{it}_collect_actions"][:, :32])
        for key in ("obs", "rewards", "values", "reversed_discounted_returns"):
            np.testing.assert_allclose(algo.buf[key], g[f"it{it}_collect_{key}"], rtol=1e-5, atol=2e-5, err_msg=key)
        # log(1 - tanh(u)^2 + eps) is ill-conditioned near saturation: an ulp in
        # tanh moves the squashed logp by ~1e-4 absolute.
        np.testing.assert_allclose(algo.buf["logp"], g[f"it{it}_collect_logp"], rtol=1e-5,
                                   atol=2e-5 if discrete else 5e-4, err_msg="logp")
        for k, w in zip(g["collect_stat_keys"], g[f"it{it}_collect_stats"]):
            assert stats[k] == pytest.approx(w, rel=2e-6), k
        assert algo.reward_scale == pytest.approx(float(g[f"it{it}_reward_scale"]), rel=2e-6)
        step = algo.step(perms=list(g[f"it{it}_perms"]))
        for k, w in zip(g["step_stat_keys"], g[f"it{it}_step_stats"]):
            assert step[k] == pytest.approx(w, rel=2e-4, abs=1e-7), (it, k, step[k], w)
        np.testing.assert_allclose(algo.buf["obs"][:, -1], g[f"it{it}_final_obs"], rtol=1e-5, atol=1e-5)
        for k, v in algo.model.state_dict().items():
            np.testing.assert_allclose(v.numpy(), g[f"it{it}_final_{k}"], rtol=2e-3, atol=1e-4, err_msg=k)


@pytest.mark.parametrize("name,kw", CASES)
def test_oracle_first_update_matches_reference_to_1e5(golden, name, kw):
    """One SGD iteration over one full-buffer minibatch from the reference's
    initial weights: the reference's own StepStats and the gradient it handed to
    ``optimizer.step()`` (tests/golden/first_update_*.npz) at north_star's 1e-5."""
    torch.set_num_threads(8)
    g = golden(name)
    first = golden(name.replace("trace_", "first_update_"))
    algo = OraclePPO(num_envs=64, horizon=32, **{**kw, "num_sgd_iters": 1, "sgd_minibatch_size": None})
    load_reference_weights(algo.model, g)
    discrete = kw["env"] == "discrete"
    algo.collect(noise=g["it0_cat_q"] if discrete else g["it0_normal_eps"], reset_state=g.get("it0_reset_state"))
    grads = {}
    real_step = algo.optimizer.step

    def recording_step(*a, **k):
        grads.update({n: p.grad.detach().clone().numpy() for n, p in algo.model.named_parameters()})
        return real_step(*a, **k)

    algo.optimizer.step = recording_step
    step = algo.step(perms=None)
    floors = {"losses/policy": 1e-6, "losses/total": 1e-6, "monitors/kl_div": 1e-7}
    for k, w in zip(first["step_stat_keys"], first["sgd1_step_stats"]):
        assert step[str(k)] == pytest.approx(w, rel=1e-5, abs=floors.get(str(k), 1e-9)), (k, step[str(k)], w)
    err_sq = ref_sq = 0.0
    for k, got in grads.items():
        w = first[f"sgd1_grad_{k}"].astype(np.float64)
        err_sq += float(((got - w) ** 2).sum())
        ref_sq += float((w ** 2).sum())
    assert (err_sq / ref_sq) ** 0.5 < 1e-5


def test_oracle_cartpole_first_update_matches_reference(golden):
    """Config 3 end to end on the CPU restatement (tests/golden/first_update_ff_cartpole.npz: the reference's
    CartPole through collect() / step(), eager Euler physics): rollout buffer, CollectStats, then one SGD iteration
    over the full buffer -- StepStats and the gradient at the first optimizer.step() at 1e-5."""
    torch.set_num_threads(8)
    g = golden("first_update_ff_cartpole.npz")
    algo = OraclePPO(env="cartpole", num_envs=64, horizon=32, num_sgd_iters=1)
    load_reference_weights(algo.model, g)
    stats = algo.collect(noise=g["it0_cat_q"], reset_state=g["it0_reset_state"])
    assert np.array_equal(algo.buf["actions"][:, :32], g["it0_collect_actions"][:, :32])
    for key in ("obs", "rewards", "reversed_discounted_returns"):
        # physics: 1e-6 absolute per step (one sin / cos per step differs by an ulp); the discounted sums relative
        np.testing.assert_allclose(algo.buf[key], g[f"it0_collect_{key}"], rtol=2e-6, atol=2e-6, err_msg=key)
    for key in ("values", "logp"):
        np.testing.assert_allclose(algo.buf[key], g[f"it0_collect_{key}"], rtol=1e-5, atol=2e-6, err_msg=key)
    for k, w in zip(g["collect_stat_keys"], g["it0_collect_stats"]):
        assert stats[str(k)] == pytest.approx(w, rel=2e-6, abs=1e-6), k
    assert algo.reward_scale == pytest.approx(float(g["it0_reward_scale"]), rel=2e-6)
    grads = {}
    real_step = algo.optimizer.step

    def recording_step(*a, **k):
        grads.update({n: p.grad.detach().clone().numpy() for n, p in algo.model.named_parameters()})
        return real_step(*a, **k)

    algo.optimizer.step = recording_step
    step = algo.step(perms=None)
    floors = {"losses/policy": 1e-6, "losses/total": 1e-6, "monitors/kl_div": 1e-7}
    for k, w in zip(g["step_stat_keys"], g["sgd1_step_stats"]):
        assert step[str(k)] == pytest.approx(w, rel=1e-5, abs=floors.get(str(k), 1e-9)), (k, step[str(k)], w)
    err_sq = ref_sq = 0.0
    for k, got in grads.items():
        w = g[f"sgd1_grad_{k}"].astype(np.float64)
        err_sq += float(((got - w) ** 2).sum())
        ref_sq += float((w ** 2).sum())
    assert (err_sq / ref_sq) ** 0.5 < 1e-5
