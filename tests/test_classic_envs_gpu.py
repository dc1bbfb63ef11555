"""N2: MountainCar and Pendulum kernels against the reference's golden steps
(examples/mountain_car/env.py:12-38, examples/pendulum/env.py:12-39) and the
oracle, through the C ABI, and end to end inside Algorithm."""

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import oracle  # noqa: E402
from rl8_amd import AlgorithmConfig, hip  # noqa: E402
from rl8_amd.distributions import SquashedNormal  # noqa: E402
from rl8_amd.envs import MountainCar, MountainCarConfig, Pendulum, PendulumConfig  # noqa: E402

DEV = "cuda:0"


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    return t.to(dtype) if dtype is not None else t


def host(t):
    return t.detach().cpu().numpy()


def test_mountain_car_steps_match_reference(golden):
    g = golden("classic_env_steps.npz")
    for tag in ("mc_default", "mc_custom"):
        kw = dict(zip(g[f"{tag}_cfg_keys"].tolist(), g[f"{tag}_cfg"].tolist()))
        cfg = MountainCarConfig(**kw).to_abi()
        chained = dev(g[f"{tag}_state0"])
        for t in range(g[f"{tag}_actions"].shape[0]):
            prev = g[f"{tag}_state0"] if t == 0 else g[f"{tag}_states"][t - 1]
            state = dev(prev)
            n = state.shape[1]
            obs, rew = torch.empty(n, 2, device=DEV), torch.empty(n, 1, device=DEV)
            hip.mountain_car_step(state, dev(g[f"{tag}_actions"][t]), cfg, obs, rew)
            np.testing.assert_allclose(host(state), g[f"{tag}_states"][t], rtol=0, atol=1.2e-7)
            np.testing.assert_allclose(host(obs), g[f"{tag}_obs"][t], rtol=0, atol=1.2e-7)
            np.testing.assert_allclose(host(rew)[:, 0], g[f"{tag}_rewards"][t], rtol=0, atol=1.2e-7)
            hip.mountain_car_step(chained, dev(g[f"{tag}_actions"][t]), cfg, obs, rew)
            np.testing.assert_allclose(host(chained), g[f"{tag}_states"][t], rtol=0, atol=1e-6)


def test_pendulum_steps_match_reference(golden):
    g = golden("classic_env_steps.npz")
    for tag in ("pd_default", "pd_custom"):
        kw = dict(zip(g[f"{tag}_cfg_keys"].tolist(), g[f"{tag}_cfg"].tolist()))
        cfg = PendulumConfig(**kw).to_abi()
        chained = dev(g[f"{tag}_state0"])
        for t in range(g[f"{tag}_actions"].shape[0]):
            prev = g[f"{tag}_state0"] if t == 0 else g[f"{tag}_states"][t - 1]
            state = dev(prev)
            n = state.shape[1]
            obs, rew = torch.empty(n, 3, device=DEV), torch.empty(n, 1, device=DEV)
            hip.pendulum_step(state, dev(g[f"{tag}_actions"][t]), cfg, obs, rew)
            np.testing.assert_allclose(host(state), g[f"{tag}_states"][t], rtol=0, atol=2e-6)
            np.testing.assert_allclose(host(obs), g[f"{tag}_obs"][t], rtol=0, atol=2e-6)
            np.testing.assert_allclose(host(rew)[:, 0], g[f"{tag}_rewards"][t], rtol=1e-6, atol=1e-6)
            hip.pendulum_step(chained, dev(g[f"{tag}_actions"][t]), cfg, obs, rew)
            np.testing.assert_allclose(host(chained), g[f"{tag}_states"][t], rtol=0, atol=2e-5)


def test_resets_match_oracle_noise():
    n = 5000
    state, obs = torch.empty(2, n, device=DEV), torch.empty(n, 2, device=DEV)
    hip.mountain_car_reset(state, seed=31, reset_count=2, env_offset=100, obs_out=obs)
    want = oracle.mountain_car_reset(n, 31, 2, 100)
    assert np.array_equal(host(state), want)
    assert np.array_equal(host(obs), want.T)
    state, obs = torch.empty(2, n, device=DEV), torch.empty(n, 3, device=DEV)
    hip.pendulum_reset(state, seed=31, reset_count=2, env_offset=100, obs_out=obs)
    want, want_obs = oracle.pendulum_reset(n, 31, 2, 100)
    assert np.array_equal(host(state), want)
    np.testing.assert_allclose(host(obs), want_obs, rtol=0, atol=1e-6)
    hip.pendulum_reset(state, seed=31, reset_count=2, env_offset=100, obs_out=None)  # obs is optional
    assert np.array_equal(host(state), want)


def test_fused_mountain_car_step_vs_oracle():
    rng = np.random.default_rng(4)
    n = 30_001
    logits = rng.standard_normal((n, 1, 3)).astype(np.float32)
    value = rng.standard_normal((n, 1)).astype(np.float32)
    state0 = np.stack([rng.uniform(-1.2, 0.6, n), rng.uniform(-0.07, 0.07, n)]).astype(np.float32)
    rdr0 = rng.standard_normal((n, 1)).astype(np.float32)
    want_a, want_lp = oracle.categorical_sample(logits, seed=21, step=5, row_offset=7)
    want_s, want_obs, want_r = oracle.mountain_car_step(state0, want_a, oracle.mountain_car_cfg())
    state = dev(state0)
    action_col = torch.empty(n, 1, dtype=torch.int64, device=DEV)
    cols = {k: torch.empty(n, 1, device=DEV) for k in ("logp", "value", "reward", "rdr1")}
    obs = torch.empty(n, 2, device=DEV)
    hip.rollout_step_mountain_car(
        logits=dev(logits), value=dev(value), noise=None, state=state, cfg=MountainCarConfig().to_abi(),
        action_col=action_col, logp_col=cols["logp"], value_col=cols["value"], reward_col=cols["reward"],
        obs_col_next=obs, rdr_t=dev(rdr0), rdr_t1=cols["rdr1"], gamma=float(np.float32(0.95)), seed=21, step=5,
        env_offset=7, deterministic=False)
    assert np.array_equal(host(action_col), want_a)
    np.testing.assert_allclose(host(cols["logp"]), want_lp, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(host(state), want_s, rtol=0, atol=1.2e-7)
    np.testing.assert_allclose(host(obs), want_obs, rtol=0, atol=1.2e-7)
    np.testing.assert_allclose(host(cols["reward"]), want_r, rtol=0, atol=1.2e-7)
    assert np.array_equal(host(cols["value"]), value)
    np.testing.assert_allclose(host(cols["rdr1"]), oracle.rdr_step(rdr0, want_r, 0.95), rtol=1e-6, atol=1e-6)
    # injected exponentials and the deterministic mode
    q = rng.exponential(size=(n, 1, 3)).astype(np.float32)
    want_a, _ = oracle.categorical_sample(logits, q)
    state = dev(state0)
    hip.rollout_step_mountain_car(
        logits=dev(logits), value=dev(value), noise=dev(q), state=state, cfg=MountainCarConfig().to_abi(),
        action_col=action_col, logp_col=cols["logp"], value_col=cols["value"], reward_col=cols["reward"],
        obs_col_next=obs, rdr_t=None, rdr_t1=None, gamma=0.95, seed=0, step=0, env_offset=0, deterministic=False)
    assert np.array_equal(host(action_col), want_a)
    hip.rollout_step_mountain_car(
        logits=dev(logits), value=dev(value), noise=None, state=state, cfg=MountainCarConfig().to_abi(),
        action_col=action_col, logp_col=cols["logp"], value_col=cols["value"], reward_col=cols["reward"],
        obs_col_next=obs, rdr_t=None, rdr_t1=None, gamma=0.95, seed=0, step=0, env_offset=0, deterministic=True)
    assert np.array_equal(host(action_col)[:, 0], logits[:, 0].argmax(-1))


@pytest.mark.parametrize("squashed", [False, True])
def test_fused_pendulum_step_vs_oracle(squashed):
    rng = np.random.default_rng(5)
    n = 30_001
    mean = rng.standard_normal((n, 1)).astype(np.float32)
    log_std = np.tanh(rng.standard_normal((n, 1))).astype(np.float32)
    value = rng.standard_normal((n, 1)).astype(np.float32)
    state0 = np.stack([rng.uniform(-7, 7, n), rng.uniform(-8, 8, n)]).astype(np.float32)
    rdr0 = rng.standard_normal((n, 1)).astype(np.float32)
    want_a, want_lp = oracle.normal_sample(mean, log_std, None, squashed=squashed, seed=3, step=9, row_offset=11)
    want_s, want_obs, want_r = oracle.pendulum_step(state0, want_a, oracle.pendulum_cfg())
    state = dev(state0)
    action_col = torch.empty(n, 1, device=DEV)
    cols = {k: torch.empty(n, 1, device=DEV) for k in ("logp", "value", "reward", "rdr1")}
    obs = torch.empty(n, 3, device=DEV)
    hip.rollout_step_pendulum(
        squashed=squashed, mean=dev(mean), log_std=dev(log_std), value=dev(value), noise=None, state=state,
        cfg=PendulumConfig().to_abi(), action_col=action_col, logp_col=cols["logp"], value_col=cols["value"],
        reward_col=cols["reward"], obs_col_next=obs, rdr_t=dev(rdr0), rdr_t1=cols["rdr1"],
        gamma=float(np.float32(0.95)), seed=3, step=9, env_offset=11, deterministic=False)
    np.testing.assert_allclose(host(action_col), want_a, rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(host(cols["logp"]), want_lp, rtol=1e-4, atol=5e-4)
    np.testing.assert_allclose(host(state), want_s, rtol=0, atol=5e-6)
    np.testing.assert_allclose(host(obs), want_obs, rtol=0, atol=5e-6)
    np.testing.assert_allclose(host(cols["reward"]), want_r, rtol=1e-6, atol=1e-5)
    np.testing.assert_allclose(host(cols["rdr1"]), oracle.rdr_step(rdr0, want_r, 0.95), rtol=1e-6, atol=1e-5)


@pytest.mark.parametrize("env_cls,kw", [(MountainCar, {}), (Pendulum, {}), (Pendulum, {"distribution_cls": SquashedNormal})])
def test_env_api_and_fused_rollout(env_cls, kw):
    env = env_cls(1000, 64, device=DEV)
    obs = env.reset()
    assert obs.shape == (1000, env.observation_spec.shape[-1]) and env.state.shape == (2, 1000)
    if env_cls is MountainCar:
        want_state = oracle.mountain_car_reset(1000, env.seed, 0, 0)
        actions = torch.randint(0, 3, (1000, 1), device=DEV)
        _, obs2, rew2 = oracle.mountain_car_step(want_state, host(actions), oracle.mountain_car_cfg())
    else:
        want_state, _ = oracle.pendulum_reset(1000, env.seed, 0, 0)
        actions = torch.randn(1000, 1, device=DEV) * 2
        _, obs2, rew2 = oracle.pendulum_step(want_state, host(actions), oracle.pendulum_cfg())
    assert np.array_equal(host(env.state), want_state)
    out = env.step(actions)
    np.testing.assert_allclose(host(out["obs"]), obs2, atol=5e-6)
    np.testing.assert_allclose(host(out["rewards"]), rew2, rtol=1e-6, atol=1e-5)
    with pytest.raises(ValueError, match="horizon"):
        env_cls(8, 513, device=DEV)
    env.reset(config={"max_speed": 0.05} if env_cls is MountainCar else {"g": 9.81, "l": 0.5})
    assert env.config["max_speed" if env_cls is MountainCar else "l"] in (0.05, 0.5)

    def run(force_generic):
        torch.manual_seed(11)
        algo = AlgorithmConfig(horizon=64, num_envs=2048, **kw).build(env_cls)
        assert algo._fusable()
        if force_generic:
            algo._fusable = lambda: False
        stats = algo.collect()
        buf = {k: v.clone() for k, v in algo.buffer.items()}
        step = algo.step()
        return stats, buf, step

    s_f, b_f, st_f = run(False)
    s_g, b_g, st_g = run(True)
    for k in b_f:
        if b_f[k].dtype == torch.int64 or env_cls is MountainCar:
            assert torch.equal(b_f[k], b_g[k]), k
        else:  # the standalone sampler and the fused one round the action identically; keep a hair of slack
            torch.testing.assert_close(b_f[k], b_g[k], rtol=1e-6, atol=1e-6, msg=k)
    assert s_f["returns/mean"] == pytest.approx(s_g["returns/mean"], rel=1e-6)
    assert st_f["losses/total"] == pytest.approx(st_g["losses/total"], rel=1e-5)
    assert AlgorithmConfig(horizon=1000, num_envs=8).build(env_cls).hparams.horizon == 512


def test_pendulum_learns():
    """A few updates on the real task: the mean return must improve."""
    torch.manual_seed(0)
    algo = AlgorithmConfig(horizon=128, num_envs=4096, horizons_per_env_reset=4).build(Pendulum)
    first = algo.collect()["returns/mean"]
    algo.step()
    for _ in range(40):
        last = algo.collect()["returns/mean"]
        algo.step()
    assert last > first + 0.1 * abs(first), (first, last)


def test_pendulum_learns_the_same_under_both_plane_schemes(monkeypatch):
    from .test_algorithm_gpu import assert_same_learning, learning_curve_ends

    ends = learning_curve_ends(Pendulum, 40, monkeypatch, horizon=128, num_envs=4096, horizons_per_env_reset=4)
    assert_same_learning(ends, 0.1)
