"""The kernels `north_star` names, at the metric's own size (BASELINE configs[1]:
N = 2^20 environments, H = 32, 2^25 samples), against the CPU oracle on the same
seeded inputs — index arithmetic, grid-stride tails and the two-level ticket fold
at full grid, not only at the few-thousand-row sizes of tests/test_hip_kernels.py.

Bars are the ones of the small-size tests: bit-exact for the GAE scan and its
normalisation (reference src/rl8/nn/functional.py:106-122), for action indices /
log-probabilities / env state of the fused rollout step (src/rl8/env.py:253-259,
distributions.py:113-132, algorithms/_feedforward.py:378-393) and for the gathers
(_utils.py:211-225); 1e-5 relative for the loss sums and 2e-5 for their gradients
(functional.py:316-363); 1e-6 for the fp64 statistics (_feedforward.py:411-436).
"""

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import oracle  # noqa: E402  (checker only)

from rl8_amd import hip  # noqa: E402

DEV = "cuda:0"
N, H = 1 << 20, 32


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a)).to(DEV)


def host(t):
    return t.detach().cpu().numpy()


@pytest.fixture(scope="module")
def rollout():
    """Rewards like the dummy env's (-|state|, a +-1 walk from U(-100, 100)), values ~ N(0, 1)."""
    rng = np.random.default_rng(2020)
    state = rng.uniform(-100, 100, (N, 1)).astype(np.float32)
    walk = np.cumsum(rng.integers(0, 2, (N, H + 1)).astype(np.float32) * 2 - 1, axis=1)
    rewards = -np.abs(state + walk).astype(np.float32).reshape(N, H + 1, 1)
    values = rng.standard_normal((N, H + 1, 1)).astype(np.float32)
    return rewards, values


@pytest.mark.parametrize("layout", [hip.LAYOUT_TIME_MAJOR, hip.LAYOUT_ENV_MAJOR])
def test_gae_full_size_bit_exact(rollout, layout):
    rewards, values = rollout
    scale = 57.25
    want = oracle.gae(rewards, values, gamma=0.95, gae_lambda=0.95, reward_scale=scale, normalize_advantages=True)
    if layout == hip.LAYOUT_TIME_MAJOR:
        r, v = dev(rewards.reshape(N, H + 1).T), dev(values.reshape(N, H + 1).T)
    else:
        r, v = dev(rewards.reshape(N, H + 1)), dev(values.reshape(N, H + 1))
    adv, ret = torch.full_like(r, 7.0), torch.full_like(r, 9.0)
    moments = hip.gae_scan(
        r, v, adv, ret, layout=layout, n=N, h=H, gamma=float(np.float32(0.95)),
        gamma_lambda=float(np.float32(0.95 * 0.95)), reward_denominator=float(np.float32(scale + 1e-8)),
        write_scaled_rewards=True)
    hip.advantage_normalise(adv, layout=layout, n=N, h=H, moments=moments)
    torch.cuda.synchronize()

    def back(t):
        a = host(t)
        return (a.T if layout == hip.LAYOUT_TIME_MAJOR else a).reshape(N, H + 1, 1)

    assert host(moments)[0] == N * H
    assert np.array_equal(back(r), want["scaled_rewards"])
    assert np.array_equal(back(ret), want["returns"])
    assert np.array_equal(back(adv), want["advantages"])


@pytest.mark.parametrize("time_major", [True, False])
def test_rollout_stats_full_size(rollout, time_major):
    rewards, _ = rollout
    rng = np.random.default_rng(3)
    rdr = (rng.standard_normal((N, H + 1, 1)) * 40).astype(np.float32)

    def put(a):
        t = dev(a)
        return t.transpose(0, 1).contiguous().transpose(0, 1) if time_major else t

    raw = host(hip.rollout_stats(put(rewards), put(rdr)))
    again = host(hip.rollout_stats(put(rewards), put(rdr)))
    assert np.array_equal(raw, again), "fp64 two-level fold must be reproducible at full grid"
    n, s1, s2, mn, mx, nh, r1, r2, rmn, rmx, d1, d2 = raw
    assert n == N and nh == N * H

    def std(cnt, a, b):
        return float(np.sqrt(max((b - a * a / cnt) / (cnt - 1), 0.0)))

    got = {
        "returns/min": mn, "returns/max": mx, "returns/mean": s1 / n, "returns/std": std(n, s1, s2),
        "rewards/min": rmn, "rewards/max": rmx, "rewards/mean": r1 / nh, "rewards/std": std(nh, r1, r2),
        "reward_scale": std(nh, d1, d2),
    }
    want = oracle.rollout_stats(rewards, rdr)
    for k in want:
        assert got[k] == pytest.approx(want[k], rel=1e-6), k
    # exact pieces: the extrema, and the sum of returns against numpy in fp64
    assert rmn == rewards[:, :H].min() and rmx == rewards[:, :H].max()
    assert r1 == pytest.approx(float(rewards[:, :H].astype(np.float64).sum()), rel=1e-12)


def test_fused_dummy_step_full_size_bit_exact():
    n = N
    rng = np.random.default_rng(11)
    logits = (rng.standard_normal((n, 1, 2)) * 0.5).astype(np.float32)
    value = rng.standard_normal((n, 1)).astype(np.float32)
    state0 = rng.uniform(-100, 100, (n, 1)).astype(np.float32)
    rdr0 = rng.standard_normal((n, 1)).astype(np.float32)
    want_a, want_lp = oracle.categorical_sample(logits, None, seed=77, step=5, row_offset=0)
    want_s, want_r = oracle.dummy_env_step(state0, want_a)
    want_rdr = oracle.rdr_step(rdr0, want_r, 0.95)
    state = dev(state0)
    cols = {k: torch.empty(n, 1, device=DEV) for k in ("logp", "value", "reward", "obs", "rdr1")}
    action_col = torch.empty(n, 1, dtype=torch.int64, device=DEV)
    hip.rollout_step_dummy(
        discrete=True, squashed=False, features=dev(logits), features2=None, value=dev(value), noise=None,
        state=state, action_col=action_col, logp_col=cols["logp"], value_col=cols["value"],
        reward_col=cols["reward"], obs_col_next=cols["obs"], rdr_t=dev(rdr0), rdr_t1=cols["rdr1"],
        gamma=float(np.float32(0.95)), seed=77, step=5, env_offset=0, deterministic=False)
    assert np.array_equal(host(action_col), want_a)
    assert 0.45 < want_a.mean() < 0.55  # the sampler really draws both actions
    assert np.array_equal(host(cols["logp"]), want_lp)
    assert np.array_equal(host(state), want_s)
    assert np.array_equal(host(cols["obs"]), want_s)
    assert np.array_equal(host(cols["reward"]), want_r)
    assert np.array_equal(host(cols["value"]), value)
    assert np.array_equal(host(cols["rdr1"]), want_rdr)


def test_ppo_loss_categorical_full_size():
    m, k = N * H, 2
    rng = np.random.default_rng(5)
    logits = (rng.standard_normal((m, 1, k), dtype=np.float32) * 1.5)
    values = rng.standard_normal((m, 1), dtype=np.float32) * 3
    returns = values + rng.standard_normal((m, 1), dtype=np.float32) * 2
    actions = rng.integers(0, k, (m, 1))
    logp_old = (np.float32(np.log(1.0 / k)) + rng.standard_normal((m, 1), dtype=np.float32) * np.float32(0.3))
    adv = rng.standard_normal((m, 1), dtype=np.float32)
    kw = dict(clip_param=0.2, dual_clip_param=None, entropy_coeff=0.0, vf_clip_param=5.0, vf_coeff=1.0)
    want, wg_logits, wg_values = oracle.ppo_loss_categorical(
        logits, values, actions, logp_old, adv, returns, oracle.ppo_hparams(grad_accumulation_steps=1, **kw))
    hp = hip.ppo_hparams(grad_scale=1.0 / m, **kw)
    d = [dev(a) for a in (logits, values, actions, logp_old, adv, returns)]
    sums, g_logits, g_value = hip.ppo_loss_categorical(*d, hp)
    s = host(sums)
    assert s[3] == m
    got = {"policy": s[1] / m, "vf": s[2] / m, "kl": s[4] / m}
    got["total"] = kw["vf_coeff"] * got["vf"] - got["policy"]
    for name, val in got.items():
        assert val == pytest.approx(want[name], rel=1e-5, abs=1e-7), name
    gl, gv = host(g_logits), host(g_value)
    # 2e-5 relative as at the small sizes; the absolute floor (1e-6 of the largest entry; the small-size tests'
    # 1e-9 at their grad_scale) covers entries that are cancellation residue: d logp / d logit = 1 - p with p
    # within 1e-4 of 1 carries fp32 softmax rounding at 1e-3 of itself in ANY evaluation order
    # (measured: 3550 of 67 M entries beyond 2e-5 relative, largest absolute difference 2.5e-14 = 1.5e-7 of max)
    np.testing.assert_allclose(gl, wg_logits, rtol=2e-5, atol=1e-6 * float(np.abs(wg_logits).max()))
    np.testing.assert_allclose(gv, wg_values, rtol=2e-5, atol=1e-6 * float(np.abs(wg_values).max()))
    assert np.array_equal(gl[:, 0, 0], -gl[:, 0, 1])
    # size independence: the first and the last 2^20 + 4 rows launched on their own give the
    # same gradients bit for bit (a row's gradient depends on the row and grad_scale alone; a multiple
    # of four rows, because the m % 4 tail of a call goes through the generic exact-order kernel, whose
    # exp / log differ from the vector kernel's hardware exp2 / log2 in the last bits)
    part = (1 << 20) + 4
    for sl in (slice(0, part), slice(m - part, m)):
        _, pl, pv = hip.ppo_loss_categorical(*[t[sl].contiguous() for t in d], hp)
        assert torch.equal(pl, g_logits[sl]) and torch.equal(pv, g_value[sl])


def test_pack_and_gather_full_size_bit_exact():
    """Batcher at 2^25 samples (src/rl8/_utils.py:211-225): every field of every sample
    through rl8_pack_samples + rl8_gather_packed under a full random permutation."""
    g = torch.Generator(device=DEV).manual_seed(1)
    shape = (H + 1, N, 1)
    obs, logp, adv, ret = (torch.randn(shape, device=DEV, generator=g).transpose(0, 1) for _ in range(4))
    act = torch.randint(0, 2, shape, device=DEV, generator=g).transpose(0, 1)
    leaves = [obs, act, logp, adv, ret]
    packed = hip.PackedSamples(H, leaves)
    perm = torch.randperm(N * H, device=DEV, generator=g)
    chunk = N * H // 8
    for i in (0, 7):
        idx = perm[i * chunk:(i + 1) * chunk].contiguous()
        outs = packed.gather(idx)
        env, t = idx // H, idx % H
        for leaf, out in zip(leaves, outs):
            assert torch.equal(out, leaf[env, t])
    # the generic strided gather on the same indices
    idx = perm[:chunk].contiguous()
    for leaf, out in zip(leaves, hip.gather_minibatch(idx, H, leaves)):
        assert torch.equal(out, leaf[idx // H, idx % H])
    # ... and every sample in order (index = NULL: the tiled transposition) = the leaves' own [N, H] prefix
    for leaf, out in zip(leaves, hip.gather_minibatch(None, H, leaves)):
        assert torch.equal(out, leaf[:, :H].reshape(N * H, 1))


# --------------------------------------------------------------------------- #
# Configs 3 and 4 at their own sizes: the fused CartPole step at 2^18 envs, the squashed-normal
# loss at 2^25 samples, the continuous squashed rollout step at 2^20 envs -- index arithmetic and grid-stride tails of
# those kernels at full grid, against the oracle on the same seeded inputs.
# --------------------------------------------------------------------------- #
def test_fused_cartpole_step_full_size():
    """BASELINE configs[2]: CartPole, 2^18 environments (examples/cartpole/env.py:12-64 of the reference + the
    sampler and bookkeeping of algorithms/_feedforward.py:362-393 in one launch): action indices and their
    log-probabilities bit-exact, physics 1e-6."""
    n = 1 << 18
    rng = np.random.default_rng(33)
    logits = rng.standard_normal((n, 1, 3)).astype(np.float32)
    value = rng.standard_normal((n, 1)).astype(np.float32)
    state0 = (rng.standard_normal((4, n)) * 0.5).astype(np.float32)
    rdr0 = rng.standard_normal((n, 1)).astype(np.float32)
    for integrator in (0, 1):
        want_a, want_lp = oracle.categorical_sample(logits, seed=21, step=97, row_offset=0)
        want_s, want_obs, want_r = oracle.cartpole_step(state0, want_a, oracle.cartpole_cfg(kinematics_integrator="semi-implicit" if integrator else "euler"))
        state = dev(state0)
        cfg = hip.CartPoleCfg(5.0, 9.8, 0.5, 0.1, 0.05, 1.1, 0.02, integrator)
        action_col = torch.empty(n, 1, dtype=torch.int64, device=DEV)
        cols = {k: torch.empty(n, 1, device=DEV) for k in ("logp", "value", "reward", "rdr1")}
        obs = torch.empty(n, 5, device=DEV)
        hip.rollout_step_cartpole(
            logits=dev(logits), value=dev(value), noise=None, state=state, cfg=cfg, action_col=action_col,
            logp_col=cols["logp"], value_col=cols["value"], reward_col=cols["reward"], obs_col_next=obs,
            rdr_t=dev(rdr0), rdr_t1=cols["rdr1"], gamma=float(np.float32(0.95)), seed=21, step=97, env_offset=0,
            deterministic=False)
        assert np.array_equal(host(action_col), want_a)
        assert set(np.unique(want_a)) == {0, 1, 2}
        assert np.array_equal(host(cols["logp"]), want_lp)
        assert np.array_equal(host(cols["value"]), value)
        np.testing.assert_allclose(host(state), want_s, rtol=0, atol=1e-6)
        np.testing.assert_allclose(host(obs), want_obs, rtol=0, atol=1e-6)
        np.testing.assert_allclose(host(cols["reward"]), want_r, rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(host(cols["rdr1"]), oracle.rdr_step(rdr0, want_r, 0.95), rtol=1e-6, atol=1e-6)


def test_fused_continuous_squashed_step_full_size():
    """BASELINE configs[3]'s rollout step: ContinuousDummyEnv + SquashedNormal at 2^20 environments
    (src/rl8/distributions.py:147-170, env.py:224-230) with the build's own Philox noise."""
    from .test_hip_kernels import assert_logp_close

    n = N
    rng = np.random.default_rng(41)
    mean = rng.standard_normal((n, 1)).astype(np.float32)
    log_std = np.tanh(rng.standard_normal((n, 1))).astype(np.float32)
    value = rng.standard_normal((n, 1)).astype(np.float32)
    state0 = rng.uniform(-100, 100, (n, 1)).astype(np.float32)
    rdr0 = rng.standard_normal((n, 1)).astype(np.float32)
    want_a, want_lp = oracle.normal_sample(mean, log_std, squashed=True, seed=9, step=31, row_offset=0)
    want_s, want_r = oracle.dummy_env_step(state0, want_a)
    state = dev(state0)
    cols = {k: torch.empty(n, 1, device=DEV) for k in ("action", "logp", "value", "reward", "obs", "rdr1")}
    hip.rollout_step_dummy(
        discrete=False, squashed=True, features=dev(mean), features2=dev(log_std), value=dev(value), noise=None,
        state=state, action_col=cols["action"], logp_col=cols["logp"], value_col=cols["value"],
        reward_col=cols["reward"], obs_col_next=cols["obs"], rdr_t=dev(rdr0), rdr_t1=cols["rdr1"],
        gamma=float(np.float32(0.95)), seed=9, step=31, env_offset=0, deterministic=False)
    got_a = host(cols["action"])
    np.testing.assert_allclose(got_a, want_a, rtol=1e-6, atol=1e-6)
    assert (np.abs(got_a) <= 1).all()  # (tanh rounds to 1.0 in fp32 beyond |u| ~ 9: the reference's logp clamps for that)
    assert_logp_close(host(cols["logp"]), want_lp, got_a, squashed=True)
    # the env step on the DEVICE's own action is exact fp32 arithmetic: state + a, -|state|
    assert np.array_equal(host(state), state0 + got_a)
    assert np.array_equal(host(cols["obs"]), host(state))
    assert np.array_equal(host(cols["reward"]), -np.abs(host(state)))
    assert np.array_equal(host(cols["value"]), value)
    np.testing.assert_allclose(host(state), want_s, rtol=1e-6, atol=1e-5)
    np.testing.assert_allclose(host(cols["rdr1"]), oracle.rdr_step(rdr0, host(cols["reward"]), 0.95), rtol=1e-6, atol=1e-5)


def test_ppo_loss_squashed_normal_full_size():
    """BASELINE configs[3]'s loss: SquashedNormal at 2^25 samples (src/rl8/nn/functional.py:316-363,
    distributions.py:147-170): sums 1e-5, gradients at the small-size bar, sub-launches bit for bit."""
    m = N * H
    rng = np.random.default_rng(6)
    mean = rng.standard_normal((m, 1), dtype=np.float32)
    log_std = np.tanh(rng.standard_normal((m, 1), dtype=np.float32))
    values = rng.standard_normal((m, 1), dtype=np.float32) * 3
    returns = values + rng.standard_normal((m, 1), dtype=np.float32) * 2
    eps = rng.standard_normal((m, 1), dtype=np.float32)
    actions, logp = oracle.normal_sample(mean, log_std, eps, squashed=True)
    logp_old = (logp + rng.standard_normal((m, 1), dtype=np.float32) * np.float32(0.3)).astype(np.float32)
    adv = rng.standard_normal((m, 1), dtype=np.float32)
    kw = dict(clip_param=0.2, dual_clip_param=None, entropy_coeff=0.0, vf_clip_param=5.0, vf_coeff=1.0)
    want, wg_mean, wg_ls, wg_values = oracle.ppo_loss_normal(
        mean, log_std, values, actions, logp_old, adv, returns, oracle.ppo_hparams(grad_accumulation_steps=1, **kw),
        squashed=True)
    hp = hip.ppo_hparams(grad_scale=1.0 / m, **kw)
    d = [dev(a) for a in (mean, log_std, values, actions, logp_old, adv, returns)]
    sums, g_mean, g_ls, g_value = hip.ppo_loss_normal(*d, hp, squashed=True)
    s = host(sums)
    assert s[3] == m
    got = {"policy": s[1] / m, "vf": s[2] / m, "kl": s[4] / m}
    got["total"] = kw["vf_coeff"] * got["vf"] - got["policy"]
    for name, val in got.items():
        assert val == pytest.approx(want[name], rel=1e-5, abs=1e-7), name
    # A sample whose probability ratio sits within rounding of a clip boundary (1 +- clip_param) has a gradient of 0 on
    # one side and of full size on the other, in ANY evaluation order: those few (5 of 2^25 here) are exempt from the
    # entrywise bar, every other entry is held to it, and the exempt ones must really sit on a boundary.
    ratio = np.exp(logp.astype(np.float64) - logp_old.astype(np.float64))
    on_boundary = (np.minimum(np.abs(ratio - 0.8), np.abs(ratio - 1.2)) < 1e-5).reshape(-1)
    assert on_boundary.sum() < 4096
    for got_g, want_g, name in ((g_mean, wg_mean, "mean"), (g_ls, wg_ls, "log_std"), (g_value, wg_values, "value")):
        got_h = host(got_g)
        bad = (np.abs(got_h - want_g) > 2e-5 * np.abs(want_g) + 1e-6 * float(np.abs(want_g).max())).reshape(-1)
        assert not (bad & ~on_boundary).any(), (name, int((bad & ~on_boundary).sum()))
        assert bad.sum() <= 64, (name, int(bad.sum()))
    part = (1 << 20) + 4
    for sl in (slice(0, part), slice(m - part, m)):
        _, pm, pl, pv = hip.ppo_loss_normal(*[t[sl].contiguous() for t in d], hp, squashed=True)
        assert torch.equal(pm, g_mean[sl]) and torch.equal(pl, g_ls[sl]) and torch.equal(pv, g_value[sl])
