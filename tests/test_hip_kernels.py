"""Parity of the gfx950 kernels (called through the C ABI via rl8_amd.hip)
against the CPU oracle on identical seeded inputs, and against the reference's
golden vectors.

Bars: bit-exact for action indices, gathers, and add/mul/div-only arithmetic
(dummy env, GAE scan and normalisation, RDR recurrence); 1e-6 absolute for
CartPole physics (sin/cos differ by an ulp between libms; the reference's own
compiled and eager steps differ by as much); 1e-5 relative for losses / 2e-5 for
their gradients (north_star tolerance).
"""

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import oracle  # noqa: E402  (checker only)

from rl8_amd import hip  # noqa: E402

DEV = "cuda:0"


def dev(a, dtype=None):
    t = torch.as_tensor(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.to(DEV)


def host(t):
    return t.detach().cpu().numpy()


def test_abi_loads_on_device():
    version, arch = hip.abi_version()
    assert version == hip.ABI_VERSION == 106 and arch == "gfx950"
    assert torch.cuda.get_device_properties(0).gcnArchName.startswith("gfx950")


# --------------------------------------------------------------------------- #
# GAE
# --------------------------------------------------------------------------- #
def run_gae(rewards, values, *, layout, gamma, lam, scale, norm, write_back=True):
    """rewards/values numpy env-major [N, H+1, 1]; returns numpy env-major outputs."""
    n, h1 = rewards.shape[:2]
    h = h1 - 1
    if layout == hip.LAYOUT_TIME_MAJOR:
        r = dev(rewards.reshape(n, h1).T.copy())
        v = dev(values.reshape(n, h1).T.copy())
    else:
        r = dev(rewards.reshape(n, h1))
        v = dev(values.reshape(n, h1))
    adv = torch.full_like(r, 123.0)
    ret = torch.full_like(r, 456.0)
    denom = float(np.float32(scale + 1e-8))
    moments = hip.gae_scan(
        r, v, adv, ret, layout=layout, n=n, h=h, gamma=float(np.float32(gamma)),
        gamma_lambda=float(np.float32(gamma * lam)), reward_denominator=denom, write_scaled_rewards=write_back,
    )
    if norm:
        hip.advantage_normalise(adv, layout=layout, n=n, h=h, moments=moments)
    torch.cuda.synchronize()

    def back(t):
        a = host(t)
        if layout == hip.LAYOUT_TIME_MAJOR:
            a = a.T
        return a.reshape(n, h1, 1)

    return back(r), back(adv), back(ret), host(moments)


@pytest.mark.parametrize("layout", [hip.LAYOUT_ENV_MAJOR, hip.LAYOUT_TIME_MAJOR])
def test_gae_matches_reference_golden_bit_exact(golden, layout):
    g = golden("gae.npz")
    for case in g["cases"]:
        gamma, lam, scale, norm = g[f"{case}_params"]
        r, adv, ret, _ = run_gae(g[f"{case}_rewards"], g[f"{case}_values"], layout=layout, gamma=gamma,
                                 lam=lam, scale=scale, norm=bool(norm))
        assert np.array_equal(r, g[f"{case}_scaled_rewards"]), case
        assert np.array_equal(ret, g[f"{case}_returns"]), case
        assert np.array_equal(adv, g[f"{case}_advantages"]), case


@pytest.mark.parametrize("layout", [hip.LAYOUT_ENV_MAJOR, hip.LAYOUT_TIME_MAJOR])
def test_gae_known_answer(layout):
    # tests/test_nn/test_functional.py:14-49 of the reference
    ones = np.ones((10, 6, 1), np.float32)
    _, adv, ret, _ = run_gae(ones, ones, layout=layout, gamma=1.0, lam=1.0, scale=1.0, norm=False)
    undiscounted = np.flip(np.cumsum(ones, axis=1), axis=1)
    assert np.array_equal(adv, undiscounted - 1)
    assert np.array_equal(ret, undiscounted)


@pytest.mark.parametrize("layout", [hip.LAYOUT_ENV_MAJOR, hip.LAYOUT_TIME_MAJOR])
@pytest.mark.parametrize("n,h", [(1, 1), (5, 3), (1000, 32), (4096, 32), (777, 128), (300, 400), (8192, 33)])
def test_gae_vs_oracle_shapes(layout, n, h):
    rng = np.random.default_rng(n * 1000 + h)
    rewards = -np.abs(rng.uniform(-100, 100, (n, h + 1, 1))).astype(np.float32)
    values = rng.standard_normal((n, h + 1, 1)).astype(np.float32)
    want = oracle.gae(rewards, values, gamma=0.95, gae_lambda=0.9, reward_scale=41.5, normalize_advantages=n * h > 1)
    r, adv, ret, moments = run_gae(rewards, values, layout=layout, gamma=0.95, lam=0.9, scale=41.5, norm=n * h > 1)
    assert np.array_equal(r, want["scaled_rewards"])
    assert np.array_equal(ret, want["returns"])
    assert np.array_equal(adv, want["advantages"])
    assert moments[0] == n * h


def test_gae_without_write_back_leaves_rewards():
    rng = np.random.default_rng(5)
    rewards = rng.standard_normal((64, 9, 1)).astype(np.float32)
    values = rng.standard_normal((64, 9, 1)).astype(np.float32)
    want = oracle.gae(rewards, values, reward_scale=3.0, normalize_advantages=False)
    for layout in (hip.LAYOUT_ENV_MAJOR, hip.LAYOUT_TIME_MAJOR):
        r, adv, ret, _ = run_gae(rewards, values, layout=layout, gamma=0.95, lam=0.95, scale=3.0, norm=False,
                                 write_back=False)
        assert np.array_equal(r, rewards)
        assert np.array_equal(adv, want["advantages"])
        assert np.array_equal(ret, want["returns"])


def test_gae_argument_checks():
    t = torch.zeros(8, device=DEV)
    with pytest.raises(ValueError):
        hip.gae_scan(t, t, t, t, layout=1, n=0, h=1, gamma=1.0, gamma_lambda=1.0, reward_denominator=1.0,
                     write_scaled_rewards=False)
    with pytest.raises(ValueError):
        hip.gae_scan(t, t, t, t, layout=7, n=2, h=3, gamma=1.0, gamma_lambda=1.0, reward_denominator=1.0,
                     write_scaled_rewards=False)
    with pytest.raises(hip.HipExtensionError):
        hip.gae_scan(t.cpu(), t, t, t, layout=1, n=2, h=3, gamma=1.0, gamma_lambda=1.0, reward_denominator=1.0,
                     write_scaled_rewards=False)


# --------------------------------------------------------------------------- #
# PPO loss
# --------------------------------------------------------------------------- #
def _hp_from(arr, m, gas=1):
    clip, dual, ent, vfclip, vfc = (float(x) for x in arr)
    return dict(clip_param=clip, dual_clip_param=dual or None, entropy_coeff=ent, vf_clip_param=vfclip, vf_coeff=vfc), \
        1.0 / (m * gas)


def _losses_from_sums(sums, kw, gas=1):
    ent_s, pol_s, vf_s, cnt, kl_s = sums
    ent = ent_s / cnt if kw["entropy_coeff"] != 0 else 0.0
    pol, vf = pol_s / cnt, vf_s / cnt
    total = kw["vf_coeff"] * vf - pol - (kw["entropy_coeff"] * ent if kw["entropy_coeff"] != 0 else 0.0)
    return [ent / gas, pol / gas, vf / gas, total / gas, kl_s / cnt]


def test_ppo_loss_matches_reference_autograd(golden):
    g = golden("ppo_losses.npz")
    for case in g["cases"]:
        m = g[f"{case}_values"].shape[0]
        kw, gscale = _hp_from(g[f"{case}_hparams"], m)
        hp = hip.ppo_hparams(grad_scale=gscale, **kw)
        common = [dev(g[f"{case}_values"])]
        tail = [dev(g[f"{case}_logp_old"]), dev(g[f"{case}_advantages"]), dev(g[f"{case}_returns"])]
        if case.startswith("cat"):
            sums, g_logits, g_value = hip.ppo_loss_categorical(
                dev(g[f"{case}_feat_logits"]), common[0], dev(g[f"{case}_actions"]), *tail, hp)
            np.testing.assert_allclose(host(g_logits), g[f"{case}_grad_logits"], rtol=2e-5, atol=1e-8, err_msg=case)
        else:
            sums, g_mean, g_ls, g_value = hip.ppo_loss_normal(
                dev(g[f"{case}_feat_mean"]), dev(g[f"{case}_feat_log_std"]), common[0], dev(g[f"{case}_actions"]),
                *tail, hp, squashed=case.startswith("squashed"))
            # 2e-5 relative like the categorical gradients; the absolute floor (1e-6 of the
            # largest entry) covers entries that are themselves cancellation residue
            # (measured: <= 9e-7 of max|g| over all 28 cases, profiles/r02_normal_tolerances.txt)
            for got, name in ((g_mean, "grad_mean"), (g_ls, "grad_log_std")):
                want = g[f"{case}_{name}"]
                np.testing.assert_allclose(host(got), want, rtol=2e-5, atol=1e-6 * float(np.abs(want).max()),
                                           err_msg=f"{case} {name}")
        np.testing.assert_allclose(host(g_value), g[f"{case}_grad_values"], rtol=2e-5, atol=1e-9, err_msg=case)
        got = _losses_from_sums(host(sums), kw)
        for i, name in enumerate(oracle.LOSS_KEYS):
            assert got[i] == pytest.approx(g[f"{case}_losses"][i], rel=1e-5, abs=1e-7), (case, name)


@pytest.mark.parametrize("m", [1, 3, 4, 7, 1025, 65536 + 3])
@pytest.mark.parametrize("k", [2, 3, 6])
def test_ppo_loss_categorical_vs_oracle(m, k):
    rng = np.random.default_rng(m * 10 + k)
    logits = (rng.standard_normal((m, 1, k)) * 1.5).astype(np.float32)
    values = rng.standard_normal((m, 1)).astype(np.float32) * 3
    returns = values + rng.standard_normal((m, 1)).astype(np.float32) * 2
    actions = rng.integers(0, k, (m, 1))
    logp_old = (np.log(1.0 / k) + rng.standard_normal((m, 1)) * 0.3).astype(np.float32)
    adv = rng.standard_normal((m, 1)).astype(np.float32)
    kw = dict(clip_param=0.2, dual_clip_param=5.0, entropy_coeff=1e-2, vf_clip_param=5.0, vf_coeff=0.7)
    gas = 4
    want, wg_logits, wg_values = oracle.ppo_loss_categorical(
        logits, values, actions, logp_old, adv, returns, oracle.ppo_hparams(grad_accumulation_steps=gas, **kw))
    hp = hip.ppo_hparams(grad_scale=1.0 / (m * gas), **kw)
    sums, g_logits, g_value = hip.ppo_loss_categorical(
        dev(logits), dev(values), dev(actions), dev(logp_old), dev(adv), dev(returns), hp)
    got = _losses_from_sums(host(sums), kw, gas)
    for i, name in enumerate(oracle.LOSS_KEYS):
        assert got[i] == pytest.approx(want[name], rel=1e-5, abs=1e-7), name
    np.testing.assert_allclose(host(g_logits), wg_logits, rtol=2e-5, atol=1e-9)
    np.testing.assert_allclose(host(g_value), wg_values, rtol=2e-5, atol=1e-10)
    if k == 2:  # a two-way categorical's gradients are exact negatives (what the pair weight-gradient kernel needs)
        gl = host(g_logits).reshape(m, 2)
        assert np.array_equal(gl[:, 0], -gl[:, 1])
    # forward-only launch gives the same sums
    sums2, none1, none2 = hip.ppo_loss_categorical(
        dev(logits), dev(values), dev(actions), dev(logp_old), dev(adv), dev(returns), hp, with_grad=False)
    assert none1 is None and none2 is None
    assert np.array_equal(host(sums2), host(sums))


def test_ppo_loss_multi_action_dims_vs_oracle():
    rng = np.random.default_rng(9)
    m, a, k = 513, 3, 4
    logits = rng.standard_normal((m, a, k)).astype(np.float32)
    values = rng.standard_normal((m, 1)).astype(np.float32)
    returns = rng.standard_normal((m, 1)).astype(np.float32)
    actions = rng.integers(0, k, (m, a))
    logp_old = (a * np.log(1.0 / k) + rng.standard_normal((m, 1)) * 0.3).astype(np.float32)
    adv = rng.standard_normal((m, 1)).astype(np.float32)
    kw = dict(clip_param=0.2, dual_clip_param=None, entropy_coeff=1e-2, vf_clip_param=5.0, vf_coeff=1.0)
    want, wg_logits, wg_values = oracle.ppo_loss_categorical(logits, values, actions, logp_old, adv, returns,
                                                             oracle.ppo_hparams(**kw))
    sums, g_logits, g_value = hip.ppo_loss_categorical(
        dev(logits), dev(values), dev(actions), dev(logp_old), dev(adv), dev(returns),
        hip.ppo_hparams(grad_scale=1.0 / m, **kw))
    got = _losses_from_sums(host(sums), kw)
    for i, name in enumerate(oracle.LOSS_KEYS):
        assert got[i] == pytest.approx(want[name], rel=1e-5, abs=1e-7), name
    np.testing.assert_allclose(host(g_logits), wg_logits, rtol=2e-5, atol=1e-9)
    np.testing.assert_allclose(host(g_value), wg_values, rtol=2e-5, atol=1e-10)


def test_ppo_loss_squashed_entropy_is_rejected():
    m = 8
    z = torch.zeros(m, 1, device=DEV)
    hp = hip.ppo_hparams(clip_param=0.2, dual_clip_param=None, entropy_coeff=0.01, vf_clip_param=5.0, vf_coeff=1.0,
                         grad_scale=1.0)
    with pytest.raises(ValueError):
        hip.ppo_loss_normal(z, z, z, z, z, z, z, hp, squashed=True)


# --------------------------------------------------------------------------- #
# Environments + samplers
# --------------------------------------------------------------------------- #
def test_dummy_env_steps_match_reference(golden):
    g = golden("env_steps.npz")
    for kind, dtype in (("disc", torch.int64), ("cont", torch.float32)):
        state = dev(g[f"{kind}_state0"])
        reward = torch.empty_like(state)
        for t in range(g[f"{kind}_actions"].shape[0]):
            hip.dummy_env_step(state, dev(g[f"{kind}_actions"][t], dtype), reward)
            assert np.array_equal(host(state), g[f"{kind}_states"][t]), (kind, t)
            assert np.array_equal(host(reward), g[f"{kind}_rewards"][t]), (kind, t)


def test_cartpole_steps_match_reference(golden):
    g = golden("env_steps.npz")
    for integ in ("euler", "semi-implicit"):
        cfg = hip.CartPoleCfg(5.0, 9.8, 0.5, 0.1, 0.05, 1.1, 0.02, 0 if integ == "euler" else 1)
        for t in range(g[f"cp_{integ}_actions"].shape[0]):
            prev = g[f"cp_{integ}_state0"] if t == 0 else g[f"cp_{integ}_states"][t - 1]
            state = dev(prev)
            n = state.shape[1]
            obs = torch.empty(n, 5, device=DEV)
            rew = torch.empty(n, 1, device=DEV)
            hip.cartpole_step(state, dev(g[f"cp_{integ}_actions"][t]), cfg, obs, rew)
            np.testing.assert_allclose(host(state), g[f"cp_{integ}_states"][t], rtol=0, atol=1e-6)
            np.testing.assert_allclose(host(obs), g[f"cp_{integ}_obs"][t], rtol=0, atol=1e-6)
            np.testing.assert_allclose(host(rew), g[f"cp_{integ}_rewards"][t], rtol=1e-6, atol=1e-6)


def test_env_resets_match_oracle_noise():
    n = 5000
    state = torch.empty(n, 1, device=DEV)
    hip.dummy_env_reset(state, 100.0, seed=1234, reset_count=3, env_offset=17)
    want = oracle.dummy_env_reset(n, 100.0, 1234, 3, 17)
    assert np.array_equal(host(state), want)
    assert np.abs(want).max() <= 100.0 and abs(float(want.mean())) < 5.0
    cp = torch.empty(4, n, device=DEV)
    obs = torch.empty(n, 5, device=DEV)
    hip.cartpole_reset(cp, 0.01, seed=99, reset_count=0, env_offset=0, obs_out=obs)
    want = oracle.cartpole_reset(n, 0.01, 99, 0, 0)
    assert np.array_equal(host(cp), want)
    assert 0.008 < float(want.std()) < 0.012
    np.testing.assert_allclose(host(obs)[:, 2], np.cos(want[2]), atol=1e-6)


@pytest.mark.parametrize("ncls", [2, 3, 5])
def test_categorical_sampler_matches_reference_actions(golden, ncls):
    g = golden("samplers.npz")
    actions, logp = hip.categorical_sample_logp(dev(g[f"cat{ncls}_logits"]), dev(g[f"cat{ncls}_q"]))
    assert np.array_equal(host(actions), g[f"cat{ncls}_actions"])
    np.testing.assert_allclose(host(logp), g[f"cat{ncls}_logp"], rtol=1e-5, atol=1e-6)
    mode, _ = hip.categorical_sample_logp(dev(g[f"cat{ncls}_logits"]), None, deterministic=True)
    assert np.array_equal(host(mode), g[f"cat{ncls}_mode"])


def test_categorical_sampler_philox_bit_exact_vs_oracle():
    rng = np.random.default_rng(0)
    m = 200_000
    logits = (rng.standard_normal((m, 1, 2)) * 1e-3).astype(np.float32)
    want_a, want_lp = oracle.categorical_sample(logits, seed=42, step=7, row_offset=1000)
    got_a, got_lp = hip.categorical_sample_logp(dev(logits), None, seed=42, step=7, row_offset=1000)
    assert np.array_equal(host(got_a), want_a)
    assert np.array_equal(host(got_lp), want_lp)
    assert 0.49 < want_a.mean() < 0.51
    logits3 = rng.standard_normal((4099, 2, 3)).astype(np.float32)
    want_a, want_lp = oracle.categorical_sample(logits3, seed=5, step=0)
    got_a, got_lp = hip.categorical_sample_logp(dev(logits3), None, seed=5, step=0)
    assert np.array_equal(host(got_a), want_a)
    np.testing.assert_allclose(host(got_lp), want_lp, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("k", [2, 3, 5])
def test_categorical_draw_ties_near_ties_and_large_logits_agree_with_the_oracle(k):
    """The sampler against the oracle bit for bit where a shortcut would be tempted to differ: logits of three scales
    (1e-3, 1, 30), exact ties (the first index wins), scores a few ulps apart, injected noise that is zero.  (Round 4
    tried deciding the index from fp32 log-scores behind a margin, with the reference's operations as the fallback:
    it passed this test and moved nothing -- the kernel issues ~420 VALU instructions per env either way,
    profiles/r04_experiments.md -- so the sampler stays the reference's arithmetic, operation by operation.)"""
    rng = np.random.default_rng(100 + k)
    m = 300_000
    logits = (rng.standard_normal((m, 1, k)) * rng.choice([1e-3, 1.0, 30.0], (m, 1, 1))).astype(np.float32)
    want_a, want_lp = oracle.categorical_sample(logits, seed=9, step=3, row_offset=17)
    got_a, got_lp = hip.categorical_sample_logp(dev(logits), None, seed=9, step=3, row_offset=17)
    assert np.array_equal(host(got_a), want_a) and np.array_equal(host(got_lp), want_lp)
    assert len(np.unique(want_a)) == k
    n = 60_000
    q = rng.exponential(1.0, (n, 1, k)).astype(np.float32)
    x = rng.standard_normal((n, 1, k)).astype(np.float32)
    tie = rng.integers(0, 3, n)
    # class 1 made to score exactly / almost what class 0 does: x1 - ln q1 = x0 - ln q0 (+ a few ulps)
    x[:, 0, 1] = (x[:, 0, 0] - np.log(q[:, 0, 0].astype(np.float64)) + np.log(q[:, 0, 1].astype(np.float64))
                  + (tie == 1) * rng.standard_normal(n) * 3e-7 + (tie == 2) * rng.standard_normal(n) * 1e-5).astype(np.float32)
    q[:100, 0, 1] = q[:100, 0, 0]
    x[:100, 0, 1] = x[:100, 0, 0]          # exact ties: the first index wins
    q[100:110, 0, 0] = 0.0                 # (the reference divides by zero: inf wins)
    want_a, want_lp = oracle.categorical_sample(x, q)
    got_a, got_lp = hip.categorical_sample_logp(dev(x), dev(q), seed=0, step=0)
    assert np.array_equal(host(got_a), want_a) and np.array_equal(host(got_lp), want_lp)
    assert (want_a[:100] != 1).all()       # class 1 never beats class 0 on an exact tie
    assert 0.1 < (want_a[110:, 0] == 1).mean() < 0.6


def assert_logp_close(got, want, actions, *, squashed):
    """Sampler log-probs at 2e-5 (north_star's 1e-5 bar, one ulp of slack on either
    side). ``Normal``: plain ``|d| <= 2e-5 * (1 + |want|)``.

    ``SquashedNormal`` adds the one term that cannot be held to that in fp32 BY
    EITHER SIDE: ``log(1 - s^2 + eps)`` (reference ``distributions.py:163-168``) is
    evaluated from the fp32 action ``s = tanh(u)``; near saturation one ulp of ``s``
    moves it by ``2|s| ulp(s) / (1 - s^2 + eps)`` -- up to 0.5 at ``|s| = 1 - 2^-24``.
    ``tanh`` on the device and in torch's CPU vector library differ by an ulp on
    ~10 % of draws, and the reference's own fp32 value is up to 1.3e-4 away from an
    fp64 evaluation of its formula on its own action
    (profiles/r02_normal_tolerances.txt). So each sample's band is widened by the
    movement of that term under 2 ulps of its action -- nothing else is loosened."""
    got, want = got.astype(np.float64), want.astype(np.float64)
    band = 2e-5 * (1.0 + np.abs(want))
    if squashed:
        s = actions.astype(np.float64)
        eps = float(np.finfo(np.float32).eps)
        moved = (2.0 * np.abs(s) / (1.0 - s * s + eps)) * np.spacing(np.abs(actions).astype(np.float32))
        band = band + 2.0 * moved.reshape(moved.shape[0], -1).sum(-1, keepdims=True).reshape(want.shape)
    bad = np.abs(got - want) > band
    assert not bad.any(), (int(bad.sum()), float(np.abs(got - want)[bad].max()))


@pytest.mark.parametrize("kind", ["normal", "squashed"])
@pytest.mark.parametrize("adim", [1, 3])
def test_normal_samplers_match_reference(golden, kind, adim):
    g = golden("samplers.npz")
    p = f"{kind}{adim}"
    actions, logp = hip.normal_sample_logp(dev(g[f"{p}_mean"]), dev(g[f"{p}_log_std"]), dev(g[f"{p}_eps"]),
                                           squashed=kind == "squashed")
    np.testing.assert_allclose(host(actions), g[f"{p}_actions"], rtol=1e-6, atol=1e-6)
    assert_logp_close(host(logp), g[f"{p}_logp"], host(actions), squashed=kind == "squashed")
    # Philox noise agrees with the oracle's
    a2, lp2 = hip.normal_sample_logp(dev(g[f"{p}_mean"]), dev(g[f"{p}_log_std"]), None, squashed=kind == "squashed",
                                     seed=3, step=11, row_offset=5)
    wa, wlp = oracle.normal_sample(g[f"{p}_mean"], g[f"{p}_log_std"], squashed=kind == "squashed", seed=3, step=11,
                                   row_offset=5)
    np.testing.assert_allclose(host(a2), wa, rtol=1e-6, atol=1e-6)
    assert_logp_close(host(lp2), wlp, host(a2), squashed=kind == "squashed")


# --------------------------------------------------------------------------- #
# Fused per-timestep kernels == composition of the standalone pieces
# --------------------------------------------------------------------------- #
@pytest.mark.parametrize("n", [1, 63, 4096, 100_001])
def test_fused_dummy_discrete_step_vs_oracle(n):
    rng = np.random.default_rng(n)
    logits = (rng.standard_normal((n, 1, 2)) * 0.5).astype(np.float32)
    value = rng.standard_normal((n, 1)).astype(np.float32)
    q = rng.exponential(1.0, (n, 1, 2)).astype(np.float32)
    state0 = rng.uniform(-100, 100, (n, 1)).astype(np.float32)
    rdr0 = rng.standard_normal((n, 1)).astype(np.float32)
    for noise in (q, None):
        want_a, want_lp = oracle.categorical_sample(logits, noise, seed=8, step=2, row_offset=3)
        want_s, want_r = oracle.dummy_env_step(state0, want_a)
        want_rdr = oracle.rdr_step(rdr0, want_r, 0.95)
        state = dev(state0)
        cols = {k: torch.empty(n, 1, device=DEV) for k in ("logp", "value", "reward", "obs", "rdr1")}
        action_col = torch.empty(n, 1, dtype=torch.int64, device=DEV)
        hip.rollout_step_dummy(
            discrete=True, squashed=False, features=dev(logits), features2=None, value=dev(value),
            noise=dev(noise) if noise is not None else None, state=state, action_col=action_col,
            logp_col=cols["logp"], value_col=cols["value"], reward_col=cols["reward"], obs_col_next=cols["obs"],
            rdr_t=dev(rdr0), rdr_t1=cols["rdr1"], gamma=float(np.float32(0.95)), seed=8, step=2, env_offset=3,
            deterministic=False)
        assert np.array_equal(host(action_col), want_a)
        assert np.array_equal(host(cols["logp"]), want_lp)
        assert np.array_equal(host(state), want_s)
        assert np.array_equal(host(cols["obs"]), want_s)
        assert np.array_equal(host(cols["reward"]), want_r)
        assert np.array_equal(host(cols["value"]), value)
        assert np.array_equal(host(cols["rdr1"]), want_rdr)


@pytest.mark.parametrize("squashed", [False, True])
def test_fused_dummy_continuous_step_vs_oracle(squashed):
    rng = np.random.default_rng(1)
    n = 10_000
    mean = rng.standard_normal((n, 1)).astype(np.float32)
    log_std = np.tanh(rng.standard_normal((n, 1))).astype(np.float32)
    value = rng.standard_normal((n, 1)).astype(np.float32)
    eps = rng.standard_normal((n, 1)).astype(np.float32)
    state0 = rng.uniform(-100, 100, (n, 1)).astype(np.float32)
    want_a, want_lp = oracle.normal_sample(mean, log_std, eps, squashed=squashed)
    want_s, want_r = oracle.dummy_env_step(state0, want_a)
    state = dev(state0)
    cols = {k: torch.empty(n, 1, device=DEV) for k in ("action", "logp", "value", "reward", "obs")}
    hip.rollout_step_dummy(
        discrete=False, squashed=squashed, features=dev(mean), features2=dev(log_std), value=dev(value),
        noise=dev(eps), state=state, action_col=cols["action"], logp_col=cols["logp"], value_col=cols["value"],
        reward_col=cols["reward"], obs_col_next=cols["obs"], rdr_t=None, rdr_t1=None, gamma=0.95, seed=0, step=0,
        env_offset=0, deterministic=False)
    np.testing.assert_allclose(host(cols["action"]), want_a, rtol=1e-6, atol=1e-6)
    assert_logp_close(host(cols["logp"]), want_lp, host(cols["action"]), squashed=squashed)
    np.testing.assert_allclose(host(state), want_s, rtol=1e-6, atol=1e-5)
    np.testing.assert_allclose(host(cols["reward"]), want_r, rtol=1e-6, atol=1e-5)


def test_fused_cartpole_step_vs_oracle():
    rng = np.random.default_rng(2)
    n = 30_000
    logits = rng.standard_normal((n, 1, 3)).astype(np.float32)
    value = rng.standard_normal((n, 1)).astype(np.float32)
    state0 = (rng.standard_normal((4, n)) * 0.5).astype(np.float32)
    rdr0 = rng.standard_normal((n, 1)).astype(np.float32)
    want_a, want_lp = oracle.categorical_sample(logits, seed=21, step=5)
    cfg_o = oracle.cartpole_cfg()
    want_s, want_obs, want_r = oracle.cartpole_step(state0, want_a, cfg_o)
    state = dev(state0)
    cfg = hip.CartPoleCfg(5.0, 9.8, 0.5, 0.1, 0.05, 1.1, 0.02, 0)
    action_col = torch.empty(n, 1, dtype=torch.int64, device=DEV)
    cols = {k: torch.empty(n, 1, device=DEV) for k in ("logp", "value", "reward", "rdr1")}
    obs = torch.empty(n, 5, device=DEV)
    hip.rollout_step_cartpole(
        logits=dev(logits), value=dev(value), noise=None, state=state, cfg=cfg, action_col=action_col,
        logp_col=cols["logp"], value_col=cols["value"], reward_col=cols["reward"], obs_col_next=obs,
        rdr_t=dev(rdr0), rdr_t1=cols["rdr1"], gamma=float(np.float32(0.95)), seed=21, step=5, env_offset=0,
        deterministic=False)
    assert np.array_equal(host(action_col), want_a)
    np.testing.assert_allclose(host(state), want_s, rtol=0, atol=1e-6)
    np.testing.assert_allclose(host(obs), want_obs, rtol=0, atol=1e-6)
    np.testing.assert_allclose(host(cols["reward"]), want_r, rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(host(cols["rdr1"]), oracle.rdr_step(rdr0, want_r, 0.95), rtol=1e-6, atol=1e-6)


def test_rollout_scatter_generic_env():
    rng = np.random.default_rng(3)
    n, od = 1000, 7
    action = rng.integers(0, 5, (n, 2))
    logp, value, reward = (rng.standard_normal((n, 1)).astype(np.float32) for _ in range(3))
    obs = rng.standard_normal((n, od)).astype(np.float32)
    rdr0 = rng.standard_normal((n, 1)).astype(np.float32)
    cols = {k: torch.zeros(n, 1, device=DEV) for k in ("logp", "value", "reward", "rdr1")}
    action_col = torch.zeros(n, 2, dtype=torch.int64, device=DEV)
    obs_col = torch.zeros(n, od, device=DEV)
    hip.rollout_scatter(dev(action), dev(logp), dev(value), dev(reward), dev(obs), action_col, cols["logp"],
                        cols["value"], cols["reward"], obs_col, dev(rdr0), cols["rdr1"], float(np.float32(0.95)))
    assert np.array_equal(host(action_col), action)
    assert np.array_equal(host(cols["logp"]), logp)
    assert np.array_equal(host(cols["value"]), value)
    assert np.array_equal(host(cols["reward"]), reward)
    assert np.array_equal(host(obs_col), obs)
    assert np.array_equal(host(cols["rdr1"]), oracle.rdr_step(rdr0, reward, 0.95))


# --------------------------------------------------------------------------- #
# Stats + gather
# --------------------------------------------------------------------------- #
def _stats_dict(raw):
    n, s1, s2, mn, mx, nh, r1, r2, rmn, rmx, d1, d2 = raw

    def std(cnt, a, b):
        return float(np.sqrt(max((b - a * a / cnt) / (cnt - 1), 0.0)))

    return {
        "returns/min": mn, "returns/max": mx, "returns/mean": s1 / n, "returns/std": std(n, s1, s2),
        "rewards/min": rmn, "rewards/max": rmx, "rewards/mean": r1 / nh, "rewards/std": std(nh, r1, r2),
        "reward_scale": std(nh, d1, d2),
    }


@pytest.mark.parametrize("time_major", [False, True])
def test_rollout_stats_vs_oracle_and_trace(golden, time_major):
    g = golden("trace_ff_discrete.npz")
    rewards, rdr = g["it0_collect_rewards"], g["it0_collect_reversed_discounted_returns"]

    def put(a):
        t = dev(a)
        if time_major:
            t = t.transpose(0, 1).contiguous().transpose(0, 1)
        return t

    got = _stats_dict(host(hip.rollout_stats(put(rewards), put(rdr))))
    want = oracle.rollout_stats(rewards, rdr)
    for k in want:
        assert got[k] == pytest.approx(want[k], rel=1e-6), k
    ref = dict(zip(g["collect_stat_keys"], g["it0_collect_stats"]))
    for k in ("returns/min", "returns/max", "returns/mean", "returns/std", "rewards/min", "rewards/max",
              "rewards/mean", "rewards/std"):
        assert got[k] == pytest.approx(ref[k], rel=2e-6), k
    assert got["reward_scale"] == pytest.approx(float(g["it0_reward_scale"]), rel=2e-6)
    rng = np.random.default_rng(4)
    big = rng.standard_normal((5000, 17, 1)).astype(np.float32)
    got = _stats_dict(host(hip.rollout_stats(put(big), None)))
    want = oracle.rollout_stats(big, None)
    for k in want:
        if k != "reward_scale":
            assert got[k] == pytest.approx(want[k], rel=1e-6, abs=1e-9), k


@pytest.mark.parametrize("time_major", [False, True])
def test_gather_minibatch_bit_exact(time_major):
    rng = np.random.default_rng(6)
    n, h = 300, 12
    obs = rng.standard_normal((n, h + 1, 5)).astype(np.float32)
    act = rng.integers(0, 9, (n, h + 1, 2))
    adv = rng.standard_normal((n, h + 1, 1)).astype(np.float32)

    def put(a):
        t = dev(a)
        if time_major:
            t = t.transpose(0, 1).contiguous().transpose(0, 1)
        return t

    index = oracle.permutation(n * h, 11, 0)[:1000]
    outs = hip.gather_minibatch(dev(index), h, [put(obs), put(act), put(adv)])
    env, t = index // h, index % h
    assert np.array_equal(host(outs[0]), obs[env, t])
    assert np.array_equal(host(outs[1]), act[env, t])
    assert np.array_equal(host(outs[2]), adv[env, t])
    # same thing through the oracle's row gather on the flattened [N*H] view
    flat = np.ascontiguousarray(obs[:, :h]).reshape(n * h, 5)
    assert np.array_equal(host(outs[0]), oracle.gather_rows(index, flat))
    # and through the packed rows (pack once, gather many): 5 + 2*2 + 1 = 10 -> 12 words per row
    packed = hip.PackedSamples(h, [put(obs), put(act), put(adv)])
    assert packed.row_words == 12 and packed.samples == n * h
    for seed in (0, 1):
        index = oracle.permutation(n * h, 11, seed)[: 1000 + seed]
        outs = packed.gather(dev(index))
        env, t = index // h, index % h
        assert np.array_equal(host(outs[0]), obs[env, t])
        assert np.array_equal(host(outs[1]), act[env, t]) and outs[1].dtype == torch.int64
        assert np.array_equal(host(outs[2]), adv[env, t])
    whole = packed.gather(dev(np.arange(n * h)))
    assert np.array_equal(host(whole[0]), flat)


@pytest.mark.parametrize("n,h,widths", [(1, 1, (1,)), (63, 5, (1, 2)), (64, 32, (1, 1, 1, 1)), (65, 33, (5, 1, 1)),
                                        (1000, 100, (7, 3, 1)), (130, 256, (28,)), (4100, 7, (3, 2, 2, 1, 1, 1, 1, 1))])
@pytest.mark.parametrize("time_major", [False, True])
def test_identity_gather_of_wide_leaves(n, h, widths, time_major):
    """``rl8_gather_minibatch(index = NULL)`` beyond one tile row (128 bytes per sample): narrow leaves go through the
    tiled transposition in groups that fit, a leaf wider than a tile row through the general kernels with the implicit
    index -- the same dense rows as the indexed gather of ``arange`` either way."""
    g = torch.Generator(device=DEV).manual_seed(n * 31 + h)
    widths = (40,) + tuple(widths) + (9, 64, 20)
    leaves = []
    for i, d in enumerate(widths):
        shape = (h + 1, n, d) if time_major else (n, h + 1, d)
        t = (torch.randint(-5, 9, shape, device=DEV, generator=g) if i == 2 else torch.randn(shape, device=DEV, generator=g))
        leaves.append(t.transpose(0, 1) if time_major else t)
    leaves = leaves[:hip.MAX_GATHER_FIELDS]
    everything = torch.arange(n * h, device=DEV)
    for dense, indexed, leaf in zip(hip.gather_minibatch(None, h, leaves), hip.gather_minibatch(everything, h, leaves), leaves):
        assert dense.dtype == leaf.dtype and torch.equal(dense, indexed)
        assert torch.equal(dense, leaf[:, :h].reshape(n * h, *leaf.shape[2:]))


@pytest.mark.parametrize("n,h,widths", [(1, 1, (1,)), (63, 5, (1, 2)), (64, 32, (1, 1, 1, 1)), (65, 33, (5, 1, 1)),
                                        (1000, 100, (7, 3, 1)), (130, 256, (28,)), (4100, 7, (3, 2, 2, 1, 1, 1, 1, 1))])
@pytest.mark.parametrize("time_major", [False, True])
def test_pack_samples_tiles_ragged_shapes(n, h, widths, time_major):
    """rl8_pack_samples as a tiled transposition (64 envs x up to 32 steps through LDS): the packed buffer itself --
    every sample's fields side by side in the reference's sample order env * H + t (src/rl8/_utils.py:211-225), zero
    padding to whole 16-byte vectors -- for ragged env / step counts, rows of 16 to 128 bytes, 4- and 8-byte elements,
    both buffer layouts."""
    g = torch.Generator(device=DEV).manual_seed(n + h)
    leaves = []
    for i, d in enumerate(widths):
        shape = (h + 1, n, d) if time_major else (n, h + 1, d)
        t = (torch.randint(-5, 9, shape, device=DEV, generator=g) if i == 1 else torch.randn(shape, device=DEV, generator=g))
        leaves.append(t.transpose(0, 1) if time_major else t)
    packed = hip.PackedSamples(h, leaves)
    words = sum(d * (2 if i == 1 else 1) for i, d in enumerate(widths))
    assert packed.row_words == (words + 3) // 4 * 4
    got = packed.packed.view(n * h, packed.row_words)
    want = torch.cat([leaf[:, :h].reshape(n * h, -1).contiguous().view(torch.int32) for leaf in leaves], 1)
    assert torch.equal(got[:, :words], want)
    assert not bool(got[:, words:].any())
    index = torch.randperm(n * h, device=DEV, generator=g)[: max(1, n * h // 3)]
    for leaf, out in zip(leaves, packed.gather(index.contiguous())):
        assert torch.equal(out, leaf[index // h, index % h])
    # rl8_gather_minibatch(index = NULL): the same tiles leave as the dense fields of a gather of every sample in order
    everything = torch.arange(n * h, device=DEV)
    for dense, indexed, leaf in zip(hip.gather_minibatch(None, h, leaves), hip.gather_minibatch(everything, h, leaves), leaves):
        assert dense.dtype == leaf.dtype and torch.equal(dense, indexed)
        assert torch.equal(dense, leaf[:, :h].reshape(n * h, *leaf.shape[2:]))


# --------------------------------------------------------------------------- #
# Single-launch reductions: the last-arriving block must see every other
# block's partial row (cross-CU / cross-XCD hand-off) on every launch.
# --------------------------------------------------------------------------- #
def test_single_launch_reductions_are_complete_and_reproducible():
    g = torch.Generator(device=DEV).manual_seed(0)
    m = (1 << 21) + 4
    logits = torch.randn(m, 1, 2, device=DEV, generator=g)
    value = torch.randn(m, 1, device=DEV, generator=g)
    ret = value + torch.randn(m, 1, device=DEV, generator=g)
    action = torch.randint(0, 2, (m, 1), device=DEV, generator=g)
    logp = torch.randn(m, 1, device=DEV, generator=g) * 0.1 - 0.69
    adv = torch.randn(m, 1, device=DEV, generator=g)
    hp = hip.ppo_hparams(clip_param=0.2, dual_clip_param=None, entropy_coeff=0.01, vf_clip_param=5.0, vf_coeff=1.0,
                         grad_scale=1.0 / m)
    n, h = 1 << 16, 32
    rewards = torch.randn(h + 1, n, device=DEV, generator=g)
    values = torch.randn(h + 1, n, device=DEV, generator=g)
    adv_b, ret_b = torch.empty_like(rewards), torch.empty_like(rewards)
    busy = torch.randn(4096, 4096, device=DEV, generator=g)
    first = None
    for i in range(150):
        if i % 3 == 0:  # uneven load: GEMMs in flight next to the reductions
            busy @ busy
        sums, _, _ = hip.ppo_loss_categorical(logits, value, action, logp, adv, ret, hp, with_grad=i % 2 == 0)
        moments = hip.gae_scan(rewards, values, adv_b, ret_b, layout=1, n=n, h=h, gamma=0.95, gamma_lambda=0.9,
                               reward_denominator=1.0, write_scaled_rewards=False)
        stats = hip.rollout_stats(rewards.T.unsqueeze(-1), values.T.unsqueeze(-1))
        snap = (sums.clone(), moments.clone(), stats.clone())
        if first is None:
            first = snap
        for a, b in zip(first, snap):
            assert torch.equal(a, b), f"launch {i} differs from launch 0"
    sums, moments, stats = (t.cpu().numpy() for t in first)
    # independent check of the totals with torch reductions in fp64
    assert sums[3] == m
    assert moments[0] == n * h
    np.testing.assert_allclose(moments[1], float(adv_b[:h].double().sum()), rtol=1e-12)
    np.testing.assert_allclose(moments[2], float((adv_b[:h].double() ** 2).sum()), rtol=1e-12)
    np.testing.assert_allclose(stats[6], float(rewards[:h].double().sum()), rtol=1e-10, atol=1e-6)
    np.testing.assert_allclose(stats[8], float(rewards[:h].min()), rtol=0)
    np.testing.assert_allclose(stats[9], float(rewards[:h].max()), rtol=0)
    vf_terms = torch.clamp(torch.nn.functional.smooth_l1_loss(value, ret, reduction="none"), 0, 5.0)
    np.testing.assert_allclose(sums[2], float(vf_terms.double().sum()), rtol=1e-6)


def test_gather_wide_rows_recurrent_states():
    rng = np.random.default_rng(8)
    n, h = 64, 16
    hidden = rng.standard_normal((h + 1, n, 1, 256)).astype(np.float32)  # time-major storage
    leaf = dev(hidden).transpose(0, 1)  # [N, H+1, 1, 256] view
    small = dev(rng.standard_normal((h + 1, n, 1)).astype(np.float32)).transpose(0, 1)
    seqs = oracle.permutation(n * h // 4, 3, 0)[:100]
    first_ids = seqs * 4
    wide, narrow = hip.gather_minibatch(dev(first_ids), h, [leaf, small])
    env, t = first_ids // h, first_ids % h
    assert np.array_equal(host(wide), hidden[t, env])
    assert np.array_equal(host(narrow), host(small)[env, t])


# --------------------------------------------------------------------------- #
# N1: fused MLP tower vs a plain PyTorch fp32 reference of the same op
# --------------------------------------------------------------------------- #
def _torch_tower(x, w1, b1, w2, b2, w3, b3):
    h1 = torch.relu(x @ w1.T + b1)
    h2 = torch.relu(h1 @ w2.T + b2)
    return h2 @ w3.T + b3, h1, h2


@pytest.mark.parametrize("m,d_in,n_out", [(1, 1, 2), (63, 1, 1), (64, 1, 2), (1000, 5, 3), (4097, 16, 8), (100_000, 1, 2)])
def test_mlp_tower_forward_matches_torch(m, d_in, n_out):
    g = torch.Generator(device=DEV).manual_seed(m + d_in)
    x = torch.randn(m, d_in, device=DEV, generator=g) * 30
    w1 = torch.randn(256, d_in, device=DEV, generator=g) * 0.5
    b1 = torch.randn(256, device=DEV, generator=g) * 0.1
    w2 = torch.randn(256, 256, device=DEV, generator=g) / 16
    b2 = torch.randn(256, device=DEV, generator=g) * 0.1
    w3 = torch.randn(n_out, 256, device=DEV, generator=g) / 16
    b3 = torch.randn(n_out, device=DEV, generator=g)
    want, h1w, h2w = _torch_tower(x.double(), w1.double(), b1.double(), w2.double(), b2.double(), w3.double(), b3.double())
    packed = hip.mlp_pack_w2(w2)
    out, h1, h2 = hip.mlp_tower_forward(x, w1, b1, packed, b2, w3, b3, save=True)
    scale = float(want.abs().max()) + 1e-6
    assert float((out.double() - want).abs().max()) / scale < 2e-6
    assert float((h1.double() - h1w).abs().max()) / (float(h1w.abs().max()) + 1e-6) < 1e-6
    assert float((h2.double() - h2w).abs().max()) / (float(h2w.abs().max()) + 1e-6) < 2e-6
    out2, none1, none2 = hip.mlp_tower_forward(x, w1, b1, packed, b2, w3, b3)
    assert none1 is None and none2 is None and torch.equal(out, out2)
    # as accurate as torch's own fp32 path
    ref32, _, _ = _torch_tower(x, w1, b1, w2, b2, w3, b3)
    err_ours = float((out.double() - want).abs().max())
    err_torch = float((ref32.double() - want).abs().max())
    assert err_ours <= 4 * err_torch + 1e-6 * scale


@pytest.mark.parametrize("m,d_in,n_out", [(1, 1, 2), (65, 1, 1), (1000, 5, 3), (40_000, 1, 2), (4097, 16, 8)])
def test_mlp_tower_backward_matches_autograd(m, d_in, n_out):
    g = torch.Generator(device=DEV).manual_seed(7 * m + d_in)
    x = torch.randn(m, d_in, device=DEV, generator=g) * 3
    params = {
        "w1": torch.randn(256, d_in, device=DEV, generator=g) * 0.5,
        "b1": torch.randn(256, device=DEV, generator=g) * 0.1,
        "w2": torch.randn(256, 256, device=DEV, generator=g) / 16,
        "b2": torch.randn(256, device=DEV, generator=g) * 0.1,
        "w3": torch.randn(n_out, 256, device=DEV, generator=g) / 16,
        "b3": torch.randn(n_out, device=DEV, generator=g),
    }
    dout = torch.randn(m, n_out, device=DEV, generator=g) / m
    ref = {k: v.double().requires_grad_(True) for k, v in params.items()}
    out_ref, _, _ = _torch_tower(x.double(), *(ref[k] for k in ("w1", "b1", "w2", "b2", "w3", "b3")))
    out_ref.backward(dout.double())
    out, h1, h2 = hip.mlp_tower_forward(x, params["w1"], params["b1"], hip.mlp_pack_w2(params["w2"]),
                                            params["b2"], params["w3"], params["b3"], save=True)
    grads = hip.mlp_tower_backward(x, h1, h2, dout, hip.mlp_pack_w2(params["w2"], transposed=True), params["w3"])
    for k in params:
        want = ref[k].grad
        scale = float(want.abs().max()) + 1e-12
        err = float((grads[k].double() - want).abs().max()) / scale
        assert err < 2e-5, (k, err)


@pytest.mark.parametrize("m", [1, 63, 64, 1000, 16384 + 7, 300_000])
def test_mlp_wgrad_matches_torch(m):
    g = torch.Generator(device=DEV).manual_seed(m)
    dz2 = torch.randn(m, 256, device=DEV, generator=g)
    h1 = torch.relu(torch.randn(m, 256, device=DEV, generator=g))
    want = dz2.double().t() @ h1.double()
    got = hip.mlp_wgrad(dz2, h1)
    scale = float(want.abs().max()) + 1e-9
    assert float((got.double() - want).abs().max()) / scale < 5e-6
    again = hip.mlp_wgrad(dz2, h1)
    assert torch.equal(got, again)  # fixed summation order
