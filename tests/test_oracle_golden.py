"""Pins the CPU oracle (oracle/rl8_oracle.c) to the golden vectors produced by
the real reference (tests/golden/generate_fixtures.py).

Bars: bit-exact for integer outputs (action indices) and for arithmetic that is
add/mul/div only (dummy env, GAE incl. normalisation); 1e-6 for CartPole (the
reference's own compiled vs eager step differ by 1 ulp); 1e-5
relative for losses (north_star).
"""

import numpy as np
import pytest

from oracle import oracle

RTOL = 1e-5


def test_philox_known_answers():
    # Random123 known-answer vectors for philox4x32-10 (Salmon et al., SC'11).
    def words(ctr, key):
        seed = key[0] | (key[1] << 32)
        row = ctr[0] | (ctr[1] << 32)
        return oracle.philox_words(seed, row, ctr[2], ctr[3]).tolist()

    assert words([0, 0, 0, 0], [0, 0]) == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    f = 0xFFFFFFFF
    assert words([f, f, f, f], [f, f]) == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    assert words([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0]) == [
        0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1,
    ]


def test_gae_matches_reference_bit_exact(golden):
    g = golden("gae.npz")
    for case in g["cases"]:
        gamma, lam, scale, norm = g[f"{case}_params"]
        out = oracle.gae(
            g[f"{case}_rewards"], g[f"{case}_values"], gamma=gamma, gae_lambda=lam,
            reward_scale=scale, normalize_advantages=bool(norm),
        )
        assert np.array_equal(out["scaled_rewards"], g[f"{case}_scaled_rewards"]), case
        assert np.array_equal(out["returns"], g[f"{case}_returns"]), case
        assert np.array_equal(out["advantages"], g[f"{case}_advantages"]), case


def test_gae_known_answer(golden):
    # Reference KAT: tests/test_nn/test_functional.py:14-49.
    g = golden("gae.npz")
    ones = np.ones((10, 6, 1), np.float32)
    out = oracle.gae(ones, ones, gamma=1, gae_lambda=1, reward_scale=1.0, normalize_advantages=False)
    undiscounted = np.flip(np.cumsum(ones, axis=1), axis=1)
    # reward_scale=1 divides by (1 + 1e-8) -> exactly 1.0f in fp32
    assert np.array_equal(out["advantages"], undiscounted - 1)
    assert np.array_equal(out["returns"], undiscounted)
    assert np.array_equal(out["advantages"], g["kat_advantages"])
    assert np.array_equal(out["returns"], g["kat_returns"])


def test_dummy_env_steps_bit_exact(golden):
    g = golden("env_steps.npz")
    for kind in ("disc", "cont"):
        state = g[f"{kind}_state0"]
        for t in range(g[f"{kind}_actions"].shape[0]):
            state, reward = oracle.dummy_env_step(state, g[f"{kind}_actions"][t])
            assert np.array_equal(state, g[f"{kind}_states"][t]), (kind, t)
            assert np.array_equal(reward, g[f"{kind}_rewards"][t]), (kind, t)


def test_cartpole_steps(golden):
    g = golden("env_steps.npz")
    for integ in ("euler", "semi-implicit"):
        cfg = oracle.cartpole_cfg(kinematics_integrator=integ)
        state = g[f"cp_{integ}_state0"]
        for t in range(g[f"cp_{integ}_actions"].shape[0]):
            # step from the reference's own previous state: per-step parity
            prev = g[f"cp_{integ}_state0"] if t == 0 else g[f"cp_{integ}_states"][t - 1]
            s, obs, rew = oracle.cartpole_step(prev, g[f"cp_{integ}_actions"][t], cfg)
            np.testing.assert_allclose(s, g[f"cp_{integ}_states"][t], rtol=0, atol=1e-6)
            np.testing.assert_allclose(obs, g[f"cp_{integ}_obs"][t], rtol=0, atol=1e-6)
            np.testing.assert_allclose(rew, g[f"cp_{integ}_rewards"][t], rtol=1e-6, atol=1e-6)
            # and chained from the start: drift stays within a few ulp
            state, _, _ = oracle.cartpole_step(state, g[f"cp_{integ}_actions"][t], cfg)
            np.testing.assert_allclose(state, g[f"cp_{integ}_states"][t], rtol=0, atol=1e-5)
    fm, grav, length, pm, pml, tm, tau = g["cp_custom_cfg"]
    cfg = oracle.cartpole_cfg(force_mag=fm, gravity=grav, length=length, pole_mass=pm,
                              pole_mass_length=pml, total_mass=tm, tau=tau)
    s, obs, rew = oracle.cartpole_step(g["cp_custom_state0"], g["cp_custom_actions"], cfg)
    np.testing.assert_allclose(s, g["cp_custom_state1"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(obs, g["cp_custom_obs"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(rew, g["cp_custom_rewards"], rtol=1e-6, atol=1e-6)


def test_mountain_car_steps(golden):
    g = golden("classic_env_steps.npz")
    for tag in ("mc_default", "mc_custom"):
        cfg = oracle.mountain_car_cfg(**dict(zip(g[f"{tag}_cfg_keys"].tolist(), g[f"{tag}_cfg"].tolist())))
        state = g[f"{tag}_state0"]
        walls = goals = 0
        for t in range(g[f"{tag}_actions"].shape[0]):
            prev = g[f"{tag}_state0"] if t == 0 else g[f"{tag}_states"][t - 1]
            s, obs, rew = oracle.mountain_car_step(prev, g[f"{tag}_actions"][t], cfg)
            # cos() is the only inexact op: last-ulp differences between libms
            np.testing.assert_allclose(s, g[f"{tag}_states"][t], rtol=0, atol=2e-8)
            np.testing.assert_allclose(obs, g[f"{tag}_obs"][t], rtol=0, atol=2e-8)
            np.testing.assert_allclose(rew[:, 0], g[f"{tag}_rewards"][t], rtol=0, atol=1e-7)
            walls += int((s[0] == np.float32(cfg.min_position)).sum())
            goals += int((rew == 1.0).sum())
            state, _, _ = oracle.mountain_car_step(state, g[f"{tag}_actions"][t], cfg)
            np.testing.assert_allclose(state, g[f"{tag}_states"][t], rtol=0, atol=1e-6)
        assert walls > 0 and goals > 0  # the fixture reaches the wall and the goal


def test_pendulum_steps(golden):
    g = golden("classic_env_steps.npz")
    for tag in ("pd_default", "pd_custom"):
        cfg = oracle.pendulum_cfg(**dict(zip(g[f"{tag}_cfg_keys"].tolist(), g[f"{tag}_cfg"].tolist())))
        state = g[f"{tag}_state0"]
        for t in range(g[f"{tag}_actions"].shape[0]):
            prev = g[f"{tag}_state0"] if t == 0 else g[f"{tag}_states"][t - 1]
            s, obs, rew = oracle.pendulum_step(prev, g[f"{tag}_actions"][t], cfg)
            np.testing.assert_allclose(s, g[f"{tag}_states"][t], rtol=0, atol=2e-6)
            np.testing.assert_allclose(obs, g[f"{tag}_obs"][t], rtol=0, atol=2e-6)
            np.testing.assert_allclose(rew[:, 0], g[f"{tag}_rewards"][t], rtol=1e-6, atol=1e-6)
            state, _, _ = oracle.pendulum_step(state, g[f"{tag}_actions"][t], cfg)
            np.testing.assert_allclose(state, g[f"{tag}_states"][t], rtol=0, atol=2e-5)


def test_classic_env_resets_are_in_range():
    s = oracle.mountain_car_reset(4096, seed=7, reset_count=0)
    assert abs(float(s[0].mean()) + 0.5) < 0.01 and abs(float(s[0].std()) - 0.05) < 0.005
    assert abs(float(s[1].mean())) < 0.01 and abs(float(s[1].std()) - 0.05) < 0.005
    s, obs = oracle.pendulum_reset(4096, seed=7, reset_count=0)
    assert float(np.abs(s[0]).max()) <= np.float32(np.pi) and float(np.abs(s[1]).max()) <= 1.0
    np.testing.assert_allclose(obs[:, 0], np.cos(s[0]), atol=1e-6)
    np.testing.assert_allclose(obs[:, 2], s[1])
    # a different reset counter / env offset gives different, reproducible draws
    s2, _ = oracle.pendulum_reset(4096, seed=7, reset_count=1)
    assert not np.array_equal(s, s2)
    s3, _ = oracle.pendulum_reset(2048, seed=7, reset_count=0, env_offset=2048)
    assert np.array_equal(s3, s[:, 2048:])


@pytest.mark.parametrize("ncls", [2, 3, 5])
def test_categorical_sampler_bit_exact_actions(golden, ncls):
    g = golden("samplers.npz")
    actions, logp = oracle.categorical_sample(g[f"cat{ncls}_logits"], g[f"cat{ncls}_q"])
    assert np.array_equal(actions, g[f"cat{ncls}_actions"])
    np.testing.assert_allclose(logp, g[f"cat{ncls}_logp"], rtol=RTOL, atol=1e-6)
    mode, _ = oracle.categorical_sample(g[f"cat{ncls}_logits"], deterministic=True)
    assert np.array_equal(mode, g[f"cat{ncls}_mode"])


@pytest.mark.parametrize("adim", [1, 3])
@pytest.mark.parametrize("kind", ["normal", "squashed"])
def test_normal_samplers(golden, kind, adim):
    g = golden("samplers.npz")
    p = f"{kind}{adim}"
    actions, logp = oracle.normal_sample(g[f"{p}_mean"], g[f"{p}_log_std"], g[f"{p}_eps"], squashed=kind == "squashed")
    np.testing.assert_allclose(actions, g[f"{p}_actions"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(logp, g[f"{p}_logp"], rtol=1e-4, atol=2e-4)
    mode, _ = oracle.normal_sample(g[f"{p}_mean"], g[f"{p}_log_std"], squashed=kind == "squashed", deterministic=True)
    np.testing.assert_allclose(mode, g[f"{p}_mode"], rtol=1e-6, atol=1e-6)


def _hp(arr):
    clip, dual, ent, vfclip, vfc = arr
    return oracle.ppo_hparams(clip_param=clip, dual_clip_param=dual or None, entropy_coeff=ent,
                              vf_clip_param=vfclip, vf_coeff=vfc)


def _check_losses(got, want, case):
    for i, k in enumerate(oracle.LOSS_KEYS):
        assert got[k] == pytest.approx(want[i], rel=RTOL, abs=1e-7), (case, k, got[k], want[i])


def test_ppo_losses_and_grads_match_reference_autograd(golden):
    g = golden("ppo_losses.npz")
    for case in g["cases"]:
        hp = _hp(g[f"{case}_hparams"])
        common = (g[f"{case}_values"], g[f"{case}_actions"], g[f"{case}_logp_old"],
                  g[f"{case}_advantages"], g[f"{case}_returns"])
        if case.startswith("cat"):
            losses, g_logits, g_values = oracle.ppo_loss_categorical(g[f"{case}_feat_logits"], *common, hp)
            np.testing.assert_allclose(g_logits, g[f"{case}_grad_logits"], rtol=2e-5, atol=1e-8, err_msg=case)
        else:
            losses, g_mean, g_ls, g_values = oracle.ppo_loss_normal(
                g[f"{case}_feat_mean"], g[f"{case}_feat_log_std"], *common, hp,
                squashed=case.startswith("squashed"),
            )
            np.testing.assert_allclose(g_mean, g[f"{case}_grad_mean"], rtol=1e-4, atol=1e-7, err_msg=case)
            np.testing.assert_allclose(g_ls, g[f"{case}_grad_log_std"], rtol=1e-4, atol=1e-7, err_msg=case)
        np.testing.assert_allclose(g_values, g[f"{case}_grad_values"], rtol=2e-5, atol=1e-9, err_msg=case)
        _check_losses(losses, g[f"{case}_losses"], case)


def test_rollout_stats_and_rdr_match_trace(golden):
    g = golden("trace_ff_discrete.npz")
    keys = list(g["collect_stat_keys"])
    want = dict(zip(keys, g["it0_collect_stats"]))
    got = oracle.rollout_stats(g["it0_collect_rewards"], g["it0_collect_reversed_discounted_returns"])
    for k in ("rewards/min", "rewards/max"):
        assert got[k] == want[k], k
    # per-env returns are fp32 sums over the horizon; torch's summation order
    # differs from a sequential loop by an ulp or two.
    for k in ("returns/min", "returns/max", "returns/mean", "returns/std", "rewards/mean", "rewards/std"):
        assert got[k] == pytest.approx(want[k], rel=1e-6), k
    assert got["reward_scale"] == pytest.approx(float(g["it0_reward_scale"]), rel=1e-6)
    # rdr recurrence, bit-exact column by column
    rdr = g["it0_collect_reversed_discounted_returns"]
    rew = g["it0_collect_rewards"]
    for t in range(rew.shape[1] - 1):
        assert np.array_equal(oracle.rdr_step(rdr[:, t], rew[:, t], 0.95), rdr[:, t + 1]), t


def test_gather_rows():
    rng = np.random.default_rng(0)
    src = rng.standard_normal((100, 3)).astype(np.float32)
    idx = rng.permutation(100)
    assert np.array_equal(oracle.gather_rows(idx, src), src[idx])
    perm = oracle.permutation(1000, 7, 3)
    assert sorted(perm.tolist()) == list(range(1000))
    assert not np.array_equal(perm, oracle.permutation(1000, 7, 4))
