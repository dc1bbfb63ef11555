"""End-to-end parity of ``Algorithm.collect()`` / ``.step()`` on the GPU against
traces of the real reference (tests/golden/trace_*.npz: initial weights, reset
state, the noise torch drew per timestep, the minibatch permutations -> buffer
after collect, CollectStats, StepStats, final weights), plus the reference's own
behavioural tests (tests/test_algorithms.py of the reference).

Bars: action indices bit-exact; buffer floats and losses to 1e-5 relative
(north_star). The policy network runs on rocBLAS here and on MKL in the
reference, so logits differ in the last ulps; everything downstream is compared
with tolerances that reflect fp32 GEMM reordering, not kernel error.
"""

import json
import math
import os
from unittest.mock import patch

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from rl8_amd import AlgorithmConfig  # noqa: E402
from rl8_amd.data import DataKeys  # noqa: E402
from rl8_amd.distributions import SquashedNormal  # noqa: E402
from rl8_amd.env import ContinuousDummyEnv, DiscreteDummyEnv  # noqa: E402
from rl8_amd.tensordict import TensorDict  # noqa: E402

NUM_ENVS = 64
HORIZON = 32


def build_from_trace(g, env_cls, **config):
    algo = AlgorithmConfig(num_envs=NUM_ENVS, horizon=HORIZON, **config).build(env_cls)
    state = {k[len("init_"):]: torch.from_numpy(g[k]) for k in g if k.startswith("init_")}
    algo.policy.model.load_state_dict(state)
    return algo


def inject(algo, g, it):
    """Make collect()/step() consume the reference's recorded randomness."""
    dev = algo.policy.device
    if f"it{it}_reset_state" in g:
        reset_state = torch.from_numpy(g[f"it{it}_reset_state"]).to(dev)
        real_reset = algo.env.reset

        def reset(*, config=None):
            real_reset(config=config)
            algo.env.state.copy_(reset_state)
            if type(algo.env).__name__ == "CartPole":  # the reference's reset returns observations of the state
                x, x_dot, theta, theta_dot = algo.env.state
                return torch.vstack((x, x_dot, torch.cos(theta), torch.sin(theta), theta_dot)).T.contiguous()
            return algo.env.state

        algo.env.reset = reset
    key = f"it{it}_cat_q" if f"it{it}_cat_q" in g else f"it{it}_normal_eps"
    noise = g[key]
    # the reference draws once more per collect? no: H sampling calls per collect,
    # the (H+1)-th forward takes no sample.  validate() drew the first recorded
    # sample only in iteration 0 if build() ran inside the recorder (it did not).
    assert noise.shape[0] == HORIZON
    algo.injected_noise = torch.from_numpy(noise).to(dev)
    algo.injected_permutations = [torch.from_numpy(p) for p in g[f"it{it}_perms"]]


def compare_collect(algo, g, it, *, discrete, loose=None):
    # it == 0: weights are bit-identical to the reference's, so only GEMM
    # rounding separates the two runs.  it >= 1: the weights have been through
    # num_sgd_iters x num_minibatches Adam steps computed with a different GEMM
    # reduction order, so per-sample floats drift by a few 1e-5.
    if loose is None:
        loose = 1.0 if it == 0 else 5.0   # (round 6: 25 until the drift was re-measured -- profiles/r06_trace_drift_vs_round2_bands.json)
    buf = algo.buffer
    want = {k[len(f"it{it}_collect_"):]: g[k] for k in g if k.startswith(f"it{it}_collect_") and not k.endswith("stats")}
    got_actions = buf[DataKeys.ACTIONS][:, :HORIZON].cpu().numpy()
    if discrete:
        assert np.array_equal(got_actions, want["actions"][:, :HORIZON])
    else:
        close_arrays(f"it{it} buffer actions", got_actions, want["actions"][:, :HORIZON], rtol=1e-5, atol=1e-5)
    for key in ("obs", "rewards", "reversed_discounted_returns"):
        close_arrays(f"it{it} buffer {key}", buf[key].cpu().numpy(), want[key], rtol=1e-5, atol=1e-4, err_msg=key)
    close_arrays(f"it{it} buffer logp", buf[DataKeys.LOGP].cpu().numpy()[:, :HORIZON], want["logp"][:, :HORIZON],
                 rtol=1e-5 * loose, atol=(2e-5 if discrete else 5e-4) * loose)
    close_arrays(f"it{it} buffer values", buf[DataKeys.VALUES].cpu().numpy(), want["values"], rtol=1e-4 * loose,
                 atol=2e-5 * loose)


# Drift monitor: with RL8_TRACE_DRIFT_JSON=<path> every comparison of the free-running traces
# also records error / allowed (1.0 = at the band) under the running test's name; the file is rewritten after each
# comparison.  profiles/r06_trace_drift*.json are that file from this round's kernels.
_DRIFT: dict = {}


def _record_drift(label: str, err: float, allowed: float) -> None:
    path = os.environ.get("RL8_TRACE_DRIFT_JSON")
    if not path:
        return
    test = os.environ.get("PYTEST_CURRENT_TEST", "?").split("::")[-1].split(" ")[0]
    slot = _DRIFT.setdefault(test, {})
    ratio = err / allowed if allowed > 0 else (0.0 if err == 0 else float("inf"))
    if label not in slot or ratio > slot[label]["used"]:
        slot[label] = {"err": float(err), "allowed": float(allowed), "used": round(float(ratio), 4)}
    with open(path, "w") as f:
        json.dump(_DRIFT, f, indent=1, sort_keys=True)


def close_arrays(label, actual, desired, rtol, atol, err_msg=""):
    """np.testing.assert_allclose that also reports how much of its band the worst element used."""
    a, d = np.asarray(actual, dtype=np.float64), np.asarray(desired, dtype=np.float64)
    if a.shape == d.shape and a.size:
        used = np.abs(a - d) / (atol + rtol * np.abs(d))
        i = int(np.nanargmax(used))
        _record_drift(label, float(np.abs(a - d).reshape(-1)[i]), float((atol + rtol * np.abs(d)).reshape(-1)[i]))
    np.testing.assert_allclose(actual, desired, rtol=rtol, atol=atol, err_msg=err_msg or label)


def compare_stats(got, keys, want, rel, abs_tol=1e-7, label=""):
    for k, w in zip(keys, want):
        _record_drift(f"{label}{k}", abs(got[k] - w), max(rel * abs(w), abs_tol))
        assert got[k] == pytest.approx(w, rel=rel, abs=abs_tol), (k, got[k], w)


def run_trace(golden, name, env_cls, *, discrete, step_rel, drift=(10.0, 2e-5), weights_atol=(2e-5, 1e-4), **config):
    # Bands re-taken in round 6.  profiles/r06_trace_drift_vs_round2_bands.json holds, per comparison, how much of its
    # ROUND-2 band this round's kernels used: <= 0.3 % of every iteration-1 StepStats band and <= 1 % of every weight band
    # on the three full-batch traces (the towers now follow the reference's fp32 arithmetic to 1e-7..1e-6 through eight
    # Adam steps), 30 % / 79 % on the 64-Adam-step minibatch trace.  Bands beaten by more than 3x were cut by 10x
    # (StepStats: step_rel, the absolute floor of iteration 1, the weights' atol; the buffer's `loose` 25 -> 5), which
    # still leaves the full-batch traces a factor >= 20; the minibatch trace keeps its own bands.
    g = golden(name)
    algo = build_from_trace(g, env_cls, **config)
    for it in range(2):
        inject(algo, g, it)
        collect_stats = algo.collect()
        compare_collect(algo, g, it, discrete=discrete)
        # absolute band: rewards/max is -min|state| of states that are O(100) sums of
        # actions, so one fp32 ulp of a state (7.6e-6) is the floor for it.
        compare_stats(collect_stats, g["collect_stat_keys"], g[f"it{it}_collect_stats"], 1e-5, abs_tol=1e-5, label=f"it{it} collect ")
        assert algo.state.reward_scale == pytest.approx(float(g[f"it{it}_reward_scale"]), rel=1e-5)
        step_stats = algo.step()
        # The 1e-5 bar against the reference's own numbers is held where weights are
        # still bit-identical: tests/test_first_update_gpu.py (first StatTracker update
        # and first gradient of all six traced variants). These averages are taken over
        # 4-32 Adam steps; Adam's first steps move every weight by ~lr * sign(g), so
        # rounding-level differences in near-zero gradient entries become 1e-3 weight
        # differences. it >= 1 adds a second rollout on drifted weights; the policy loss
        # is a mean of O(1) terms that nearly cancel, so it also gets an absolute band.
        # (The reference alone, re-run with 1/2/4/8 MKL threads, moves its own
        # iteration-1 KL by 7e-4 relative on the minibatched config:
        # profiles/r02_reference_drift.json.)
        compare_stats(step_stats, g["step_stat_keys"], g[f"it{it}_step_stats"], step_rel * (1 if it == 0 else drift[0]),
                      1e-7 if it == 0 else drift[1], label=f"it{it} step ")
        final_obs = algo.buffer[DataKeys.OBS][:, -1].cpu().numpy()
        close_arrays(f"it{it} final_obs", final_obs, g[f"it{it}_final_obs"], rtol=1e-5, atol=1e-4)
        # the rest of the buffer was zeroed (reference re-allocates it, :603-609)
        assert float(algo.buffer[DataKeys.REWARDS].abs().sum()) == 0.0
        assert float(algo.buffer[DataKeys.OBS][:, :-1].abs().sum()) == 0.0
        sd = algo.policy.model.state_dict()
        for k, v in sd.items():
            # Adam moves each weight by <= lr (1e-3) per step whatever the gradient's
            # size, so rounding-level gradient differences show up at the 1e-4 level
            # after tens of optimizer steps.
            close_arrays(f"it{it} weights {k}", v.cpu().numpy(), g[f"it{it}_final_{k}"], rtol=2e-3,
                         atol=weights_atol[0] if it == 0 else weights_atol[1], err_msg=f"it{it} {k}")


def test_trace_feedforward_discrete_full_batch(golden):
    run_trace(golden, "trace_ff_discrete.npz", DiscreteDummyEnv, discrete=True, step_rel=1e-5)


def test_trace_feedforward_discrete_minibatches(golden):
    # 64 Adam steps on 256-sample minibatches by the end of iteration 1: the most drift-prone
    # trace.  Replayed with the towers evaluated four arithmetic-equivalent ways (eager rocBLAS,
    # fp32 MFMA, bf16 planes, fp16 planes), its iteration-1 averages spread over 3e-3 around the
    # reference's (profiles/r02_trace_gemm_modes.json; final weights of ALL four, eager
    # included, are 1e-3 = one Adam step off the reference's); from our own seeds the same
    # config differs by 5-18 % between ANY two of the four by iteration 1, eager vs fp32 MFMA
    # included (profiles/r02_gemm_mode_drift.json).  Iteration 0 is held to 1e-4 as before.
    run_trace(golden, "trace_ff_discrete_minibatch.npz", DiscreteDummyEnv, discrete=True, step_rel=1e-4,
              drift=(50.0, 5e-4), weights_atol=(2e-5, 1e-3), sgd_minibatch_size=256, entropy_coeff=1e-2, dual_clip_param=5.0, horizons_per_env_reset=2)


def test_trace_feedforward_continuous_squashed(golden):
    run_trace(golden, "trace_ff_continuous_squashed.npz", ContinuousDummyEnv, discrete=False, step_rel=1e-4,
              distribution_cls=SquashedNormal)


def test_trace_feedforward_continuous_normal_entropy(golden):
    run_trace(golden, "trace_ff_continuous_normal.npz", ContinuousDummyEnv, discrete=False, step_rel=1e-4,
              entropy_coeff=1e-2)


def test_first_sgd_iteration_losses_match_reference_to_1e5(golden):
    """Before any optimizer step the weights are the reference's, so the loss of
    the first SGD iteration isolates kernel + GEMM error: 1e-5 relative."""
    g = golden("trace_ff_discrete.npz")
    algo = build_from_trace(g, DiscreteDummyEnv, num_sgd_iters=1)
    inject(algo, g, 0)
    algo.collect()
    got = algo.step()
    # reference: same config but 4 iterations -> recompute its first-iteration
    # loss with the oracle from the recorded buffer
    from oracle import oracle
    from oracle.ppo_cpu import first_iteration_losses

    want = first_iteration_losses(g, 0, oracle)
    for k in ("losses/policy", "losses/vf", "losses/total", "monitors/kl_div"):
        assert got[k] == pytest.approx(want[k], rel=1e-5, abs=1e-7), k


# --- behaviour tests mirrored from the reference's tests/test_algorithms.py ---
@pytest.mark.parametrize("env_cls", [ContinuousDummyEnv, DiscreteDummyEnv])
def test_accumulated_gradients_match_full_batch(env_cls):
    # reference tests/test_algorithms.py:16-68 (rel_tol 1e-5)
    def run(**kw):
        torch.manual_seed(42)
        algo = AlgorithmConfig(num_envs=NUM_ENVS, horizon=HORIZON, entropy_coeff=1e-2, **kw).build(env_cls)
        algo.collect()
        return algo.step()

    full = run()
    accumulated = run(accumulate_grads=True, sgd_minibatch_size=NUM_ENVS)
    for k in ("losses/entropy", "losses/policy", "losses/total", "losses/vf", "monitors/kl_div"):
        assert math.isclose(full[k], accumulated[k], rel_tol=1e-5), k


@pytest.mark.parametrize("env_cls", [ContinuousDummyEnv, DiscreteDummyEnv])
def test_algorithm_validate(env_cls):
    algo = AlgorithmConfig(horizon=HORIZON, num_envs=NUM_ENVS).build(env_cls)
    algo.validate()


def test_feedforward_algorithm_resets():
    # reference tests/test_algorithms.py:85-100
    algo = AlgorithmConfig(horizon=HORIZON, num_envs=NUM_ENVS, horizons_per_env_reset=2).build(DiscreteDummyEnv)
    with patch.object(DiscreteDummyEnv, "reset", wraps=algo.env.reset) as reset:
        algo.collect()
        assert algo.state.horizons == 1 and reset.call_count == 1
        algo.collect()
        assert algo.state.horizons == 2 and reset.call_count == 1
        algo.collect()
        assert algo.state.horizons == 3 and reset.call_count == 2


def test_step_requires_collect_and_hparam_errors():
    algo = AlgorithmConfig(horizon=4, num_envs=8).build(DiscreteDummyEnv)
    with pytest.raises(RuntimeError, match="is not buffered"):
        algo.step()
    algo.collect()
    algo.step()
    with pytest.raises(RuntimeError):
        algo.step()
    with pytest.raises(ValueError, match="sgd_minibatch_size"):
        AlgorithmConfig(horizon=4, num_envs=8, sgd_minibatch_size=5).build(DiscreteDummyEnv)
    with pytest.raises(ValueError, match="clip_param"):
        AlgorithmConfig(horizon=4, num_envs=8, clip_param=1.5).build(DiscreteDummyEnv)


def test_collect_stats_keys_and_types():
    algo = AlgorithmConfig(horizon=8, num_envs=128).build(DiscreteDummyEnv)
    stats = algo.collect()
    assert set(stats) == {
        "env/resets", "env/steps", "profiling/collect_ms", "returns/min", "returns/max", "returns/mean",
        "returns/std", "rewards/min", "rewards/max", "rewards/mean", "rewards/std",
    }
    assert stats["env/steps"] == 128 * 8 and stats["env/resets"] == 128
    step = algo.step()
    assert set(step) == {
        "coefficients/entropy", "coefficients/vf", "losses/entropy", "losses/policy", "losses/vf",
        "losses/total", "monitors/kl_div", "profiling/step_ms",
    }
    assert all(isinstance(v, float) for v in step.values())


def test_fused_and_generic_rollouts_agree():
    """The one-launch-per-timestep path and the generic policy.sample -> env.step ->
    scatter path must fill identical buffers from the same Philox stream."""
    def run(force_generic):
        torch.manual_seed(7)
        algo = AlgorithmConfig(horizon=16, num_envs=512).build(DiscreteDummyEnv)
        if force_generic:
            algo._fusable = lambda: False
        algo.collect()
        return {k: v.clone() for k, v in algo.buffer.items()}

    a, b = run(False), run(True)
    for k in a:
        assert torch.equal(a[k], b[k]), k


def test_row_chunking_does_not_change_the_update():
    def run(rows):
        torch.manual_seed(3)
        algo = AlgorithmConfig(horizon=16, num_envs=256).build(DiscreteDummyEnv)
        algo.max_rows_per_pass = rows
        algo.collect()
        stats = algo.step()
        return stats, [p.detach().clone() for p in algo.policy.model.parameters()]

    s1, p1 = run(1 << 22)
    s2, p2 = run(1000)
    for k in ("losses/policy", "losses/vf", "losses/total", "monitors/kl_div"):
        assert s1[k] == pytest.approx(s2[k], rel=1e-5), k
    for a, b in zip(p1, p2):
        torch.testing.assert_close(a, b, rtol=1e-3, atol=1e-5)


def test_early_stopping_on_kl():
    torch.manual_seed(0)
    algo = AlgorithmConfig(horizon=16, num_envs=256, target_kl_div=1e-9, num_sgd_iters=8).build(DiscreteDummyEnv)
    algo.collect()
    before = [p.detach().clone() for p in algo.policy.model.parameters()]
    stats = algo.step()
    # first minibatch: kl == 0 exactly (same weights) -> update happens; the
    # second iteration's kl > 1.5e-9 stops the loop before its update.
    assert stats["monitors/kl_div"] >= 0.0
    after = [p.detach().clone() for p in algo.policy.model.parameters()]
    assert any(not torch.equal(a, b) for a, b in zip(before, after))


# --- CartPole (BASELINE config 3 shape, small) --------------------------------
def test_cartpole_env_api_and_fused_rollout():
    from oracle import oracle
    from rl8_amd.envs.cartpole import CartPole

    env = CartPole(1000, 64, device="cuda:0")
    obs = env.reset()
    assert obs.shape == (1000, 5) and env.state.shape == (4, 1000)
    want_state = oracle.cartpole_reset(1000, 0.01, env.seed, 0, 0)
    assert np.array_equal(env.state.cpu().numpy(), want_state)
    actions = torch.randint(0, 3, (1000, 1), device="cuda:0")
    out = env.step(actions)
    s2, obs2, rew2 = oracle.cartpole_step(want_state, actions.cpu().numpy(), oracle.cartpole_cfg())
    np.testing.assert_allclose(out["obs"].cpu().numpy(), obs2, atol=1e-6)
    np.testing.assert_allclose(out["rewards"].cpu().numpy(), rew2, rtol=1e-6, atol=1e-6)
    with pytest.raises(ValueError, match="horizon"):
        CartPole(8, 129, device="cuda:0")

    def run(force_generic):
        torch.manual_seed(11)
        algo = AlgorithmConfig(horizon=64, num_envs=2048).build(CartPole)
        if force_generic:
            algo._fusable = lambda: False
        stats = algo.collect()
        buf = {k: v.clone() for k, v in algo.buffer.items()}
        step = algo.step()
        return stats, buf, step

    s_f, b_f, st_f = run(False)
    s_g, b_g, st_g = run(True)
    for k in b_f:
        assert torch.equal(b_f[k], b_g[k]), k
    assert s_f["returns/mean"] == s_g["returns/mean"]
    assert st_f["losses/total"] == pytest.approx(st_g["losses/total"], rel=1e-5)
    assert AlgorithmConfig(horizon=500, num_envs=8).build(CartPole).hparams.horizon == 128


def test_cartpole_learns():
    """A few updates on the real task: mean return must improve (the reference's
    README claims CartPole is solved in seconds)."""
    from rl8_amd.envs.cartpole import CartPole

    torch.manual_seed(0)
    algo = AlgorithmConfig(horizon=64, num_envs=4096).build(CartPole)
    first = algo.collect()["returns/mean"]
    algo.step()
    for _ in range(30):
        last = algo.collect()["returns/mean"]
        algo.step()
    assert last > first + 0.2 * abs(first), (first, last)


SEEDS = 6


def learning_curve_ends(env_cls, iterations, monkeypatch, **config):
    """(first, last) mean return of a short run, for six seeds, under the shipped weight-gradient planes (fp16 under the
    guard) and under the exact bf16 planes (RL8_WGRAD_PLANES / RL8_WGRAD_GATE_PLANES = bf16)."""
    ends = {}
    for planes in ("f16", "bf16"):
        for name in ("RL8_WGRAD_PLANES", "RL8_WGRAD_GATE_PLANES"):
            if planes == "bf16":
                monkeypatch.setenv(name, "bf16")
            else:
                monkeypatch.delenv(name, raising=False)
        for seed in range(SEEDS):
            torch.manual_seed(seed)
            algo = AlgorithmConfig(**config).build(env_cls)
            first = algo.collect()["returns/mean"]
            algo.step()
            for _ in range(iterations):
                last = algo.collect()["returns/mean"]
                algo.step()
            ends[planes, seed] = (first, last)
    return ends


def assert_same_learning(ends, gain):
    """Every run improves by `gain`; the two plane schemes end within seed noise of each other (the
    fp16 planes must not change what is learned).  A run is chaotic in its last digits -- any difference in arithmetic
    moves a 40-iteration curve by as much as another seed does (a round-4 diagnostic, profiles/r04_experiments.md: eight seeds,
    -366 +- 55 against -356 +- 64) -- so the comparison is of the two MEANS against their standard errors."""
    for (planes, seed), (first, last) in ends.items():
        assert last > first + gain * abs(first), (planes, seed, first, last)
    last = {p: np.array([ends[p, s][1] for s in range(SEEDS)]) for p in ("f16", "bf16")}
    gap = abs(last["f16"].mean() - last["bf16"].mean())
    noise = math.sqrt(last["f16"].var(ddof=1) / SEEDS + last["bf16"].var(ddof=1) / SEEDS)
    assert gap <= 2.5 * noise + 0.03 * abs(last["bf16"].mean()), (ends, gap, noise)


def test_cartpole_learns_the_same_under_both_plane_schemes(monkeypatch):
    from rl8_amd.envs.cartpole import CartPole

    assert_same_learning(learning_curve_ends(CartPole, 30, monkeypatch, horizon=64, num_envs=4096), 0.2)


# --- recurrent algorithm (BASELINE config 5 shape, small) ----------------------
def run_recurrent_trace(golden, name, env_cls, *, discrete, **config):
    from rl8_amd import RecurrentAlgorithmConfig

    g = golden(name)
    algo = RecurrentAlgorithmConfig(num_envs=NUM_ENVS, horizon=HORIZON, **config).build(env_cls)
    state = {k[len("init_"):]: torch.from_numpy(g[k]) for k in g if k.startswith("init_")}
    algo.policy.model.load_state_dict(state)
    for it in range(2):
        inject(algo, g, it)
        collect_stats = algo.collect()
        loose = 1.0 if it == 0 else 5.0   # (25 until round 6: the recurrent traces used <= 2 % of those bands)
        buf = algo.buffer
        got_actions = buf[DataKeys.ACTIONS][:, :HORIZON].cpu().numpy()
        if discrete:
            assert np.array_equal(got_actions, g[f"it{it}_collect_actions"][:, :HORIZON])
        else:
            close_arrays(f"it{it} buffer actions", got_actions, g[f"it{it}_collect_actions"][:, :HORIZON], rtol=1e-4, atol=1e-4 * loose)
        for key in ("obs", "rewards", "reversed_discounted_returns"):
            close_arrays(f"it{it} buffer {key}", buf[key].cpu().numpy(), g[f"it{it}_collect_{key}"], rtol=1e-5, atol=2e-4 * loose, err_msg=key)
        close_arrays(f"it{it} buffer values", buf[DataKeys.VALUES].cpu().numpy(), g[f"it{it}_collect_values"], rtol=1e-4 * loose, atol=5e-5 * loose)
        for sk in ("hidden_states", "cell_states"):
            leaf = buf[DataKeys.STATES][sk].cpu().numpy()
            close_arrays(f"it{it} {sk} last", leaf[:, -1], g[f"it{it}_collect_states_{sk}_last"], rtol=1e-4 * loose, atol=2e-5 * loose)
            close_arrays(f"it{it} {sk} col6", leaf[:, 6], g[f"it{it}_collect_states_{sk}_col6"], rtol=1e-4 * loose, atol=2e-5 * loose)
        compare_stats(collect_stats, g["collect_stat_keys"], g[f"it{it}_collect_stats"], 1e-5 * loose, 1e-7, label=f"it{it} collect ")
        assert algo.state.reward_scale == pytest.approx(float(g[f"it{it}_reward_scale"]), rel=1e-5 * loose)
        step_stats = algo.step()
        # (round 6: 1e-3 / 1e-2 relative with floors 2e-6 / 1e-4 until the drift was re-measured: <= 0.4 % of those used)
        compare_stats(step_stats, g["step_stat_keys"], g[f"it{it}_step_stats"], 1e-4 * (1 if it == 0 else 10),
                      1e-6 if it == 0 else 1e-5, label=f"it{it} step ")
        # final states survive the buffer reset (:645-646)
        for sk in ("hidden_states", "cell_states"):
            leaf = algo.buffer[DataKeys.STATES][sk]
            assert float(leaf[:, :-1].abs().sum()) == 0.0
            np.testing.assert_allclose(leaf[:, -1].cpu().numpy(), g[f"it{it}_collect_states_{sk}_last"],
                                       rtol=1e-4 * loose, atol=2e-5 * loose)
    assert algo.state.seqs == 2 * (HORIZON // algo.hparams.seq_len)


def test_trace_recurrent_discrete(golden):
    run_recurrent_trace(golden, "trace_rec_discrete.npz", DiscreteDummyEnv, discrete=True)


def test_trace_recurrent_continuous_minibatches(golden):
    run_recurrent_trace(golden, "trace_rec_continuous_minibatch.npz", ContinuousDummyEnv, discrete=False,
                        sgd_minibatch_size=128, entropy_coeff=1e-2, seq_len=8, seqs_per_state_reset=2,
                        horizons_per_env_reset=2)


@pytest.mark.parametrize("env_cls", [ContinuousDummyEnv, DiscreteDummyEnv])
def test_recurrent_accumulated_gradients_match_full_batch(env_cls):
    from rl8_amd import RecurrentAlgorithmConfig

    def run(**kw):
        torch.manual_seed(42)
        algo = RecurrentAlgorithmConfig(num_envs=NUM_ENVS, horizon=HORIZON, entropy_coeff=1e-2, **kw).build(env_cls)
        algo.collect()
        return algo.step()

    full = run()
    accumulated = run(accumulate_grads=True, sgd_minibatch_size=NUM_ENVS)
    for k in ("losses/entropy", "losses/policy", "losses/total", "losses/vf", "monitors/kl_div"):
        assert math.isclose(full[k], accumulated[k], rel_tol=1e-5), k


def test_recurrent_algorithm_resets():
    # reference tests/test_algorithms.py:103-125
    from rl8_amd import RecurrentAlgorithmConfig
    from rl8_amd.policies_recurrent import RecurrentPolicy

    algo = RecurrentAlgorithmConfig(horizon=HORIZON, num_envs=NUM_ENVS, seq_len=4, seqs_per_state_reset=8).build(DiscreteDummyEnv)
    with (
        patch.object(DiscreteDummyEnv, "reset", wraps=algo.env.reset) as reset,
        patch.object(RecurrentPolicy, "init_states", wraps=algo.policy.init_states) as init_states,
    ):
        algo.collect()
        assert algo.state.horizons == 1 and reset.call_count == 1
        assert algo.state.seqs == 8 and init_states.call_count == 1
        algo.collect()
        assert algo.state.horizons == 2 and reset.call_count == 2
        assert algo.state.seqs == 16 and init_states.call_count == 2


def test_recurrent_fused_and_generic_rollouts_agree():
    from rl8_amd import RecurrentAlgorithmConfig

    def run(force_generic):
        torch.manual_seed(7)
        algo = RecurrentAlgorithmConfig(horizon=16, num_envs=256, seqs_per_state_reset=4).build(DiscreteDummyEnv)
        if force_generic:
            algo._fusable = lambda: False
        algo.collect()
        return {k: v.clone() for k, v in algo.buffer.items() if torch.is_tensor(v)}

    a, b = run(False), run(True)
    for k in a:
        assert torch.equal(a[k], b[k]), k
    with pytest.raises(ValueError, match="seq_len"):
        RecurrentAlgorithmConfig(horizon=30, num_envs=8, seq_len=4).build(DiscreteDummyEnv)


def test_recurrent_lean_and_plumbed_rollouts_agree():
    """The lean per-timestep launch path of RecurrentAlgorithm.collect() (straight C-ABI
    calls, states written into the next buffer column) against the same rollout through
    policy.sample() + tensordicts: identical buffers, states and statistics, and the
    update that follows lands on identical losses."""
    from rl8_amd import RecurrentAlgorithmConfig
    from rl8_amd.algorithms._recurrent import _LeanRollout

    def run(lean):
        torch.manual_seed(9)
        algo = RecurrentAlgorithmConfig(horizon=32, num_envs=300, seq_len=4, seqs_per_state_reset=4,
                                        horizons_per_env_reset=2).build(DiscreteDummyEnv)
        assert _LeanRollout.available(algo)
        algo.lean_rollout = lean
        out = []
        for _ in range(2):
            stats = algo.collect()
            buf = {k: v.clone() for k, v in algo.buffer.items() if torch.is_tensor(v)}
            states = {k: v.clone() for k, v in algo.buffer[DataKeys.STATES].items()}
            out.append((stats, buf, states, algo.step()))
        return out, algo.state.seqs

    (a, seqs_a), (b, seqs_b) = run(True), run(False)
    assert seqs_a == seqs_b
    for (s0, b0, st0, u0), (s1, b1, st1, u1) in zip(a, b):
        for k in b0:
            assert torch.equal(b0[k], b1[k]), k
        for k in st0:
            assert torch.equal(st0[k], st1[k]), k
        for k in s0:
            if not k.startswith("profiling"):
                assert s0[k] == s1[k], k
        for k in ("losses/policy", "losses/vf", "losses/total", "monitors/kl_div"):
            assert u0[k] == u1[k], k


# --- fused MLP towers (N1) -------------------------------------------------------
@pytest.mark.parametrize("env_cls", [ContinuousDummyEnv, DiscreteDummyEnv])
def test_fused_towers_match_eager_towers(env_cls):
    """Same seed, same Philox noise: a collect()+step() with the fused tower kernels
    must agree with the eager-PyTorch towers (fp32 reduction order differs)."""
    from rl8_amd.nn import fused_mlp

    def run(enabled):
        fused_mlp.ENABLED = enabled
        try:
            torch.manual_seed(5)
            algo = AlgorithmConfig(num_envs=256, horizon=16, entropy_coeff=0.0).build(env_cls)
            c = algo.collect()
            buf = {k: v.clone() for k, v in algo.buffer.items()}
            s = algo.step()
            return c, buf, s, [p.detach().clone() for p in algo.policy.model.parameters()]
        finally:
            fused_mlp.ENABLED = True

    c1, b1, s1, p1 = run(True)
    c0, b0, s0, p0 = run(False)
    if env_cls is DiscreteDummyEnv:
        assert torch.equal(b1[DataKeys.ACTIONS], b0[DataKeys.ACTIONS])
    torch.testing.assert_close(b1[DataKeys.VALUES], b0[DataKeys.VALUES], rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(b1[DataKeys.LOGP], b0[DataKeys.LOGP], rtol=1e-4, atol=1e-5)
    assert c1["returns/mean"] == pytest.approx(c0["returns/mean"], rel=1e-5)
    for k in ("losses/policy", "losses/vf", "losses/total", "monitors/kl_div"):
        assert s1[k] == pytest.approx(s0[k], rel=2e-4, abs=1e-6), k
    for a, b in zip(p1, p0):
        torch.testing.assert_close(a, b, rtol=5e-3, atol=2e-4)


# --- N4: a model with rolling-window view requirements ------------------------
def test_windowed_model_trains_like_the_plain_model():
    """A custom model that asks for the last 4 observations (padded rolling
    window) but reads only the newest one must reproduce the default model's run
    step for step: same rollout, same losses, same updated weights. Exercises
    apply_last in collect(), apply_all + minibatch indexing in step()."""
    from rl8_amd.models import DefaultDiscreteModel
    from rl8_amd.views import ViewRequirement

    seen = {}

    class WindowedModel(DefaultDiscreteModel):
        def __init__(self, observation_spec, action_spec, /, **config):
            super().__init__(observation_spec, action_spec, **config)
            self.view_requirements = {DataKeys.OBS: ViewRequirement(shift=3, method="padded_rolling_window")}

        def forward(self, batch, /):
            window = batch[DataKeys.OBS]
            inputs, mask = window[DataKeys.INPUTS], window[DataKeys.PADDING_MASK]
            seen[tuple(inputs.shape[1:])] = seen.get(tuple(inputs.shape[1:]), 0) + 1
            assert inputs.shape[1] == 4 and mask.shape == inputs.shape[:2] and not mask[:, -1].any()
            return super().forward(TensorDict({DataKeys.OBS: inputs[:, -1]}, batch_size=batch.batch_size))

    for kw in ({}, {"sgd_minibatch_size": 1024}):
        # same minibatch order for both runs
        perms = [torch.randperm(256 * 16, generator=torch.Generator().manual_seed(i)) for i in range(4)]

        def with_perms(model_cls):
            torch.manual_seed(5)
            algo = AlgorithmConfig(horizon=16, num_envs=256, model_cls=model_cls, **kw).build(DiscreteDummyEnv)
            assert algo._fusable() == (model_cls is DefaultDiscreteModel)
            out = []
            for _ in range(2):
                collect = algo.collect()
                buf = {k: v.clone() for k, v in algo.buffer.items()}
                algo.injected_permutations = perms
                out.append((collect, buf, algo.step()))
            return out, torch.cat([p.detach().flatten() for p in algo.policy.model.parameters()])

        plain, plain_params = with_perms(DefaultDiscreteModel)
        windowed, windowed_params = with_perms(WindowedModel)
        for (c0, b0, s0), (c1, b1, s1) in zip(plain, windowed):
            for k in b0:
                # the window's newest element is a strided slice, which the fused
                # towers decline -> eager GEMMs, last-bit differences in logits
                if b0[k].dtype == torch.int64:
                    assert torch.equal(b0[k], b1[k]), k
                else:
                    torch.testing.assert_close(b0[k], b1[k], rtol=1e-5, atol=1e-6, msg=k)
            assert c0["returns/mean"] == c1["returns/mean"]
            for k in ("losses/policy", "losses/vf", "losses/total", "monitors/kl_div"):
                assert s0[k] == pytest.approx(s1[k], rel=1e-5, abs=1e-8), k
        torch.testing.assert_close(plain_params, windowed_params, rtol=1e-4, atol=1e-6)
    assert (4, 1) in seen


def test_windowed_model_that_drops_samples_is_rejected():
    from rl8_amd.models import DefaultDiscreteModel
    from rl8_amd.views import ViewRequirement

    class DroppingModel(DefaultDiscreteModel):
        def __init__(self, observation_spec, action_spec, /, **config):
            super().__init__(observation_spec, action_spec, **config)
            self.view_requirements = {DataKeys.OBS: ViewRequirement(shift=2, method="rolling_window")}

        def forward(self, batch, /):
            return super().forward(TensorDict({DataKeys.OBS: batch[DataKeys.OBS][:, -1]}, batch_size=batch.batch_size))

    algo = AlgorithmConfig(horizon=8, num_envs=64, model_cls=DroppingModel).build(DiscreteDummyEnv)
    algo.collect()
    with pytest.raises(ValueError, match="one window per sample"):
        algo.step()


def test_enable_amp_keeps_the_fused_fp32_towers():
    """`enable_amp=True` (what the reference's example scripts pass on a GPU) must
    not fall off the fused path: same kernels, same numbers as fp32 up to the
    grad-scaler's power-of-two scaling."""
    from rl8_amd import hip

    def run(amp):
        torch.manual_seed(3)
        algo = AlgorithmConfig(num_envs=512, horizon=16, enable_amp=amp).build(DiscreteDummyEnv)
        hip.timer.reset()
        hip.timer.enabled = True
        collect = algo.collect()
        step = algo.step()
        hip.timer.enabled = False
        return collect, step, set(hip.timer.summary()), torch.cat([p.detach().flatten() for p in algo.policy.model.parameters()])

    c0, s0, k0, p0 = run(False)
    c1, s1, k1, p1 = run(True)
    # (two actions: policy and value tower both take the gate-plane weight-gradient kernel)
    assert {"mlp_tower_forward", "mlp_tower_forward_save", "mlp_tower_backward_gate", "mlp_wgrad_gate"} <= k1 and k0 == k1
    assert c0["returns/mean"] == c1["returns/mean"]
    for k in ("losses/policy", "losses/vf", "losses/total"):
        assert s0[k] == pytest.approx(s1[k], rel=1e-5, abs=1e-8), k
    torch.testing.assert_close(p0, p1, rtol=1e-4, atol=1e-6)


def test_first_sgd_iteration_of_a_two_action_policy_stores_no_h2():
    """Algorithm knows its policy is a two-way Categorical under the fused loss (logit gradients exact negatives by
    construction) and says so to the tower: from the FIRST iteration of a fresh model the training forwards keep the
    gate bits only and both towers' backward passes run in gate mode."""
    from rl8_amd import hip

    torch.manual_seed(0)
    algo = AlgorithmConfig(num_envs=256, horizon=8, num_sgd_iters=2).build(DiscreteDummyEnv)
    stored = []
    real = hip.mlp_tower_forward_split

    def spy(*a, **k):
        out = real(*a, **k)
        if k.get("save"):
            stored.append(out[2] is not None)
        return out

    hip.timer.reset()
    hip.timer.enabled = True
    try:
        with patch.object(hip, "mlp_tower_forward_split", spy):
            algo.collect()
            # (round 4) the rollout's own launches keep the gate bits of both towers for SGD iteration 0
            assert stored == [False] * 16, stored     # 8 timesteps x (policy, value): gate bits, never h2
            del stored[:]
            algo.step()
        launched = hip.timer.summary()
    finally:
        hip.timer.enabled = False
    assert stored == [False] * 2, stored              # iteration 1 x (policy, value) towers: never h2; iteration 0 replays
    assert "mlp_tower_backward_gate" in launched and "mlp_wgrad_gate" in launched
    assert "mlp_tower_backward" not in launched and "mlp_wgrad" not in launched


def test_pair_check_is_skipped_only_for_the_loss_kernels_own_gradient():
    """The two-class loss kernel stores g and -g, and ``fused_ppo_loss`` vouches for exactly that tensor: the policy
    tower's backward then neither launches ``rl8_mlp_dout_pair_check`` nor reads its answer (a host round trip per
    backward).  Without the registration -- or when the gradient reaching the tower is another tensor, e.g. modified in
    place after the loss -- the check runs as before; the update is bit for bit the same either way."""
    from rl8_amd import hip
    from rl8_amd.nn import fused_mlp

    def run(register, touch=False):
        torch.manual_seed(3)
        algo = AlgorithmConfig(num_envs=512, horizon=8, num_sgd_iters=3).build(DiscreteDummyEnv)
        algo.collect()
        checks = []
        real_flag, real_trust = hip._pair_flag, fused_mlp.trust_pair_gradient

        def trust(g):
            if register:
                real_trust(g)
                if touch:
                    g.mul_(1.0)  # (same values, new version: no longer the tensor that was vouched for)

        with patch.object(hip, "_pair_flag", lambda dev: checks.append(1) or real_flag(dev)), \
                patch.object(fused_mlp, "trust_pair_gradient", trust):
            stats = algo.step()
        assert not fused_mlp._TRUSTED_PAIRS  # (consumed or released with the step)
        return len(checks), stats, torch.cat([p.detach().flatten() for p in algo.policy.model.parameters()])

    n_trusted, s0, p0 = run(True)
    n_checked, s1, p1 = run(False)
    n_touched, s2, p2 = run(True, touch=True)
    assert n_trusted == 0 and n_checked == 3 and n_touched == 3, (n_trusted, n_checked, n_touched)
    assert torch.equal(p0, p1) and torch.equal(p0, p2)
    for k in s0:
        assert s0[k] == s1[k] or (isinstance(s0[k], float) and math.isnan(s0[k])) or k.startswith("profiling"), k


def test_recurrent_full_buffer_minibatch_is_gathered_once_in_buffer_order():
    """One minibatch = the whole buffer: the recurrent step() lays the sequences out once, in buffer order, and reads
    that copy in every SGD iteration -- bit for bit what injecting the identity permutation (a gather per iteration)
    gives; with several minibatches a fresh permutation is drawn per iteration as before."""
    from rl8_amd import RecurrentAlgorithmConfig

    def run(inject, **kw):
        torch.manual_seed(11)
        algo = RecurrentAlgorithmConfig(num_envs=96, horizon=32, **kw).build(DiscreteDummyEnv)
        out = []
        for _ in range(2):
            collect = algo.collect()
            if inject:
                seqs = 96 * (32 // algo.hparams.seq_len)
                algo.injected_permutations = [torch.arange(seqs) for _ in range(algo.hparams.num_sgd_iters)]
            out.append((collect, algo.step()))
            assert algo._flat_full is None
        return out, torch.cat([p.detach().flatten() for p in algo.policy.model.parameters()])

    once, once_params = run(False)
    every, every_params = run(True)
    for (c0, s0), (c1, s1) in zip(once, every):
        assert c0["returns/mean"] == c1["returns/mean"]
        for k in ("losses/policy", "losses/vf", "losses/total", "monitors/kl_div"):
            assert s0[k] == s1[k], k
    assert torch.equal(once_params, every_params)
    # several minibatches: shuffled (two runs with different generator states differ)
    torch.manual_seed(11)
    algo = RecurrentAlgorithmConfig(num_envs=96, horizon=32, sgd_minibatch_size=192).build(DiscreteDummyEnv)
    algo.collect()
    first = [b[DataKeys.OBS].clone() for b in algo._iter_minibatches(0)]
    second = [b[DataKeys.OBS].clone() for b in algo._iter_minibatches(1)]
    assert len(first) == 4 and not all(torch.equal(a, b) for a, b in zip(first, second))


def test_recurrent_step_with_wide_observations():
    """the recurrent step()'s whole-buffer copy (``rl8_gather_minibatch(index = NULL)``) with more
    than 128 bytes per sample -- a 40-float observation -- used to return RL8_ESIZE.  Same update as the indexed
    gather of the identity permutation, bit for bit (reference ``src/rl8/algorithms/_recurrent.py:510-518``: the
    default models take any 1-D observation width)."""
    from rl8_amd import RecurrentAlgorithmConfig

    from ._envs import walk_env

    def run(inject):
        torch.manual_seed(5)
        algo = RecurrentAlgorithmConfig(num_envs=48, horizon=32).build(walk_env(40, 2))
        algo.collect()
        if inject:
            seqs = 48 * (32 // algo.hparams.seq_len)
            algo.injected_permutations = [torch.arange(seqs) for _ in range(algo.hparams.num_sgd_iters)]
        stats = algo.step()
        return stats, torch.cat([p.detach().flatten() for p in algo.policy.model.parameters()])

    (s0, p0), (s1, p1) = run(False), run(True)
    for k in ("losses/policy", "losses/vf", "losses/total", "monitors/kl_div"):
        assert math.isfinite(s0[k]) and s0[k] == s1[k], k
    assert torch.equal(p0, p1)


@pytest.mark.parametrize("env_cls", [DiscreteDummyEnv, ContinuousDummyEnv])
def test_recurrent_algorithm_with_a_users_distribution(env_cls):
    """VERDICT r5 Missing #2: the reference builds ``self.policy.distribution_cls(features, model)`` and hands whatever
    class the user supplied to ``ppo_losses`` (``src/rl8/algorithms/_recurrent.py:560-585``).  (1) A subclass whose
    ``logp`` / ``entropy`` are overridden WITHOUT changing the maths leaves the fused loss kernel for the composed
    tensor-op loss through the eager heads, and must land on the built-in class's update (same seed, same Philox
    noise); (2) a distribution with different maths (tempered logits) trains: finite losses, gradients reach the LSTM,
    and its update differs from the built-in one."""
    from rl8_amd import RecurrentAlgorithmConfig
    from rl8_amd.distributions import Categorical, Normal
    from rl8_amd.nn.functional import has_fused_loss

    base = Categorical if env_cls is DiscreteDummyEnv else Normal

    class SameMaths(base):
        def logp(self, samples):
            return super().logp(samples)

        def entropy(self):
            return super().entropy()

    class Tempered(base):
        def logp(self, samples):
            if base is Categorical:
                nl = torch.log_softmax(self.logits / 2.0, -1)
                return nl.gather(-1, samples.long().unsqueeze(-1)).squeeze(-1).sum(-1, keepdim=True)
            return 0.5 * super().logp(samples)

    assert has_fused_loss(base) and not has_fused_loss(SameMaths) and not has_fused_loss(Tempered)

    def run(dist_cls):
        torch.manual_seed(21)
        algo = RecurrentAlgorithmConfig(num_envs=96, horizon=32, seq_len=4, entropy_coeff=1e-2,
                                        distribution_cls=dist_cls).build(env_cls)
        before = {k: v.detach().clone() for k, v in algo.policy.model.named_parameters()}
        algo.collect()
        buf = {k: v.clone() for k, v in algo.buffer.items() if torch.is_tensor(v)}
        stats = algo.step()
        after = dict(algo.policy.model.named_parameters())
        moved = {k: float((after[k].detach() - before[k]).abs().max()) for k in before}
        return stats, buf, moved, torch.cat([p.detach().flatten() for p in after.values()])

    s0, b0, m0, p0 = run(base)
    s1, b1, m1, p1 = run(SameMaths)
    for k in b0:  # the rollout samples through the built-in sampler either way (same Philox stream)
        assert torch.equal(b0[k], b1[k]), k
    for k in ("losses/entropy", "losses/policy", "losses/vf", "losses/total", "monitors/kl_div"):
        assert math.isclose(s0[k], s1[k], rel_tol=2e-5, abs_tol=1e-7), (k, s0[k], s1[k])
    np.testing.assert_allclose(p1.cpu().numpy(), p0.cpu().numpy(), rtol=0, atol=2e-5)
    s2, _, m2, p2 = run(Tempered)
    for k in ("losses/policy", "losses/vf", "losses/total", "monitors/kl_div"):
        assert math.isfinite(s2[k]), k
    lstm_keys = [k for k in m2 if "lstm" in k]
    assert lstm_keys and all(m2[k] > 0 for k in lstm_keys), m2
    assert not torch.equal(p0, p2)


@pytest.mark.parametrize("d,a", [(4, 3), (7, 2), (6, 4)])
def test_recurrent_walk_env_of_four_six_seven_observations_runs_the_plane_lstm(d, a, monkeypatch):
    """Round 6: the LSTM's plane kernels (fp16-plane step, backward through time, four-gate weight gradient) take every
    observation width the [w_ih | bias] rows hold (d_in <= 7; 1, 2, 3, 5 until then).  Two collect() + step() rounds of
    the recurrent algorithm on the tests' walk environment against the SAME seeded run on the fp32-MFMA LSTM kernels
    (RL8_AMD_LSTM_GEMM=f32's switch): rollouts' returns to 1e-4, losses to 2e-3."""
    from rl8_amd import RecurrentAlgorithmConfig, hip
    from rl8_amd.nn import fused_lstm

    from ._envs import walk_env

    assert hip.lstm_split_supports(d)
    steps = []
    real = hip.lstm_forward_split

    def spy(*args, **kw):
        steps.append(args[0].shape[-1])
        return real(*args, **kw)

    def run(gemm):
        monkeypatch.setattr(fused_lstm, "FORWARD_GEMM", gemm)
        torch.manual_seed(13)
        algo = RecurrentAlgorithmConfig(num_envs=512, horizon=32).build(walk_env(d, a))
        return [(algo.collect(), algo.step()) for _ in range(2)]

    monkeypatch.setattr(hip, "lstm_forward_split", spy)
    planes = run("split")
    assert steps and all(w == d for w in steps), steps
    monkeypatch.setattr(hip, "lstm_forward_split", real)
    f32 = run("f32")
    for (c0, s0), (c1, s1) in zip(f32, planes):
        assert c1["returns/mean"] == pytest.approx(c0["returns/mean"], rel=1e-4)
        for k in ("losses/policy", "losses/vf", "losses/total"):
            assert s1[k] == pytest.approx(s0[k], rel=2e-3, abs=2e-6), k


@pytest.mark.parametrize("d,a", [(7, 3), (6, 4), (7, 2)])
def test_walk_env_with_six_and_seven_observations_trains_on_the_plane_kernels(d, a, monkeypatch):
    """Round 6 (VERDICT r5 missing #3, in part): d_in 6 and 7 (n_out <= 4; 7 x 4 excepted) train on the plane kernels end to
    end -- class-8 data gradients, weight gradients compiled per width with their scalar loads made in front of the wait
    -- like the 4- and 5-wide towers: three collect() + step() rounds against the SAME seeded run on the fp32-MFMA towers
    (returns to 1e-4, losses to 2e-3, the return improves), and every backward call carried the gate bits of h2, which
    only the plane path takes."""
    from rl8_amd import AlgorithmConfig, hip
    from rl8_amd.nn import fused_mlp

    from ._envs import walk_env

    assert hip.mlp_forward_f16_supports(d, a) and hip.mlp_backward_f16_supports(d, a) and hip.mlp_backward_f16_supports(d, 1)
    calls = []
    real = hip.mlp_tower_backward

    def spy(*args, **kw):
        calls.append(kw.get("gate2") is not None)
        return real(*args, **kw)

    def run(gemm):
        monkeypatch.setattr(fused_mlp, "FORWARD_GEMM", gemm)
        monkeypatch.setattr(fused_mlp, "BACKWARD_GEMM", gemm)
        torch.manual_seed(11)
        algo = AlgorithmConfig(num_envs=2048, horizon=16).build(walk_env(d, a))
        return [(algo.collect(), algo.step()) for _ in range(3)]

    monkeypatch.setattr(hip, "mlp_tower_backward", spy)
    planes = run("f16")
    assert calls and all(calls), calls
    monkeypatch.setattr(hip, "mlp_tower_backward", real)
    f32 = run("f32")
    for (c0, s0), (c1, s1) in zip(f32, planes):
        assert c1["returns/mean"] == pytest.approx(c0["returns/mean"], rel=1e-4)
        for k in ("losses/policy", "losses/vf", "losses/total"):
            assert s1[k] == pytest.approx(s0[k], rel=2e-3, abs=2e-6), k
    assert planes[-1][0]["returns/mean"] > planes[0][0]["returns/mean"]


@pytest.mark.parametrize("d,a", [(12, 3), (7, 4), (16, 2)])
def test_walk_env_with_wide_observations_trains(d, a, monkeypatch):
    """Observations of 7, 12 and 16 floats (the tests' walk environment): rollouts on the plane forward's classes 8 / 16,
    training through the mixed path (plane forward with h1 / h2 stored, fp32-MFMA data gradient, bf16-plane weight
    gradient) -- three collect() + step() rounds against the SAME seeded run on the fp32-MFMA towers: rollouts' returns to
    1e-4, losses to 2e-3, and the return improves."""
    from rl8_amd import AlgorithmConfig, hip
    from rl8_amd.nn import fused_mlp

    from ._envs import walk_env

    def run(gemm):
        monkeypatch.setattr(fused_mlp, "FORWARD_GEMM", gemm)
        monkeypatch.setattr(fused_mlp, "BACKWARD_GEMM", gemm)
        torch.manual_seed(11)
        algo = AlgorithmConfig(num_envs=2048, horizon=16).build(walk_env(d, a))
        return [(algo.collect(), algo.step()) for _ in range(3)]

    assert hip.mlp_forward_f16_supports(d, a) and not hip.mlp_backward_f16_supports(d, a)
    planes, f32 = run("f16"), run("f32")
    for (c0, s0), (c1, s1) in zip(f32, planes):
        assert c1["returns/mean"] == pytest.approx(c0["returns/mean"], rel=1e-4)
        for k in ("losses/policy", "losses/vf", "losses/total"):
            assert s1[k] == pytest.approx(s0[k], rel=2e-3, abs=2e-6), k
    assert planes[-1][0]["returns/mean"] > planes[0][0]["returns/mean"]
