"""The assembled path against numbers the REFERENCE itself produced, at
north_star's 1e-5 bar (tests/golden/first_update_*.npz, early_stop.npz; written by
tests/golden/generate_fixtures.py from the unmodified reference).

Before the first optimizer step the weights are bit-identical to the
reference's, so the first ``StatTracker.update`` of a ``step()``
(``src/rl8/algorithms/_feedforward.py:562-574``; recurrent twin
``_recurrent.py``) and the gradient handed to the first ``optimizer.step()``
(``:585-590``) isolate this build's kernels -- rollout, GAE, gather, towers /
LSTM forward and backward, fused loss -- from optimizer drift. All six traced
variants are held to that: feed-forward discrete (full batch / minibatched +
entropy + dual clip), feed-forward Normal + entropy, feed-forward
SquashedNormal, recurrent discrete, recurrent continuous minibatched -- and
(round 3) CartPole, the only built-in path through a three-way head on a
five-wide observation (the general, non-rank-one tower kernels), whose rollout is
also compared with the reference's buffer here (no separate trace exists for it).
"""

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from rl8_amd import AlgorithmConfig, RecurrentAlgorithmConfig, hip  # noqa: E402
from rl8_amd import _utils as host_utils  # noqa: E402
from rl8_amd.distributions import SquashedNormal  # noqa: E402
from rl8_amd.env import ContinuousDummyEnv, DiscreteDummyEnv  # noqa: E402
from rl8_amd.envs.cartpole import CartPole  # noqa: E402

from ._envs import walk_env  # noqa: E402
from .test_algorithm_gpu import compare_collect, inject  # noqa: E402

NUM_ENVS, HORIZON = 64, 32

#: (first-update fixture, trace with the inputs, env, traced config, recurrent)
VARIANTS = {
    "ff_discrete": ("trace_ff_discrete.npz", DiscreteDummyEnv, {}, False),
    "ff_discrete_minibatch": (
        "trace_ff_discrete_minibatch.npz", DiscreteDummyEnv,
        dict(sgd_minibatch_size=256, entropy_coeff=1e-2, dual_clip_param=5.0, horizons_per_env_reset=2), False),
    "ff_continuous_squashed": (
        "trace_ff_continuous_squashed.npz", ContinuousDummyEnv, dict(distribution_cls=SquashedNormal), False),
    "ff_continuous_normal": ("trace_ff_continuous_normal.npz", ContinuousDummyEnv, dict(entropy_coeff=1e-2), False),
    "rec_discrete": ("trace_rec_discrete.npz", DiscreteDummyEnv, {}, True),
    "rec_continuous_minibatch": (
        "trace_rec_continuous_minibatch.npz", ContinuousDummyEnv,
        dict(sgd_minibatch_size=128, entropy_coeff=1e-2, seq_len=8, seqs_per_state_reset=2,
             horizons_per_env_reset=2), True),
    # config 3: the reference's CartPole (examples/cartpole/env.py) through collect() / step();
    # the fixture is self-contained (initial weights, reset state, noise, rollout, updates, first gradient)
    "ff_cartpole": ("first_update_ff_cartpole.npz", CartPole, {}, False),
    # round 5: widths outside the built-in environments' -- a 4-wide observation and a 4-way head
    # (tests/_envs.py; the reference ran the same arithmetic through its DefaultDiscreteModel) on the plane kernels
    "ff_walk4": ("first_update_ff_walk4.npz", walk_env(4, 4), {}, False),
}

STAT_KEYS = ("coefficients/entropy", "coefficients/vf", "losses/entropy", "losses/policy", "losses/vf",
             "losses/total", "monitors/kl_div")

#: north_star: fp32 losses within 1e-5 (relative). Absolute floors only where the
#: reference's own number is rounding noise: with normalised advantages and
#: ratio == 1 the full-batch policy loss is -mean(adv) ~ 1e-8 (a sum of 2048 O(1)
#: fp32 terms that cancel), and the first KL is exactly 0 in the reference because
#: it evaluates the same module twice.
REL = 1e-5
ABS = {"losses/policy": 1e-6, "losses/total": 1e-6, "monitors/kl_div": 1e-7}


class Recorder:
    """Every ``StatTracker.update`` of a ``step()`` and the gradient at its first
    ``optimizer.step()`` (the reference-side recorder of the fixtures, mirrored)."""

    def __init__(self, algo):
        self.algo, self.updates, self.first_grads = algo, [], None

    def __enter__(self):
        rec = self
        self._update, self._opt_step = host_utils.StatTracker.update, self.algo.optimizer.step

        def update(tracker, data, /, *, reduce=False):
            rec.updates.append([float(data[k]) for k in STAT_KEYS] + [float(reduce)])
            return rec._update(tracker, data, reduce=reduce)

        def opt_step(*args, **kwargs):
            if rec.first_grads is None:
                rec.first_grads = {k: p.grad.detach().clone() for k, p in rec.algo.policy.model.named_parameters()
                                   if p.grad is not None}
            return rec._opt_step(*args, **kwargs)

        host_utils.StatTracker.update = update
        self.algo.optimizer.step = opt_step
        return self

    def __exit__(self, *exc):
        host_utils.StatTracker.update = self._update
        self.algo.optimizer.step = self._opt_step


def build(golden, variant, **overrides):
    trace_name, env_cls, config, recurrent = VARIANTS[variant]
    trace = golden(trace_name)
    cfg_cls = RecurrentAlgorithmConfig if recurrent else AlgorithmConfig
    algo = cfg_cls(num_envs=NUM_ENVS, horizon=HORIZON, **{**config, **overrides}).build(env_cls)
    algo.policy.model.load_state_dict(
        {k[len("init_"):]: torch.from_numpy(trace[k]) for k in trace if k.startswith("init_")})
    inject(algo, trace, 0)
    return algo, trace


def assert_update(got, want, label):
    for i, k in enumerate(STAT_KEYS):
        assert got[i] == pytest.approx(want[i], rel=REL, abs=ABS.get(k, 1e-9)), (label, k, got[i], want[i])
    assert got[-1] == want[-1], (label, "reduce flag")


@pytest.mark.parametrize("variant", list(VARIANTS))
def test_one_sgd_iteration_matches_reference_to_1e5(golden, variant):
    """``num_sgd_iters=1`` over one full-buffer minibatch: StepStats, the clipped
    gradient of the optimizer step and the weights after it."""
    check_one_sgd_iteration(golden, variant)


def check_one_sgd_iteration(golden, variant, towers="matrix"):
    g = golden(f"first_update_{variant}.npz")
    algo, _ = build(golden, variant, num_sgd_iters=1, sgd_minibatch_size=None)
    algo.collect()
    hip.timer.reset()
    hip.timer.enabled = True
    try:
        with Recorder(algo) as rec:
            stats = algo.step()
        launched = set(hip.timer.summary())
    finally:
        hip.timer.enabled = False
    # which tower kernels these reference-held numbers pin: a two-action policy runs the gate-mode (rank-one)
    # backward kernels from its FIRST update (Algorithm's pair hint), like every value tower; CartPole's three-way
    # head and the continuous (mean | log_std) heads the general ones
    if towers == "piecewise":  # (tests/test_piecewise_gpu.py: the opt-in tables of a scalar observation)
        assert {"pw_tower_forward", "pw_segment_sums"} <= launched and not any(k.startswith("mlp_") for k in launched), launched
    elif variant.startswith("ff_"):
        assert {"mlp_tower_backward_gate", "mlp_wgrad_gate"} <= launched, launched
        general = {"mlp_tower_backward", "mlp_wgrad"} <= launched
        assert general == (variant not in ("ff_discrete", "ff_discrete_minibatch")), (variant, launched)
    assert len(rec.updates) == 1
    assert_update(rec.updates[0], g["sgd1_updates"][0], variant)
    for k, w in zip(g["step_stat_keys"], g["sgd1_step_stats"]):
        assert stats[str(k)] == pytest.approx(w, rel=REL, abs=ABS.get(str(k), 1e-9)), (variant, k)
    # Gradient of the whole model as the optimizer saw it (after clipping to
    # max_grad_norm): error relative to the gradient's norm, and per tensor
    # relative to that tensor's largest entry.
    want = {k[len("sgd1_grad_"):]: g[k] for k in g if k.startswith("sgd1_grad_")}
    assert set(want) == set(rec.first_grads)
    err_sq = ref_sq = 0.0
    for k, w in want.items():
        got = rec.first_grads[k].double().cpu().numpy()
        err_sq += float(((got - w) ** 2).sum())
        ref_sq += float((w.astype(np.float64) ** 2).sum())
        np.testing.assert_allclose(got, w, rtol=0, atol=2e-5 * float(np.abs(w).max()) + 1e-9, err_msg=f"{variant} {k}")
    assert (err_sq / ref_sq) ** 0.5 < 1e-5, (variant, (err_sq / ref_sq) ** 0.5)
    assert ref_sq ** 0.5 == pytest.approx(float(g["sgd1_clipped_grad_norm"]), rel=1e-6)
    # One Adam step moves a weight by lr * g / (|g| + eps'): ~ +-1e-3 whatever the
    # gradient's size, so only entries whose gradient is not itself rounding noise
    # are comparable, and those must land on the reference's weights.
    lr = 1e-3
    for k, p in algo.policy.model.named_parameters():
        w, gw = g[f"sgd1_final_{k}"], want[k]
        solid = np.abs(gw) > 1e-3 * np.abs(gw).max()
        np.testing.assert_allclose(p.detach().cpu().numpy()[solid], w[solid], rtol=0, atol=0.02 * lr,
                                   err_msg=f"{variant} weights {k}")


@pytest.mark.parametrize("variant", list(VARIANTS))
def test_first_minibatch_of_traced_config_matches_reference_to_1e5(golden, variant):
    """The traced configs themselves (4 SGD iterations, minibatched where the
    trace is): the first per-minibatch update at 1e-5; the later ones, taken after
    1..31 Adam steps on differently-ordered GEMMs, within the drift band of
    test_algorithm_gpu.run_trace."""
    g = golden(f"first_update_{variant}.npz")
    algo, _ = build(golden, variant)
    algo.collect()
    with Recorder(algo) as rec:
        algo.step()
    want = g["traced_updates"]
    assert len(rec.updates) == len(want)
    assert_update(rec.updates[0], want[0], variant)
    got = np.array(rec.updates)
    assert np.array_equal(got[:, -1], want[:, -1])
    np.testing.assert_allclose(got[1:, :-1], want[1:, :-1], rtol=2e-2, atol=2e-3)


def compare_recurrent_collect(algo, g, it, *, discrete):
    """Rollout of a recurrent algorithm against the reference's buffer snapshot (actions bit-exact when discrete,
    floats at the bars of test_algorithm_gpu.run_recurrent_trace with weights identical to the reference's)."""
    from rl8_amd.data import DataKeys

    buf = algo.buffer
    got_actions = buf[DataKeys.ACTIONS][:, :HORIZON].cpu().numpy()
    if discrete:
        assert np.array_equal(got_actions, g[f"it{it}_collect_actions"][:, :HORIZON])
    else:
        np.testing.assert_allclose(got_actions, g[f"it{it}_collect_actions"][:, :HORIZON], rtol=1e-4, atol=1e-4)
    for key in ("obs", "rewards", "reversed_discounted_returns"):
        np.testing.assert_allclose(buf[key].cpu().numpy(), g[f"it{it}_collect_{key}"], rtol=1e-5, atol=2e-4, err_msg=key)
    np.testing.assert_allclose(buf[DataKeys.VALUES].cpu().numpy(), g[f"it{it}_collect_values"], rtol=1e-4, atol=5e-5)
    for sk in ("hidden_states", "cell_states"):
        leaf = buf[DataKeys.STATES][sk].cpu().numpy()
        np.testing.assert_allclose(leaf[:, -1], g[f"it{it}_collect_states_{sk}_last"], rtol=1e-4, atol=2e-5)
        np.testing.assert_allclose(leaf[:, 6], g[f"it{it}_collect_states_{sk}_col6"], rtol=1e-4, atol=2e-5)


def teacher_force(algo, g):
    """Put the algorithm where the REFERENCE stood after its iteration 0: weights, environment state, and the
    columns collect() carries over (last observation, reversed discounted return, recurrent states:
    src/rl8/algorithms/_feedforward.py:336-357, _recurrent.py:380-392)."""
    from rl8_amd.data import DataKeys

    dev = algo.policy.device
    model = algo.policy.model
    model.load_state_dict({k: torch.from_numpy(g[f"it0_final_{k}"]) for k in model.state_dict()})
    algo.buffer[DataKeys.OBS][:, -1].copy_(torch.from_numpy(g["it0_final_obs"]).to(dev))
    if "it0_final_rdr" in g:
        algo.buffer[DataKeys.REVERSED_DISCOUNTED_RETURNS][:, -1].copy_(torch.from_numpy(g["it0_final_rdr"]).to(dev))
    if "it0_env_state" in g:
        algo.env.state.copy_(torch.from_numpy(g["it0_env_state"]).to(dev))
    else:
        algo.env.state.copy_(torch.from_numpy(g["it0_final_obs"]).to(dev))  # the dummy envs' observation is their state
    if DataKeys.STATES in algo.buffer.keys():
        for sk in ("hidden_states", "cell_states"):
            last = torch.from_numpy(g[f"it0_collect_states_{sk}_last"]).to(dev)
            algo.buffer[DataKeys.STATES][sk][:, -1].copy_(last)


def assert_first_update_and_gradient(rec, g2, label):
    want = g2["it1_updates"]
    assert len(rec.updates) == len(want)
    assert_update(rec.updates[0], want[0], label)
    grads = {k[len("it1_first_grad_"):]: g2[k] for k in g2 if k.startswith("it1_first_grad_")}
    assert set(grads) == set(rec.first_grads)
    err_sq = ref_sq = 0.0
    for k, w in grads.items():
        got = rec.first_grads[k].double().cpu().numpy()
        err_sq += float(((got - w) ** 2).sum())
        ref_sq += float((w.astype(np.float64) ** 2).sum())
        np.testing.assert_allclose(got, w, rtol=0, atol=2e-5 * float(np.abs(w).max()) + 1e-9, err_msg=f"{label} {k}")
    assert (err_sq / ref_sq) ** 0.5 < 1e-5, (label, (err_sq / ref_sq) ** 0.5)
    assert ref_sq ** 0.5 == pytest.approx(float(g2["it1_first_clipped_grad_norm"]), rel=1e-6)


@pytest.mark.parametrize("variant", ["rec_discrete", "rec_continuous_minibatch"])
def test_second_iteration_recurrent_teacher_forced_matches_reference_to_1e5(golden, variant):
    """The recurrent twins of the test below (tests/golden/second_iteration_rec_*.npz): iteration 1 from the
    reference's weights after iteration 0, its first update and first gradient at the first-update bar."""
    g2 = golden(f"second_iteration_{variant}.npz")
    algo, trace = build(golden, variant)
    algo.collect()
    algo.step()
    teacher_force(algo, trace)
    inject(algo, trace, 1)
    algo.collect()
    compare_recurrent_collect(algo, trace, 1, discrete=variant == "rec_discrete")
    assert algo.state.reward_scale == pytest.approx(float(trace["it1_reward_scale"]), rel=1e-5)
    with Recorder(algo) as rec:
        algo.step()
    assert_first_update_and_gradient(rec, g2, variant)


CARRIED = {
    # recurrent states, observation and returns carried through a second collect() (nothing re-initialised)
    "rec_carry": (DiscreteDummyEnv, dict(seq_len=4, seqs_per_state_reset=16, horizons_per_env_reset=2), True),
    # CartPole's [4, N] physics state carried through a second collect() on a used buffer
    "ff_cartpole": (CartPole, dict(horizons_per_env_reset=2), False),
}


@pytest.mark.parametrize("variant", list(CARRIED))
def test_second_iteration_on_carried_state_matches_reference_to_1e5(golden, variant):
    """Self-contained two-iteration fixtures (generate_fixtures.gen_two_iterations): iteration 0 on the
    reference's inputs, then iteration 1 teacher-forced from the reference's weights / env state / carried
    columns -- a rollout that starts from carried LSTM states (src/rl8/algorithms/_recurrent.py:380-392, no
    re-initialisation inside it) or from CartPole's carried state (examples/cartpole/env.py:138-150), its
    statistics, the first StatTracker.update and the first gradient at 1e-5."""
    from rl8_amd.data import DataKeys

    env_cls, config, recurrent = CARRIED[variant]
    g = golden(f"second_iteration_{variant}.npz")
    cfg_cls = RecurrentAlgorithmConfig if recurrent else AlgorithmConfig
    algo = cfg_cls(num_envs=NUM_ENVS, horizon=HORIZON, **config).build(env_cls)
    algo.policy.model.load_state_dict({k[len("init_"):]: torch.from_numpy(g[k]) for k in g if k.startswith("init_")})

    def check_collect(it):
        buf = algo.buffer
        if recurrent:
            compare_recurrent_collect(algo, g, it, discrete=True)
        else:
            assert np.array_equal(buf[DataKeys.ACTIONS][:, :HORIZON].cpu().numpy(), g[f"it{it}_collect_actions"][:, :HORIZON])
            for key in ("obs", "rewards", "reversed_discounted_returns"):
                np.testing.assert_allclose(buf[key].cpu().numpy(), g[f"it{it}_collect_{key}"], rtol=2e-6, atol=2e-6,
                                           err_msg=key)
            np.testing.assert_allclose(buf[DataKeys.LOGP].cpu().numpy()[:, :HORIZON], g[f"it{it}_collect_logp"][:, :HORIZON],
                                       rtol=1e-5, atol=2e-6)
            np.testing.assert_allclose(buf[DataKeys.VALUES].cpu().numpy(), g[f"it{it}_collect_values"], rtol=1e-5, atol=2e-6)

    inject(algo, g, 0)
    stats = algo.collect()
    assert stats["env/resets"] == NUM_ENVS
    check_collect(0)
    algo.step()
    teacher_force(algo, g)
    inject(algo, g, 1)
    stats = algo.collect()
    assert stats["env/resets"] == 0  # nothing was reset: the whole rollout runs on carried state
    check_collect(1)
    for k, w in zip(g["collect_stat_keys"], g["it1_collect_stats"]):
        assert stats[str(k)] == pytest.approx(w, rel=1e-5, abs=1e-5), k
    assert algo.state.reward_scale == pytest.approx(float(g["it1_reward_scale"]), rel=1e-5)
    with Recorder(algo) as rec:
        algo.step()
    assert_first_update_and_gradient(rec, g, variant)


@pytest.mark.parametrize("variant", [v for v in VARIANTS if v.startswith("ff_") and v not in ("ff_cartpole", "ff_walk4")])
def test_second_iteration_teacher_forced_matches_reference_to_1e5(golden, variant):
    """Iteration 1 of the traced configs from the REFERENCE's weights after iteration 0
    (``it0_final_*`` of the trace) instead of this build's drifted ones: a second rollout
    (carried or re-drawn states, reward scale of a used buffer) and the first per-minibatch
    update + gradient of a ``step()`` whose optimizer has history, held to the first-update
    bar (tests/golden/second_iteration_*.npz). The free-running comparison of the same
    iteration keeps its wide band in test_algorithm_gpu.run_trace."""
    from rl8_amd.data import DataKeys

    g2 = golden(f"second_iteration_{variant}.npz")
    algo, trace = build(golden, variant)
    algo.collect()
    algo.step()
    model = algo.policy.model
    model.load_state_dict({k: torch.from_numpy(trace[f"it0_final_{k}"]) for k in model.state_dict()})
    final_obs = torch.from_numpy(trace["it0_final_obs"]).to(algo.policy.device)
    algo.buffer[DataKeys.OBS][:, -1].copy_(final_obs)  # carried when horizons_per_env_reset > 1
    algo.env.state.copy_(final_obs)  # the dummy envs' observation is their state
    inject(algo, trace, 1)
    algo.collect()
    compare_collect(algo, trace, 1, discrete=variant.startswith("ff_discrete"), loose=1.0)
    assert algo.state.reward_scale == pytest.approx(float(trace["it1_reward_scale"]), rel=1e-5)
    with Recorder(algo) as rec:
        algo.step()
    assert_first_update_and_gradient(rec, g2, variant)


def test_kl_early_stop_matches_reference(golden):
    """``target_kl_div`` (reference ``_feedforward.py:577-582``): same number of
    updates, same stats, every ``.grad`` None afterwards, same weights as the
    reference's run on the same rollout and permutations."""
    g = golden("early_stop.npz")
    with pytest.raises(ValueError, match="not compatible with gradient"):
        AlgorithmConfig(num_envs=NUM_ENVS, horizon=HORIZON, accumulate_grads=True, sgd_minibatch_size=512,
                        target_kl_div=0.1).build(DiscreteDummyEnv)
    algo, _ = build(golden, "ff_discrete", sgd_minibatch_size=512, target_kl_div=float(g["target_kl_div"]))
    algo.collect()
    algo.injected_permutations = [torch.from_numpy(p) for p in g["perms"]]
    with Recorder(algo) as rec:
        stats = algo.step()
    want = g["updates"]
    assert len(rec.updates) == len(want) == int(g["stopped_at_update"]) + 1
    assert_update(rec.updates[0], want[0], "early stop")
    np.testing.assert_allclose(np.array(rec.updates)[1:, :-1], want[1:, :-1], rtol=2e-3, atol=1e-5)
    for k, w in zip(g["step_stat_keys"], g["step_stats"]):
        assert stats[str(k)] == pytest.approx(w, rel=2e-3, abs=1e-5), k
    assert bool(g["grads_are_none_after_stop"])
    assert all(p.grad is None for p in algo.policy.model.parameters())
    for k, p in algo.policy.model.named_parameters():
        np.testing.assert_allclose(p.detach().cpu().numpy(), g[f"final_{k}"], rtol=0, atol=2e-4, err_msg=k)


@pytest.mark.parametrize("variant", ["ff_cartpole", "ff_walk4"])
def test_env_rollout_matches_reference(golden, variant):
    """collect() on the reference's rollout inputs (reset state, initial weights, multinomial noise) for the
    self-contained fixtures -- CartPole, and the 4-wide / 4-way walk of tests/_envs.py: action indices bit-exact,
    physics 1e-6 absolute per step (obs, rewards; one sin / cos per step differs by an ulp between devices),
    log-probabilities / values / CollectStats at 1e-5."""
    from rl8_amd.data import DataKeys

    g = golden(f"first_update_{variant}.npz")
    algo, _ = build(golden, variant)
    stats = algo.collect()
    buf = algo.buffer
    assert np.array_equal(buf[DataKeys.ACTIONS][:, :HORIZON].cpu().numpy(), g["it0_collect_actions"][:, :HORIZON])
    for key in ("obs", "rewards", "reversed_discounted_returns"):
        np.testing.assert_allclose(buf[key].cpu().numpy(), g[f"it0_collect_{key}"], rtol=2e-6, atol=2e-6, err_msg=key)
    np.testing.assert_allclose(buf[DataKeys.LOGP].cpu().numpy()[:, :HORIZON], g["it0_collect_logp"][:, :HORIZON],
                               rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(buf[DataKeys.VALUES].cpu().numpy(), g["it0_collect_values"], rtol=1e-5, atol=2e-6)
    for k, w in zip(g["collect_stat_keys"], g["it0_collect_stats"]):
        assert stats[str(k)] == pytest.approx(w, rel=1e-5, abs=1e-5), k
    assert algo.state.reward_scale == pytest.approx(float(g["it0_reward_scale"]), rel=1e-5)
