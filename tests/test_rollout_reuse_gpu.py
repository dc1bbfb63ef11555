"""The first SGD pass starts from the rollout's own tower evaluations (fused_mlp.RolloutRecord) instead of running
the forward kernels again on the same rows with the same weights (reference: algorithms/_feedforward.py:362-367
evaluates the model during collect(), :519-524 evaluates it again in step()).  The shortcut must be invisible:
identical numbers, bit for bit, with it and without it; and it must switch itself off whenever a weight or an
observation changed between collect() and step()."""

import pytest
import torch

pytestmark = pytest.mark.gpu

from rl8_amd import AlgorithmConfig  # noqa: E402
from rl8_amd.data import DataKeys  # noqa: E402
from rl8_amd.distributions import SquashedNormal  # noqa: E402
from rl8_amd.env import ContinuousDummyEnv, DiscreteDummyEnv  # noqa: E402
from rl8_amd.envs.cartpole import CartPole  # noqa: E402
from rl8_amd.nn import fused_mlp  # noqa: E402

from .test_first_update_gpu import Recorder  # noqa: E402


def run(env_cls, reuse, *, seed=5, before_step=None, rows_per_pass=None, **config):
    torch.manual_seed(seed)
    algo = AlgorithmConfig(num_envs=512, horizon=16, **config).build(env_cls)
    algo.reuse_rollout_forward = reuse
    if rows_per_pass:
        algo.max_rows_per_pass = rows_per_pass
    algo.collect()
    if before_step is not None:
        before_step(algo)
    for k in fused_mlp.replay_stats:
        fused_mlp.replay_stats[k] = 0
    with Recorder(algo) as rec:
        stats = algo.step()
    return algo, rec, stats, dict(fused_mlp.replay_stats)


def assert_identical(a, b):
    (algo_a, rec_a, stats_a, _), (algo_b, rec_b, stats_b, _) = a, b
    assert rec_a.updates == rec_b.updates  # every StatTracker.update of the step, floats compared exactly
    for k in stats_a:
        if not k.startswith("profiling"):
            assert stats_a[k] == stats_b[k], k
    assert set(rec_a.first_grads) == set(rec_b.first_grads)
    for k, g in rec_a.first_grads.items():
        assert torch.equal(g, rec_b.first_grads[k]), k
    for pa, pb in zip(algo_a.policy.model.parameters(), algo_b.policy.model.parameters()):
        assert torch.equal(pa, pb)
    for key in (DataKeys.VALUES, DataKeys.LOGP, DataKeys.ACTIONS):
        assert torch.equal(algo_a.buffer[key], algo_b.buffer[key]), key


CASES = {
    # two-way categorical + value tower: gate bits only (the headline config's towers)
    "discrete": (DiscreteDummyEnv, {}),
    # general heads (mean | log_std): h2 recorded too
    "squashed": (ContinuousDummyEnv, dict(distribution_cls=SquashedNormal)),
    "normal_entropy": (ContinuousDummyEnv, dict(entropy_coeff=1e-2)),
    # d_in = 5, three-way head
    "cartpole": (CartPole, {}),
}


@pytest.mark.parametrize("case", list(CASES))
def test_full_batch_step_is_bit_identical_with_and_without_the_record(case):
    env_cls, config = CASES[case]
    with_record = run(env_cls, True, **config)
    without = run(env_cls, False, **config)
    assert_identical(with_record, without)
    # iteration 0 of 4 took both towers from the record, over every row; nothing was replayed without it
    assert with_record[3]["replayed_towers"] == 2 and with_record[3]["replayed_rows"] == 2 * 512 * 16
    assert without[3]["replayed_towers"] == 0


def test_chunked_passes_replay_their_own_rows():
    with_record = run(DiscreteDummyEnv, True, rows_per_pass=3000)
    without = run(DiscreteDummyEnv, False, rows_per_pass=3000)
    assert_identical(with_record, without)
    assert with_record[3]["replayed_rows"] == 2 * 512 * 16 and with_record[3]["replayed_towers"] == 2 * 3


@pytest.mark.parametrize("accumulate", [False, True])
def test_shuffled_minibatches_gather_the_record_rows(accumulate):
    """8 minibatches: without accumulation only the first minibatch of iteration 0 still sees the rollout's weights;
    with it (one optimizer step per iteration) all eight do."""
    config = dict(sgd_minibatch_size=512 * 16 // 8, accumulate_grads=accumulate)
    with_record = run(DiscreteDummyEnv, True, **config)
    without = run(DiscreteDummyEnv, False, **config)
    assert_identical(with_record, without)
    assert with_record[3]["replayed_towers"] == (16 if accumulate else 2)
    general = run(ContinuousDummyEnv, True, distribution_cls=SquashedNormal, **config)
    assert_identical(general, run(ContinuousDummyEnv, False, distribution_cls=SquashedNormal, **config))
    # (mean | log_std) heads need h2, 1 KiB per row: recorded only where every row is read back
    assert general[3]["replayed_towers"] == (16 if accumulate else 1)


def test_a_changed_weight_or_observation_disables_the_record():
    def nudge_weight(algo):
        with torch.no_grad():
            algo.policy.model.vf_model[0][2].weight.mul_(1.0)  # values unchanged, version counter bumped

    changed = run(DiscreteDummyEnv, True, before_step=nudge_weight)
    assert changed[3]["replayed_towers"] == 1  # the policy tower's record still stands, the value tower's does not
    assert_identical(changed, run(DiscreteDummyEnv, False, before_step=nudge_weight))

    def really_change_weight(algo):
        with torch.no_grad():
            algo.policy.model.feature_model[0][0].weight.add_(0.01)
            algo.policy.model.vf_model[2].bias.add_(0.5)

    changed = run(DiscreteDummyEnv, True, before_step=really_change_weight)
    assert changed[3]["replayed_towers"] == 0
    assert_identical(changed, run(DiscreteDummyEnv, False, before_step=really_change_weight))

    def touch_obs(algo):
        algo.buffer[DataKeys.OBS][:, 3] += 1.0

    changed = run(DiscreteDummyEnv, True, before_step=touch_obs)
    assert changed[3]["replayed_towers"] == 0
    assert_identical(changed, run(DiscreteDummyEnv, False, before_step=touch_obs))


def test_second_collect_records_again_and_the_record_is_reused_across_iterations():
    torch.manual_seed(1)
    algo = AlgorithmConfig(num_envs=256, horizon=8).build(DiscreteDummyEnv)
    for _ in range(3):
        algo.collect()
        record = algo._record
        assert record is not None and record.valid() and len(record.towers) == 2
        slabs = [tr.gate.data_ptr() for tr in record.towers.values()]
        before = fused_mlp.replay_stats["replayed_towers"]
        algo.step()
        assert fused_mlp.replay_stats["replayed_towers"] == before + 2
        assert not record.valid()  # the optimizer has moved the weights
    assert [tr.gate.data_ptr() for tr in algo._record.towers.values()] == slabs  # allocated once


def test_record_is_not_replayed_for_a_tower_evaluated_twice_per_timestep_or_on_other_rows():
    """the record keys on the tower and the row count alone.  A tower evaluated twice inside one timestep's
    recording context (the second call would overwrite the slab rows of t), or recorded on rows that are not the
    time-major observations, must never be replayed -- it is evaluated normally instead."""
    from rl8_amd.nn import fused_mlp as fm

    torch.manual_seed(2)
    algo = AlgorithmConfig(num_envs=128, horizon=4).build(DiscreteDummyEnv)
    model = algo.policy.model
    trunk, head = model.vf_model[:2], model.vf_model[2]
    rec = fm.RolloutRecord(4, 128, keep_general=False)
    rec.begin()
    xs = torch.randn(4, 128, 1, device="cuda")
    with torch.no_grad():
        for t in range(4):
            with rec.at(t):
                a = fm.tower_forward(trunk, [head], xs[t])
                b = fm.tower_forward(trunk, [head], xs[t] + 1.0) if t == 2 else None
            assert torch.equal(a, head(trunk(xs[t]))) or torch.allclose(a, head(trunk(xs[t])), atol=1e-5)
            if b is not None:
                assert torch.allclose(b, head(trunk(xs[t] + 1.0)), atol=1e-5)
    (tr,) = rec.towers.values()
    assert tr.spoiled and not rec.valid()
    # recorded once per timestep, but on rows that are not where the caller will replay from
    rec.begin()
    with torch.no_grad():
        for t in range(4):
            with rec.at(t):
                fm.tower_forward(trunk, [head], xs[t])
    assert rec.valid()
    rec.require_inputs(xs.data_ptr(), 128 * 4)
    assert rec.valid()
    rec.require_inputs(xs.data_ptr() + 16, 128 * 4)
    assert not rec.valid()


def test_general_head_slabs_respect_the_byte_budget_and_are_released_when_memory_is_tight(monkeypatch):
    """the h2 slab of a general head (1 KiB per row) is admitted against an absolute budget as well as a third
    of free memory, a refused tower simply runs its forward again, and the slabs go when the device runs short."""
    from rl8_amd.distributions import Normal
    from rl8_amd.env import ContinuousDummyEnv
    from rl8_amd.nn import fused_mlp as fm

    def steps(algo):
        out = []
        for _ in range(2):
            algo.collect()
            out.append(algo.step())
        return out

    torch.manual_seed(4)
    algo = AlgorithmConfig(num_envs=512, horizon=8, distribution_cls=Normal).build(ContinuousDummyEnv)
    want = steps(algo)
    assert any(tr.h2 is not None for tr in algo._record.towers.values())
    monkeypatch.setattr(fm, "RECORD_H2_BUDGET_BYTES", 1 << 20)  # 1 MiB: the 4 MiB slab does not fit
    torch.manual_seed(4)
    algo2 = AlgorithmConfig(num_envs=512, horizon=8, distribution_cls=Normal).build(ContinuousDummyEnv)
    got = steps(algo2)
    assert all(tr.h2 is None for tr in algo2._record.towers.values()) and algo2._record.refused
    for a, b in zip(want, got):
        for k in ("losses/policy", "losses/vf", "losses/total", "monitors/kl_div"):
            assert a[k] == b[k], k  # the record never changes a number
    # memory pressure at the end of a step(): the slabs are released and not made again
    monkeypatch.setattr(fm, "RECORD_H2_KEEP_FREE_BYTES", 1 << 60)
    algo.collect()
    algo.step()
    assert all(tr.h2 is None for tr in algo._record.towers.values()) and algo._record.refused
    algo.collect()
    assert all(tr.h2 is None for tr in algo._record.towers.values())
    assert all(torch.isfinite(torch.tensor(v)) for k, v in algo.step().items() if k.startswith("losses"))
