"""CPU-only tests: the host-side mirror of the reference interface (containers,
specs, hparams validation, schedulers, batcher, stat tracker), the C-ABI library
(loads, exports every symbol include/rl8_amd.h declares), and the "no CPU
fallback" contract. No kernel is launched here."""

import os
import re

import numpy as np
import pytest
import torch

from rl8_amd import hip
from rl8_amd._utils import Batcher, CumulativeAverage, StatTracker, reduce_stats
from rl8_amd.data import AlgorithmHparams, DataKeys, RecurrentAlgorithmHparams
from rl8_amd.schedulers import EntropyScheduler, LRScheduler
from rl8_amd.specs import Categorical, Composite, Unbounded
from rl8_amd.tensordict import TensorDict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# --- C ABI ------------------------------------------------------------------
def declared_symbols():
    text = open(os.path.join(ROOT, "include", "rl8_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rl8_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = hip.load()
    names = declared_symbols()
    assert len(names) >= 17
    for name in names:
        assert hasattr(lib, name), f"librl8_amd.so lacks {name}"
    assert sorted(hip.SIGNATURES) == names, "rl8_amd/hip.py bindings out of sync with include/rl8_amd.h"
    assert hip.abi_version() == (hip.ABI_VERSION, "gfx950") and hip.ABI_VERSION == 106
    assert lib.rl8_scratch_bytes() >= 2048 * 16 * 8


def test_no_torch_types_in_abi_header():
    text = open(os.path.join(ROOT, "include", "rl8_amd.h")).read()
    code = re.sub(r"/\*.*?\*/", "", text, flags=re.S)  # comments may cite torch
    assert "torch" not in code.lower() and "at::" not in code and "c10::" not in code
    assert "#include <torch" not in text and "#include <ATen" not in text


def test_argument_checks_happen_before_any_launch():
    lib = hip.load()
    # NULL pointers / bad sizes are rejected in the C layer with negative codes
    assert lib.rl8_dummy_env_step_f32(None, None, 1, None, 8, None) == -1
    assert lib.rl8_gae_scan_f32(None, None, None, None, 8, 4, 1, 0.9, 0.9, 1.0, 0, None, None, None) == -1
    assert lib.rl8_gather_minibatch(None, 1, 1, None, 1, None) == -1


def test_cpu_tensors_are_refused_loudly():
    t = torch.zeros(8, 1)
    with pytest.raises(hip.HipExtensionError, match="no CPU fallback"):
        hip.dummy_env_step(t, torch.zeros(8, 1, dtype=torch.int64), torch.zeros(8, 1))
    from rl8_amd.nn import generalized_advantage_estimate

    batch = TensorDict({"rewards": torch.ones(4, 3, 1), "values": torch.ones(4, 3, 1)}, batch_size=[4, 3])
    with pytest.raises(hip.HipExtensionError):
        generalized_advantage_estimate(batch)
    if not torch.cuda.is_available():
        from rl8_amd import AlgorithmConfig
        from rl8_amd.env import DiscreteDummyEnv

        with pytest.raises(hip.HipExtensionError, match="no CPU path"):
            AlgorithmConfig(num_envs=8, horizon=4, device="cpu").build(DiscreteDummyEnv)


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "rl8_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), f
                assert "librl8_oracle" not in src, f


# --- containers ---------------------------------------------------------------
def test_tensordict_indexing_views_and_assignment():
    td = TensorDict({"a": torch.arange(24.0).reshape(4, 3, 2), "b": torch.zeros(4, 3, 1)}, batch_size=[4, 3])
    assert td.batch_size == (4, 3) and td.size(1) == 3 and set(td.keys()) == {"a", "b"}
    sub = td[:, :2, ...]
    assert sub.batch_size == (4, 2) and sub["a"].shape == (4, 2, 2)
    sub["b"][...] = 7.0
    assert float(td["b"][:, :2].sum()) == 56.0  # views, not copies
    last = td[:, -1, ...]
    assert last.batch_size == (4,)
    idx = torch.tensor([3, 0])
    picked = td.reshape(-1)[idx, ...]
    assert picked.batch_size == (2,) and torch.equal(picked["a"], td["a"].reshape(12, 2)[idx])
    td[:, 0, ...] = TensorDict({"b": torch.ones(4, 1)}, batch_size=[4])
    assert float(td["b"][:, 0].sum()) == 4.0
    del td["b"]
    assert "b" not in td.keys()
    nested = TensorDict({"s": {"h": torch.zeros(4, 3, 5)}}, batch_size=[4, 3])
    assert nested[("s", "h")].shape == (4, 3, 5) and nested[:, 1]["s"].batch_size == (4,)
    halves = td.apply(lambda x: x / 2)
    assert torch.equal(halves["a"], td["a"] / 2)
    with pytest.raises(RuntimeError):
        td["bad"] = torch.zeros(5, 3)


def test_specs_zero_rand_and_membership():
    obs = Unbounded(3)
    act = Categorical(4, shape=torch.Size([2]))
    assert obs.shape == (3,) and obs.ndim == 1 and act.space.n == 4 and act.dtype == torch.int64
    spec = Composite({DataKeys.OBS: obs, DataKeys.ACTIONS: act})
    buf = spec.zero([5, 7])
    assert buf.batch_size == (5, 7) and buf[DataKeys.OBS].shape == (5, 7, 3) and buf[DataKeys.ACTIONS].shape == (5, 7, 2)
    act.assert_is_in(act.rand([9]))
    with pytest.raises(AssertionError):
        act.assert_is_in(torch.full((9, 2), 4))
    with pytest.raises(AssertionError):
        obs.assert_is_in(torch.zeros(9, 2))
    spec.set("extra", Unbounded(1))
    assert [k for k in spec] == [DataKeys.OBS, DataKeys.ACTIONS, "extra"]


# --- hparams (reference src/rl8/data.py:196-270) -------------------------------
def hparams(**kw):
    base = dict(
        accumulate_grads=False, clip_param=0.2, device="cuda", dual_clip_param=None, enable_amp=False,
        gae_lambda=0.95, gamma=0.95, horizon=32, horizons_per_env_reset=1, max_grad_norm=5.0,
        normalize_advantages=True, normalize_rewards=True, num_envs=64, num_sgd_iters=4,
        sgd_minibatch_size=2048, shuffle_minibatches=True, target_kl_div=None, vf_clip_param=5.0, vf_coeff=1.0,
    )
    base.update(kw)
    return AlgorithmHparams(**base)


@pytest.mark.parametrize("kw,msg", [
    (dict(clip_param=1.0), "clip_param"),
    (dict(dual_clip_param=1.0), "dual_clip_param"),
    (dict(device="cpu", enable_amp=True), "enable_amp"),
    (dict(gae_lambda=0.0), "gae_lambda"),
    (dict(gamma=1.5), "gamma"),
    (dict(horizon=0), "horizon"),
    (dict(horizons_per_env_reset=0), "horizons_per_env_reset"),
    (dict(max_grad_norm=0.0), "max_grad_norm"),
    (dict(num_sgd_iters=0), "num_sgd_iters"),
    (dict(sgd_minibatch_size=0), "sgd_minibatch_size"),
    (dict(target_kl_div=0.1, accumulate_grads=True, sgd_minibatch_size=64), "target_kl_div"),
    (dict(target_kl_div=0.1, enable_amp=True), "target_kl_div"),
    (dict(target_kl_div=-1.0), "target_kl_div"),
    (dict(vf_clip_param=0.0), "vf_clip_param"),
    (dict(vf_coeff=0.0), "vf_coeff"),
    (dict(accumulate_grads=True), "accumulate_grads"),
])
def test_hparam_validation_errors(kw, msg):
    with pytest.raises(ValueError, match=msg):
        hparams(**kw)


def test_hparam_derived_values():
    hp = hparams(sgd_minibatch_size=256).validate()
    assert hp.num_minibatches == 8 and hp.device_type == "cuda"
    with pytest.raises(ValueError, match="factor"):
        hparams(sgd_minibatch_size=100).validate()
    rhp = RecurrentAlgorithmHparams(**{**hp.__dict__, "seq_len": 4, "seqs_per_state_reset": 8, "sgd_minibatch_size": 512})
    assert rhp.num_minibatches == 1
    with pytest.raises(ValueError, match="seq_len"):
        RecurrentAlgorithmHparams(**{**hp.__dict__, "seq_len": 5, "seqs_per_state_reset": 8})


# --- schedulers (reference tests/test_schedulers.py) ---------------------------
def test_schedulers_match_reference_cases():
    e = EntropyScheduler(0.0, schedule=[[0.0, 1.0], [1.0, 2.0]], kind="interp")
    assert (e.step(0.0), e.step(0.5), e.step(1.0)) == (1.0, 1.5, 2.0)
    e = EntropyScheduler(0.0, schedule=[[0.0, 1.0], [1.0, 2.0]], kind="step")
    assert (e.step(0.0), e.step(2.0), e.step(3.0)) == (1.0, 2.0, 2.0)
    opt = torch.optim.Adam([torch.nn.Parameter(torch.tensor([0.0]))])
    lr = LRScheduler(opt, schedule=[[0.0, 1.0], [1.0, 2.0]], kind="interp")
    assert (lr.step(0.0), lr.step(0.5), lr.step(1.0)) == (1.0, 1.5, 2.0)
    assert opt.param_groups[0]["lr"] == 2.0
    lr = LRScheduler(opt, schedule=[[0.0, 1.0], [1.0, 2.0]], kind="step")
    assert (lr.step(0.0), lr.step(2.0), lr.step(3.0)) == (1.0, 2.0, 2.0)
    const = LRScheduler(torch.optim.Adam([torch.nn.Parameter(torch.tensor([0.0]))], lr=3e-4))
    const.step(10_000)
    assert const.optimizer.param_groups[0]["lr"] == 3e-4 and const.coeff == 0.0
    assert EntropyScheduler(0.25).step(99) == 0.25
    with pytest.raises(ValueError, match="first"):
        EntropyScheduler(0.0, schedule=[[5, 1.0]])
    with pytest.raises(ValueError, match="kinds"):
        EntropyScheduler(0.0, schedule=[[0, 1.0]], kind="cosine")


# --- batcher / stat tracker ----------------------------------------------------
def test_batcher_chunks_and_reshuffles_every_iteration():
    td = TensorDict({"x": torch.arange(12.0).unsqueeze(-1)}, batch_size=[12])
    chunks = [b["x"].flatten().tolist() for b in Batcher(td, batch_size=5)]
    assert chunks == [[0, 1, 2, 3, 4], [5, 6, 7, 8, 9], [10, 11]]
    torch.manual_seed(0)
    b = Batcher(td, batch_size=12, shuffle=True)
    first = next(iter(b))["x"].flatten().tolist()
    second = next(iter(b))["x"].flatten().tolist()
    assert sorted(first) == list(range(12)) and first != second
    injected = Batcher(td, batch_size=6, shuffle=True, permutation_fn=lambda n: torch.arange(n).flip(0))
    assert next(iter(injected))["x"].flatten().tolist() == [11, 10, 9, 8, 7, 6]


def test_stat_tracker_sums_then_averages():
    ca = CumulativeAverage()
    assert ca.update(0.0) == 0.0 and ca.update(2.0) == 1.0
    st = StatTracker(["a", "c"], sum_keys=["a"])
    st.update({"a": 1.0, "c": 10.0})
    st.update({"a": 2.0, "c": 20.0}, reduce=True)
    st.update({"a": 5.0, "c": 30.0}, reduce=True)
    assert st.items() == {"a": 4.0, "c": 20.0}
    assert reduce_stats({"r/min": [1, -2], "r/max": [1, 3], "r/mean": [1, 3], "r/std": [3, 4], "n": [1, 2]}) == {
        "r/min": -2, "r/max": 3, "r/mean": 2.0, "r/std": (12.5) ** 0.5, "n": 3}


def test_buffer_layout_detection():
    env_major = torch.zeros(6, 4, 1)
    time_major = torch.zeros(4, 6, 1).transpose(0, 1)
    assert hip.buffer_layout(env_major)[0] == hip.LAYOUT_ENV_MAJOR
    assert hip.buffer_layout(time_major)[0] == hip.LAYOUT_TIME_MAJOR
    assert hip.buffer_layout(torch.zeros(6, 8, 1)[:, ::2])[0] == -1


def test_collect_stats_from_raw_moments():
    from rl8_amd.algorithms._feedforward import _collect_stats_from_raw

    rng = np.random.default_rng(0)
    rewards = rng.standard_normal((50, 9)).astype(np.float64)
    rdr = rng.standard_normal((50, 10)).astype(np.float64)
    ret = rewards.sum(1)
    raw = [50, ret.sum(), (ret**2).sum(), ret.min(), ret.max(), 450, rewards.sum(), (rewards**2).sum(),
           rewards.min(), rewards.max(), rdr[:, 1:].sum(), (rdr[:, 1:] ** 2).sum()]
    stats, scale = _collect_stats_from_raw(raw)
    assert stats["returns/std"] == pytest.approx(ret.std(ddof=1))
    assert stats["rewards/std"] == pytest.approx(rewards.std(ddof=1))
    assert stats["rewards/mean"] == pytest.approx(rewards.mean())
    assert scale == pytest.approx(rdr[:, 1:].std(ddof=1))


def test_column_sums_fold_equals_the_plain_sum():
    """hip._column_sums (the head bias gradient: torch's own reduction of a [2^23, 3] tensor over its long dimension ran at
    24 GB/s) folds [m, n] to [m / 1024, 1024 n] first; same sums up to fp32 reassociation, any m, n, tail or not."""
    import torch

    from rl8_amd import hip

    g = torch.Generator().manual_seed(5)
    for m, n in ((10, 3), (65_536, 3), (65_536 + 777, 2), (200_003, 3), (70_000, 1)):
        t = torch.randn(m, n, generator=g)
        got, want = hip._column_sums(t), t.double().sum(0)
        assert got.shape == (n,)
        assert float((got.double() - want).abs().max()) <= 1e-5 * float(t.abs().sum(0).max())
    assert torch.equal(hip._column_sums(torch.ones(131_072, 3)), torch.full((3,), 131_072.0))
