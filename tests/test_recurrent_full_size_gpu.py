"""BASELINE configs[4] at its stated size on ONE device: ``RecurrentAlgorithmConfig`` on
DiscreteDummyEnv, num_envs = 2^16, horizon = 256 -- 2 x 17.2 GB of LSTM states in the rollout buffer
(``/root/reference/src/rl8/models/_recurrent.py:285-296``), 2^24 transitions per ``collect()``, 2^22 sequences of 4
steps per SGD pass (``algorithms/_recurrent.py:325-479``, ``:481-652``) -- and its two big kernels at that row count.

The oracle does not finish this size in seconds, so the checks are the size-independent properties the domain offers:
the dummy env's walk (every observation one unit from the previous one, reward = -|state|), the state re-initialisation
cadence (zeros fed every ``seq_len * seqs_per_state_reset`` timesteps, nowhere else), action indices in range, counters,
finite statistics and losses; and, for the kernels, an fp64 model of the recurrences on a subset of the rows that
includes the launch's first and last workgroups (rows are independent: the grid's index arithmetic is what a size
changes).
"""

import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from rl8_amd import RecurrentAlgorithmConfig, hip  # noqa: E402
from rl8_amd.data import DataKeys  # noqa: E402
from rl8_amd.env import DiscreteDummyEnv  # noqa: E402

from .test_lstm_gpu import _backward_through_time_fp64, _rows_backward_inputs  # noqa: E402

DEV = "cuda:0"
N, H = 1 << 16, 256


def _enough_memory(gib: float) -> bool:
    free, _ = torch.cuda.mem_get_info()
    return free >= gib * (1 << 30)


def test_config5_full_size_rollout_and_update():
    if not _enough_memory(120):
        pytest.skip("needs ~100 GiB of device memory")
    torch.manual_seed(3)
    algo = RecurrentAlgorithmConfig(num_envs=N, horizon=H).build(DiscreteDummyEnv)
    hp = algo.hparams
    assert (hp.seq_len, hp.seqs_per_state_reset, hp.sgd_minibatch_size) == (4, 8, N * (H // 4))
    for it in range(2):
        stats = algo.collect()
        assert stats["env/steps"] == N * H and stats["env/resets"] == N
        assert algo.state.horizons == it + 1 and algo.state.seqs == (it + 1) * (H // hp.seq_len)
        buf = algo.buffer
        obs, rewards = buf[DataKeys.OBS], buf[DataKeys.REWARDS]          # [N, H + 1, 1] views of the time-major leaves
        actions = buf[DataKeys.ACTIONS][:, :H]
        assert int(actions.min()) == 0 and int(actions.max()) == 1
        assert 0.4 < float(actions.float().mean()) < 0.6
        # src/rl8/env.py:253-259: state += 2 a - 1 (one fp32 rounding, the same on any device), reward = -|state|
        assert torch.equal(obs[:, 1:], obs[:, :H] + (2 * actions - 1).to(torch.float32))
        assert torch.equal(rewards[:, :H], -obs[:, 1:].abs())
        assert float(obs[:, 0].abs().max()) <= 100.0
        for key in (DataKeys.LOGP, DataKeys.VALUES):
            assert bool(torch.isfinite(buf[key]).all()), key
        assert float(buf[DataKeys.LOGP][:, :H].max()) <= 0.0
        # state cadence: zeros are fed at t = 0, 32, 64, ... (seq_len 4 x seqs_per_state_reset 8) and only there
        hidden = buf[DataKeys.STATES]["hidden_states"]                    # [N, H + 1, 1, 256]
        cell = buf[DataKeys.STATES]["cell_states"]
        every = hp.seq_len * hp.seqs_per_state_reset
        for t in range(0, H + 1, 16):
            col_h, col_c = hidden[:, t], cell[:, t]
            if t % every == 0 and t < H:
                assert not bool(col_h.any()) and not bool(col_c.any()), t
            else:
                assert bool(col_h.any(dim=-1).all()) and bool(torch.isfinite(col_c).all()), t
                assert float(col_h.abs().max()) < 1.0
        for k, v in stats.items():
            assert math.isfinite(v), k
        assert stats["rewards/max"] <= 0.0 and stats["returns/mean"] < 0.0
        if it == 0:
            step = algo.step()
            for k in ("losses/policy", "losses/vf", "losses/total", "monitors/kl_div"):
                assert math.isfinite(step[k]), k
            assert step["monitors/kl_div"] >= 0.0 and step["losses/vf"] > 0.0
            assert all(bool(torch.isfinite(p).all()) for p in algo.policy.model.parameters())


def _subset(b: int) -> torch.Tensor:
    g = torch.Generator(device=DEV).manual_seed(b)
    middle = torch.randint(4096, b - 4096, (8192,), device=DEV, generator=g)
    return torch.cat([torch.arange(4096, device=DEV), middle, torch.arange(b - 4096, b, device=DEV)])


def test_lstm_rows_backward_at_config5_rows():
    """One launch of the backward through time at 2^19 sequences x 4 steps (what one SGD pass of configs[4]'s 8-GPU
    shard -- and one pass chunk of the full size -- hands it), against the fp64 recurrences on 16 384 of the sequences
    incl. the first and last 4 096: per sequence within 2e-6 of its largest gate gradient (the small-size bar)."""
    if not _enough_memory(60):
        pytest.skip("needs ~45 GiB of device memory")
    b, l = 1 << 19, 4
    c0, gates, cs, dhs, w_hh = _rows_backward_inputs(b, l, 55)
    got = hip.lstm_rows_backward(c0, gates, cs, dhs, hip.lstm_rows_backward_pack(w_hh))
    assert bool(torch.isfinite(got).all())
    rows = _subset(b)
    want = _backward_through_time_fp64(c0[rows], gates[rows], cs[rows], dhs[rows], w_hh)
    scale = want.abs().amax(dim=(1, 2, 3), keepdim=True).clamp_min(1e-300)
    assert float(((got[rows].double() - want).abs() / scale).max()) < 2e-6
    # ... and the launch repeats bit for bit
    assert torch.equal(got, hip.lstm_rows_backward(c0, gates, cs, dhs, hip.lstm_rows_backward_pack(w_hh)))


def test_lstm_step_at_config5_rows():
    """The fp16-plane LSTM step at 2^19 rows x 4 timesteps in training mode (gates and cell states saved) against an
    fp64 LSTM on the same subset of rows: outputs and final states as close to fp64 as at the small sizes."""
    if not _enough_memory(60):
        pytest.skip("needs ~45 GiB of device memory")
    b, l, d = 1 << 19, 4, 1
    g = torch.Generator(device=DEV).manual_seed(19)
    lstm = torch.nn.LSTM(d, 256, batch_first=True).to(DEV)
    with torch.no_grad():
        for p in lstm.parameters():
            p.copy_(torch.randn(p.shape, device=DEV, generator=g) * 0.2)
    x = torch.randn(b, l, d, device=DEV, generator=g) * 2
    h0 = torch.randn(b, 256, device=DEV, generator=g) * 0.5
    c0 = torch.randn(b, 256, device=DEV, generator=g)
    packed, wb = hip.lstm_pack_split(lstm.weight_ih_l0, lstm.weight_hh_l0, lstm.bias_ih_l0, lstm.bias_hh_l0)
    hs, hn, cn, gates, cs = hip.lstm_forward_split(x, h0, c0, packed, wb, save=True)
    rows = _subset(b)
    lstm64 = torch.nn.LSTM(d, 256, batch_first=True).double().to(DEV)
    lstm64.load_state_dict({k: v.double() for k, v in lstm.state_dict().items()})
    with torch.no_grad(), torch.backends.cudnn.flags(enabled=False):
        hs64, (hn64, cn64) = lstm64(x[rows].double(), (h0[rows].double()[None], c0[rows].double()[None]))
    assert float((hs[rows].double() - hs64).abs().max()) < 4e-6
    assert float((hn[rows].double() - hn64[0]).abs().max()) < 4e-6
    assert float((cn[rows].double() - cn64[0]).abs().max()) < 8e-6
    # the saved cell states are the recurrence's own: c_t of the last step is the final state, and the gates are in range
    assert torch.equal(cs[:, -1], cn) and torch.equal(hs[:, -1], hn)
    assert float(gates[:, :, (0, 1, 3)].min()) >= 0.0 and float(gates.abs().max()) <= 1.0
    again = hip.lstm_forward_split(x, h0, c0, packed, wb, save=True)
    assert all(torch.equal(a, c) for a, c in zip((hs, hn, cn, gates, cs), again))
