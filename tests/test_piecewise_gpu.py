"""OPT-IN prototype (RL8_AMD_TOWERS=piecewise): a tower of a scalar observation evaluated from its
exact piecewise-linear table (rl8_amd/nn/piecewise_mlp.py, csrc/piecewise_kernels.hip) against fp64 evaluations of the
same tower (reference arithmetic: src/rl8/models/_feedforward.py:336-375), against the general matrix kernels it must
never replace, and -- through Algorithm -- against the reference's own first-update numbers at north_star's 1e-5."""

import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu

from rl8_amd import hip  # noqa: E402
from rl8_amd.nn import fused_mlp, piecewise_mlp  # noqa: E402

DEV = "cuda:0"


def make_tower(n_out, seed, *, trained=False, dead_units=False, scale=1.0):
    torch.manual_seed(seed)
    tower = nn.Sequential(nn.Linear(1, 256), nn.ReLU(), nn.Linear(256, 256), nn.ReLU(), nn.Linear(256, n_out)).to(DEV)
    if trained:
        opt = torch.optim.Adam(tower.parameters(), 3e-3)
        for _ in range(200):
            xb = (torch.rand(2048, 1, device=DEV) * 2 - 1) * 130
            loss = ((tower(xb) - torch.cat([xb.abs(), torch.sin(xb / 20), xb / 50], 1)[:, :n_out]) ** 2).mean()
            opt.zero_grad()
            loss.backward()
            opt.step()
    with torch.no_grad():
        if dead_units:
            tower[0].weight[::7] = 0.0       # units without a kink: constant in x
            tower[0].bias[::11] = -1e9       # units that never switch on inside any sane range
        tower[4].weight.mul_(scale)
    return tower


def trunk_of(tower):
    """The nesting fused_mlp matches: Sequential(MLP(Linear, ReLU, Linear), ReLU) -- the same modules, so gradients land
    on ``tower``'s parameters."""
    return nn.Sequential(nn.Sequential(tower[0], tower[1], tower[2]), tower[3])


def params_of(tower):
    return tuple(t.detach() for t in (tower[0].weight, tower[0].bias, tower[2].weight, tower[2].bias, tower[4].weight, tower[4].bias))


@pytest.mark.parametrize("m", [1, 63, 4097, 300_000])
@pytest.mark.parametrize("n_out,kw", [(1, {}), (2, {}), (3, {}), (2, dict(trained=True)), (1, dict(dead_units=True)),
                                      (2, dict(trained=True, dead_units=True, scale=50.0))])
def test_forward_from_the_table_is_the_tower(m, n_out, kw):
    tower = make_tower(n_out, 3 + n_out, **kw)
    p = params_of(tower)
    g = torch.Generator(device=DEV).manual_seed(m)
    x = (torch.rand(m, 1, device=DEV, generator=g) * 2 - 1) * 140
    table = piecewise_mlp.build_table(*p)
    assert table is not None and 1 <= table.p <= hip.pw_max_breaks()
    # rows ON breakpoints and one ulp to either side of them
    k = min(m, table.p)
    x[:k, 0] = table.breaks[:k]
    if m > 3 * table.p:
        x[k:2 * k, 0] = torch.nextafter(table.breaks[:k], table.breaks.new_tensor(float("inf")))
        x[2 * k:3 * k, 0] = torch.nextafter(table.breaks[:k], table.breaks.new_tensor(float("-inf")))
    want = tower.double()(x.double())
    tower.float()
    got = hip.pw_tower_forward(x, table.flat, table.p, n_out)
    general = hip.mlp_tower_forward_split(x, p[0], p[1], hip.mlp_pack_w2_f16(p[2]), p[3], p[4], p[5])[0]
    scale = float(want.abs().max()) + 1e-30
    err, err_general = (float((t.double() - want).abs().max()) / scale for t in (got, general))
    assert err < 2e-6, (err, err_general)
    assert err <= 3 * err_general + 3e-7, (err, err_general)
    assert torch.equal(got, hip.pw_tower_forward(x, table.flat, table.p, n_out))


@pytest.mark.parametrize("n_out", [1, 2, 3])
@pytest.mark.parametrize("case", ["plain", "rows_decades_apart", "one_outlier_row", "zero_rows", "tiny"])
def test_segment_sums_are_exact_and_order_independent(n_out, case):
    m = 200_003
    g = torch.Generator(device=DEV).manual_seed(11 + n_out)
    x = (torch.rand(m, 1, device=DEV, generator=g) * 2 - 1) * 130
    d = torch.randn(m, n_out, device=DEV, generator=g) / m
    if case == "rows_decades_apart":
        d *= 10.0 ** torch.randint(-8, 3, (m, 1), device=DEV, generator=g).float()
    elif case == "one_outlier_row":
        d[777] *= 1e9
    elif case == "zero_rows":
        d[torch.rand(m, device=DEV, generator=g) < 0.7] = 0.0
    elif case == "tiny":
        d *= 1e-30
    breaks = torch.sort((torch.rand(300, device=DEV, generator=g) * 2 - 1) * 120).values.contiguous()
    breaks[150] = breaks[151]  # (an empty interval)
    idx = torch.searchsorted(breaks, x[:, 0].contiguous())
    want = torch.zeros(301, 2, n_out, dtype=torch.float64, device=DEV)
    size = torch.zeros_like(want)
    want[:, 0].index_add_(0, idx, d.double())
    want[:, 1].index_add_(0, idx, d.double() * x.double())
    size[:, 0].index_add_(0, idx, d.double().abs())
    size[:, 1].index_add_(0, idx, (d.double() * x.double()).abs())
    got = hip.pw_segment_sums(x, d, breaks, 300)
    # fp64 index_add rounds once per addition; the kernel adds integers -- every term on a common scale, 74 bits below
    # the call's largest possible term -- and rounds twice in all: agreement to fp64's own accumulation error, plus the
    # integer grid's half step per row where one outlier row has lifted the scale
    count = torch.zeros(301, dtype=torch.float64, device=DEV).index_add_(0, idx, torch.ones(m, dtype=torch.float64, device=DEV))
    grid = float(d.abs().max()) * max(float(x.abs().max()), 1.0) * 2.0 ** -72
    assert bool(((got - want).abs() <= 1e-12 * size + grid * count[:, None, None] + 1e-300).all())
    perm = torch.randperm(m, device=DEV, generator=g)
    assert torch.equal(got, hip.pw_segment_sums(x[perm].contiguous(), d[perm].contiguous(), breaks, 300))   # any order: same bits
    assert torch.equal(got, hip.pw_segment_sums(x, d, breaks, 300))
    d[5, 0] = float("nan")
    assert bool(torch.isnan(hip.pw_segment_sums(x, d, breaks, 300)).all())


@pytest.mark.parametrize("n_out,kw", [(1, {}), (2, {}), (3, dict(trained=True)), (2, dict(trained=True, dead_units=True))])
@pytest.mark.parametrize("case", ["plain", "rows_decades_apart", "clipped_rows"])
def test_backward_from_the_interval_sums_is_the_towers_gradient(n_out, kw, case, monkeypatch):
    """Every parameter gradient against fp64 autograd of the same tower, entry by entry relative to the tensor's largest,
    beside the general kernels' own error on the same inputs."""
    m = 150_000
    tower = make_tower(n_out, 21 + n_out, **kw)
    g = torch.Generator(device=DEV).manual_seed(5)
    x = (torch.rand(m, 1, device=DEV, generator=g) * 2 - 1) * 130
    d = torch.randn(m, n_out, device=DEV, generator=g) / m
    if case == "rows_decades_apart":
        d *= 10.0 ** torch.randint(-4, 3, (m, 1), device=DEV, generator=g).float()
    elif case == "clipped_rows":
        d[torch.rand(m, device=DEV, generator=g) < 0.7] = 0.0
    tower.double()
    tower.zero_grad()
    (tower(x.double()) * d.double()).sum().backward()
    want = [t.grad.clone() for t in tower.parameters()]
    tower.float()

    def grads(piecewise):
        monkeypatch.setattr(piecewise_mlp, "ENABLED", piecewise)
        tower.zero_grad()
        before = dict(piecewise_mlp.stats)
        out = fused_mlp.tower_forward(trunk_of(tower), [tower[4]], x)
        assert out is not None
        (out * d).sum().backward()
        assert (piecewise_mlp.stats["backwards"] - before["backwards"]) == int(piecewise)
        return [t.grad.clone() for t in tower.parameters()]

    got, general = grads(True), grads(False)
    for name, gw, gg, w in zip(("w1", "b1", "w2", "b2", "w3", "b3"), got, general, want):
        top = float(w.abs().max()) + 1e-300
        err, err_general = (float((t.double() - w).abs().max()) / top for t in (gw, gg))
        assert err < 5e-6, (name, err, err_general)
        assert err <= 3 * err_general + 5e-7, (name, err, err_general)


@pytest.mark.parametrize("variant", ["ff_discrete", "ff_continuous_squashed", "ff_continuous_normal"])
def test_first_update_matches_the_reference_to_1e5_from_tables(golden, variant, monkeypatch):
    """The assembled path with both towers evaluated from tables, against the REFERENCE's first StatTracker update,
    StepStats, clipped gradient and weights (tests/golden/first_update_*.npz) at the bars of the matrix kernels."""
    from .test_first_update_gpu import check_one_sgd_iteration

    monkeypatch.setattr(piecewise_mlp, "ENABLED", True)
    check_one_sgd_iteration(golden, variant, towers="piecewise")


def test_a_table_too_large_for_the_kernels_steps_aside(monkeypatch):
    tower = make_tower(2, 1)
    monkeypatch.setattr(piecewise_mlp, "ENABLED", True)
    monkeypatch.setattr(hip, "pw_max_breaks", lambda: 8)
    before = piecewise_mlp.stats["declined"]
    x = torch.randn(1000, 1, device=DEV)
    out = fused_mlp.tower_forward(trunk_of(tower), [tower[4]], x)
    assert piecewise_mlp.stats["declined"] == before + 1
    assert torch.equal(out, hip.mlp_tower_forward_split(x, *[t.detach() for t in (tower[0].weight, tower[0].bias)],
                                                        hip.mlp_pack_w2_f16(tower[2].weight.detach()), tower[2].bias.detach(),
                                                        tower[4].weight.detach(), tower[4].bias.detach())[0])


@pytest.mark.parametrize("env_name", ["discrete", "continuous_squashed"])
def test_six_updates_from_tables_track_the_matrix_towers(env_name, monkeypatch):
    """Six collect() + step() rounds of the same seeded algorithm with the towers from tables and from the matrix kernels:
    the two evaluate the same function to fp32 rounding, so rollouts (same Philox noise), losses and the weights after
    six updates stay together -- collect statistics to 1e-4, losses to 2e-3 of their size, weights to 2e-3 of the
    largest; from the third update on 5e-4 and 4e-2.  What separates two roundings of the same update (traced in round 5, when the
    rows-shape general data gradient changed dW1's last bits and moved the KL of update 3 by 0.4 %): a nearly dead ReLU
    unit of layer 2 that ONE sample opens in one run and not in the other -- its row of dW2 goes from exactly zero to
    something tiny, and Adam's normalisation makes a full-size step of it (109 entries of W2 moved by 1e-4, everything
    else agreed to 6e-8).  A property of ReLU + Adam, not of either kernel: each kernel matches fp64 on the same
    inputs to rounding (tests/test_mlp_split_gpu.py)."""
    from rl8_amd import AlgorithmConfig
    from rl8_amd.distributions import SquashedNormal
    from rl8_amd.env import ContinuousDummyEnv, DiscreteDummyEnv

    def run(tables, tile="0"):
        monkeypatch.setattr(piecewise_mlp, "ENABLED", tables)
        monkeypatch.setenv("RL8_MLP_DGRAD_TILE", tile)  # (read per call by rl8_mlp_tower_backward_f16_f32)
        torch.manual_seed(5)
        if env_name == "discrete":
            algo = AlgorithmConfig(num_envs=4096, horizon=16).build(DiscreteDummyEnv)
        else:
            algo = AlgorithmConfig(num_envs=4096, horizon=16, distribution_cls=SquashedNormal).build(ContinuousDummyEnv)
        out = [(algo.collect(), algo.step()) for _ in range(6)]
        return out, torch.cat([p.detach().flatten() for p in algo.policy.model.parameters()])

    def compare(a, b, loose_from, what):
        for update, ((c0, s0), (c1, s1)) in enumerate(zip(a, b)):
            loose = 1.0 if update < loose_from else 5.0
            for k in ("returns/mean", "rewards/mean", "returns/std"):
                assert c1[k] == pytest.approx(c0[k], rel=1e-4 * loose), (what, update, k)
            for k in ("losses/policy", "losses/vf", "losses/total", "monitors/kl_div"):
                assert s1[k] == pytest.approx(s0[k], rel=2e-3 * loose * (4.0 if update >= loose_from else 1.0), abs=2e-6), (what, update, k)

    before = dict(piecewise_mlp.stats)
    #  The ORIGINAL bars for all six updates against the matrix towers pinned to the data-gradient kernel they were
    # set with (RL8_MLP_DGRAD_TILE=1: the tile kernel of rounds 2-4, general heads only -- the discrete variant's
    # rank-one heads never ran it); the five-times bars from the third update on only between the tables and the
    # PRODUCT's rows-shape data gradient, where the flipped unit was traced.
    (pinned, w_pinned), (matrix, w_matrix), (tables, w_tables) = run(False, "1"), run(False), run(True)
    assert piecewise_mlp.stats["forwards"] > before["forwards"] and piecewise_mlp.stats["backwards"] > before["backwards"]
    compare(pinned, tables, 6, "tile-pinned matrix towers vs tables")
    compare(matrix, tables, 2, "product matrix towers vs tables")
    assert float((w_tables - w_pinned).abs().max()) <= 2e-3 * float(w_pinned.abs().max())
    assert float((w_tables - w_matrix).abs().max()) <= 2e-3 * float(w_matrix.abs().max())
