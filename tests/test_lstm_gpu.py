"""a-9: the fused LSTM kernels against torch.nn.LSTM (one layer, hidden 256,
batch_first -- what the reference's default recurrent models use,
src/rl8/models/_recurrent.py:201-321), through the C ABI."""

import pytest
import torch

pytestmark = pytest.mark.gpu

from rl8_amd import hip  # noqa: E402

DEV = "cuda:0"


def reference_lstm(d_in, seed):
    torch.manual_seed(seed)
    return torch.nn.LSTM(d_in, 256, num_layers=1, batch_first=True).to(DEV)


def pack(lstm):
    return hip.lstm_pack(lstm.weight_ih_l0, lstm.weight_hh_l0, lstm.bias_ih_l0, lstm.bias_hh_l0)


@pytest.mark.parametrize("b,l,d_in", [(1, 1, 1), (31, 3, 1), (32, 4, 2), (100, 5, 5), (1000, 8, 1), (4097, 2, 7)])
def test_lstm_forward_matches_torch(b, l, d_in):
    lstm = reference_lstm(d_in, b + l)
    g = torch.Generator(device=DEV).manual_seed(b)
    x = torch.randn(b, l, d_in, device=DEV, generator=g) * 3
    h0 = torch.randn(b, 256, device=DEV, generator=g) * 0.5
    c0 = torch.randn(b, 256, device=DEV, generator=g)
    with torch.no_grad(), torch.backends.cudnn.flags(enabled=False):
        want, (hn_w, cn_w) = lstm(x, (h0.unsqueeze(0), c0.unsqueeze(0)))
    packed = pack(lstm)
    hs, hn, cn, gates, cs = hip.lstm_forward(x, h0, c0, packed, save=True)
    torch.testing.assert_close(hs, want, rtol=1e-5, atol=2e-6)
    torch.testing.assert_close(hn, hn_w[0], rtol=1e-5, atol=2e-6)
    torch.testing.assert_close(cn, cn_w[0], rtol=1e-5, atol=4e-6)
    assert torch.equal(hs[:, -1], hn) and torch.equal(cs[:, -1], cn)
    # saved gates reproduce the cell update: c_t = f c_{t-1} + i g, h_t = o tanh(c_t)
    i, f, gg, o = gates.unbind(2)
    c_prev = torch.cat([c0.unsqueeze(1), cs[:, :-1]], 1)
    torch.testing.assert_close(cs, f * c_prev + i * gg, rtol=1e-5, atol=2e-6)
    torch.testing.assert_close(hs, o * torch.tanh(cs), rtol=1e-5, atol=2e-6)
    # inference launch: same numbers, nothing saved
    hs2, hn2, cn2, none_g, none_c = hip.lstm_forward(x, h0, c0, packed)
    assert none_g is None and none_c is None
    assert torch.equal(hs2, hs) and torch.equal(hn2, hn) and torch.equal(cn2, cn)


def test_lstm_argument_checks():
    assert hip.lstm_supports(1) and hip.lstm_supports(7) and not hip.lstm_supports(8) and not hip.lstm_supports(0)
    lstm = reference_lstm(1, 0)
    packed = pack(lstm)
    x = torch.zeros(4, 2, 1, device=DEV)
    with pytest.raises(ValueError):
        hip.lstm_forward(x, torch.zeros(4, 128, device=DEV), torch.zeros(4, 256, device=DEV), packed)
    with pytest.raises(ValueError):
        hip.lstm_pack(lstm.weight_ih_l0, lstm.weight_hh_l0[:, :128], lstm.bias_ih_l0, lstm.bias_hh_l0)


@pytest.mark.parametrize("b,l,d_in", [(1, 1, 1), (33, 3, 1), (64, 4, 2), (300, 5, 5), (2000, 8, 1), (4097, 2, 7)])
def test_lstm_backward_matches_autograd(b, l, d_in):
    lstm = reference_lstm(d_in, b + 7 * l)
    g = torch.Generator(device=DEV).manual_seed(b + 1)
    x = torch.randn(b, l, d_in, device=DEV, generator=g) * 3
    h0 = torch.randn(b, 256, device=DEV, generator=g) * 0.5
    c0 = torch.randn(b, 256, device=DEV, generator=g)
    dhs = torch.randn(b, l, 256, device=DEV, generator=g) / (b * l)
    with torch.backends.cudnn.flags(enabled=False):
        out, _ = lstm(x, (h0.unsqueeze(0), c0.unsqueeze(0)))
    out.backward(dhs)
    hs, hn, cn, gates, cs = hip.lstm_forward(x, h0, c0, pack(lstm), save=True)
    grads = hip.lstm_backward(x, h0, c0, hs, gates, cs, dhs, hip.lstm_pack_transposed(lstm.weight_hh_l0))

    def close(got, want, name):
        scale = float(want.abs().max()) + 1e-12
        assert float((got - want).abs().max()) / scale < 2e-5, (name, float((got - want).abs().max()), scale)

    close(grads["w_hh"], lstm.weight_hh_l0.grad, "w_hh")
    close(grads["w_ih"], lstm.weight_ih_l0.grad, "w_ih")
    close(grads["b"], lstm.bias_ih_l0.grad, "b_ih")
    close(grads["b"], lstm.bias_hh_l0.grad, "b_hh")


@pytest.mark.parametrize("m,n", [(1, 1), (67, 3), (5000, 2), (100_001, 8)])
def test_linear_heads_match_torch(m, n):
    g = torch.Generator(device=DEV).manual_seed(m)
    h = torch.randn(m, 256, device=DEV, generator=g)
    w = (torch.randn(n, 256, device=DEV, generator=g) / 16).requires_grad_(True)
    b = torch.randn(n, device=DEV, generator=g).requires_grad_(True)
    dout = torch.randn(m, n, device=DEV, generator=g) / m
    hr = h.clone().requires_grad_(True)
    want = torch.nn.functional.linear(hr.double(), w.double(), b.double())
    want.backward(dout.double())
    out = hip.linear_heads_forward(h, w, b)
    torch.testing.assert_close(out.double(), want, rtol=1e-5, atol=1e-5)
    dh, dw, db = hip.linear_heads_backward(h, dout, w)
    torch.testing.assert_close(dh.double(), hr.grad.double(), rtol=1e-5, atol=1e-9)
    torch.testing.assert_close(dw.double(), w.grad.double(), rtol=1e-4, atol=1e-7)
    torch.testing.assert_close(db.double(), b.grad.double(), rtol=1e-4, atol=1e-7)


@pytest.mark.parametrize("m,n_a,n_b", [(1, 1, 1), (67, 2, 1), (8192, 2, 1), (100_001, 5, 3)])
def test_linear_heads_pair_is_the_two_single_launches(m, n_a, n_b):
    """The rollout's logits + value heads in one pass over h: bit-identical to one launch per layer."""
    g = torch.Generator(device=DEV).manual_seed(m + n_a)
    h = torch.randn(m, 256, device=DEV, generator=g)
    w_a, w_b = torch.randn(n_a, 256, device=DEV, generator=g) / 16, torch.randn(n_b, 256, device=DEV, generator=g) / 16
    b_a, b_b = torch.randn(n_a, device=DEV, generator=g), torch.randn(n_b, device=DEV, generator=g)
    out_a, out_b = hip.linear_heads_forward_pair(h, w_a, b_a, w_b, b_b)
    assert torch.equal(out_a, hip.linear_heads_forward(h, w_a, b_a))
    assert torch.equal(out_b, hip.linear_heads_forward(h, w_b, b_b))


@pytest.mark.parametrize("b", [1, 127, 128, 1000, 40_000])
def test_state_split_reports_max_abs_h0(b):
    """rl8_lstm_split_state_bound: the same planes as rl8_lstm_split_state, and max |h| exactly."""
    g = torch.Generator(device=DEV).manual_seed(b)
    h = (torch.rand(b, 256, device=DEV, generator=g) * 2 - 1) * 3.5
    h[b // 2, 17] = -7.25 if b > 1 else 0.5
    lib = hip.load()
    n = int(lib.rl8_lstm_split_state_bytes(b))
    planes, planes_b = torch.zeros(n, dtype=torch.uint8, device=DEV), torch.zeros(n, dtype=torch.uint8, device=DEV)
    bound = torch.full((1,), -1.0, device=DEV)
    hip._check(lib.rl8_lstm_split_state(hip._ptr(h), 256, b, hip._ptr(planes), hip._stream()), "rl8_lstm_split_state")
    hip._check(lib.rl8_lstm_split_state_bound(hip._ptr(h), 256, b, hip._ptr(planes_b), hip._ptr(bound), hip._stream()),
               "rl8_lstm_split_state_bound")
    assert torch.equal(planes, planes_b)
    assert float(bound) == float(h.abs().max())
    assert lib.rl8_lstm_split_state_bound(hip._ptr(h), 256, b, hip._ptr(planes_b), None, hip._stream()) != 0


def test_fused_recurrent_model_matches_the_eager_modules():
    """The default recurrent model through the fused LSTM + heads must give the
    outputs and parameter gradients of the same modules run by PyTorch."""
    from rl8_amd.data import DataKeys
    from rl8_amd.env import DiscreteDummyEnv
    from rl8_amd.models_recurrent import DefaultDiscreteRecurrentModel
    from rl8_amd.nn import fused_lstm
    from rl8_amd.tensordict import TensorDict

    env = DiscreteDummyEnv(4, 8, device=DEV)
    torch.manual_seed(2)
    model = DefaultDiscreteRecurrentModel(env.observation_spec, env.action_spec).to(DEV)
    b, l = 300, 4
    g = torch.Generator(device=DEV).manual_seed(0)
    obs = torch.randn(b, l, 1, device=DEV, generator=g) * 10
    states = TensorDict(
        {DataKeys.HIDDEN_STATES: torch.randn(b, l, 1, 256, device=DEV, generator=g) * 0.3,
         DataKeys.CELL_STATES: torch.randn(b, l, 1, 256, device=DEV, generator=g)}, batch_size=[b, l])
    w_logits = torch.randn(b * l, 1, 2, device=DEV, generator=g)
    w_value = torch.randn(b * l, 1, device=DEV, generator=g)

    def run(enabled):
        fused_lstm.ENABLED = enabled
        try:
            model.zero_grad()
            feats, new_states = model(TensorDict({DataKeys.OBS: obs}, batch_size=[b, l]), states)
            ((feats["logits"] * w_logits).sum() + (model.value_function() * w_value).sum()).backward()
            return (feats["logits"].detach().clone(), model.value_function().detach().clone(),
                    new_states[DataKeys.HIDDEN_STATES].detach().clone(), new_states[DataKeys.CELL_STATES].detach().clone(),
                    {k: p.grad.clone() for k, p in model.named_parameters()})
        finally:
            fused_lstm.ENABLED = True

    fused, eager = run(True), run(False)
    for a, e in zip(fused[:4], eager[:4]):
        torch.testing.assert_close(a, e, rtol=1e-5, atol=2e-6)
    for k in eager[4]:
        scale = float(eager[4][k].abs().max()) + 1e-12
        assert float((fused[4][k] - eager[4][k]).abs().max()) / scale < 5e-5, k


def test_lstm_kernels_repeat_bit_for_bit_under_load():
    """Thirty back-to-back forward + backward runs at a config-5-like shape must
    return the same bits (fixed summation orders; these kernels use packed fp32
    ops beside fp32 MFMAs -- the combination that, beside bf16 MFMAs, once
    produced wrong lanes in the tower kernels)."""
    b, l, d_in = 8192, 4, 1
    lstm = reference_lstm(d_in, 3)
    g = torch.Generator(device=DEV).manual_seed(17)
    x = torch.randn(b, l, d_in, device=DEV, generator=g) * 3
    h0 = torch.randn(b, 256, device=DEV, generator=g) * 0.5
    c0 = torch.randn(b, 256, device=DEV, generator=g)
    dhs = torch.randn(b, l, 256, device=DEV, generator=g) / (b * l)
    packed, packed_t = pack(lstm), hip.lstm_pack_transposed(lstm.weight_hh_l0)

    def run():
        hs, hn, cn, gates, cs = hip.lstm_forward(x, h0, c0, packed, save=True)
        grads = hip.lstm_backward(x, h0, c0, hs, gates, cs, dhs, packed_t)
        return [hs, hn, cn, gates, cs, grads["w_hh"], grads["w_ih"], grads["b"]]

    first = run()
    for trial in range(30):
        for a, c in zip(first, run()):
            assert torch.equal(a, c), trial


@pytest.mark.parametrize("b,l,d", [(1, 1, 1), (127, 4, 1), (129, 3, 2), (1000, 4, 5), (4097, 2, 3), (640, 7, 1),
                                   (300, 4, 4), (2000, 3, 6), (129, 5, 7)])   # (round 6: every width up to seven)
def test_lstm_split_step_matches_the_fp32_kernel_and_fp64(b, l, d):
    """The bf16-plane step kernel (one launch per timestep, h_{t-1} through bf16 planes)
    against the fp32-MFMA kernel on the same inputs -- outputs, final states, saved gates
    and cell states -- and both against an fp64 LSTM: the split kernel must be as close to
    fp64 as the fp32 kernel is (fp32-accurate products, same gate arithmetic)."""
    assert hip.lstm_split_supports(d)
    g = torch.Generator(device=DEV).manual_seed(100 * b + 10 * l + d)
    lstm = torch.nn.LSTM(d, 256, batch_first=True).to(DEV)
    with torch.no_grad():
        for p in lstm.parameters():
            p.copy_(torch.randn(p.shape, device=DEV, generator=g) * 0.2)
    x = torch.randn(b, l, d, device=DEV, generator=g) * 2
    h0 = torch.randn(b, 256, device=DEV, generator=g) * 0.5
    c0 = torch.randn(b, 256, device=DEV, generator=g)
    params = (lstm.weight_ih_l0, lstm.weight_hh_l0, lstm.bias_ih_l0, lstm.bias_hh_l0)
    ref = hip.lstm_forward(x, h0, c0, hip.lstm_pack(*params), save=True)
    packed, wb = hip.lstm_pack_split(*params)
    got = hip.lstm_forward_split(x, h0, c0, packed, wb, save=True)
    lstm64 = torch.nn.LSTM(d, 256, batch_first=True).double().to(DEV)
    lstm64.load_state_dict({k: v.double() for k, v in lstm.state_dict().items()})
    with torch.no_grad(), torch.backends.cudnn.flags(enabled=False):
        hs64, (hn64, cn64) = lstm64(x.double(), (h0.double()[None], c0.double()[None]))
    names = ("hs", "hn", "cn", "gates", "cs")
    for name, a, r in zip(names, got, ref):
        # (two fp32 evaluations of pre-activations of size ~3: a few 1e-6 apart)
        torch.testing.assert_close(a, r, rtol=2e-5, atol=6e-6, msg=name)
    for name, a, r, want in (("hs", got[0], ref[0], hs64), ("hn", got[1], ref[1], hn64[0]), ("cn", got[2], ref[2], cn64[0])):
        err_split = float((a.double() - want).abs().max())
        err_f32 = float((r.double() - want).abs().max())
        assert err_split <= max(2.0 * err_f32, 2e-6), (name, err_split, err_f32)
    # inference launch (no saved gates) returns the same states
    again = hip.lstm_forward_split(x, h0, c0, packed, wb, save=False)
    assert torch.equal(again[0], got[0]) and again[3] is None


def _rows_backward_inputs(b, l, seed):
    g = torch.Generator(device=DEV).manual_seed(seed)
    r = lambda *s: torch.rand(*s, device=DEV, generator=g)  # noqa: E731
    gates = r(b, l, 4, 256)
    gates[:, :, 2] = gates[:, :, 2] * 2 - 1            # i, f, o in (0, 1), g in (-1, 1)
    cs = (r(b, l, 256) * 2 - 1) * 1.5
    c0 = (r(b, 256) * 2 - 1) * 1.5
    # gradients of a mean loss: tiny, and of very different size from sequence to sequence
    dhs = (r(b, l, 256) * 2 - 1) * torch.exp(-12 * r(b, 1, 1)) * 1e-3
    w_hh = (r(1024, 256) * 2 - 1) / 16
    return c0, gates, cs, dhs, w_hh


def _backward_through_time_fp64(c0, gates, cs, dhs, w_hh):
    b, l = dhs.shape[:2]
    c0, gates, cs, dhs, w = (t.double() for t in (c0, gates, cs, dhs, w_hh))
    dg = torch.zeros(b, l, 4, 256, dtype=torch.float64, device=DEV)
    dh_carry = torch.zeros(b, 256, dtype=torch.float64, device=DEV)
    dc_carry = torch.zeros_like(dh_carry)
    for t in range(l - 1, -1, -1):
        i, f, g, o = gates[:, t, 0], gates[:, t, 1], gates[:, t, 2], gates[:, t, 3]
        c_prev = cs[:, t - 1] if t > 0 else c0
        dh = dhs[:, t] + dh_carry
        tc = torch.tanh(cs[:, t])
        dg[:, t, 3] = dh * tc * o * (1 - o)
        dc = dh * o * (1 - tc * tc) + dc_carry
        dg[:, t, 0] = dc * g * i * (1 - i)
        dg[:, t, 2] = dc * i * (1 - g * g)
        dg[:, t, 1] = dc * c_prev * f * (1 - f)
        dc_carry = dc * f
        dh_carry = dg[:, t].reshape(b, 1024) @ w
    return dg


@pytest.mark.parametrize("b,l", [(1, 1), (31, 4), (32, 2), (100, 7), (128, 4), (129, 3), (1000, 4), (4101, 2), (33000, 4)])
def test_lstm_rows_backward_matches_fp64(b, l):
    """The backward through time on bf16 planes (rl8_lstm_rows_backward_f32; a wave per 32 sequences, ragged last wave and
    last workgroup, one to seven steps) against an fp64 model of the recurrences: per sequence within 2e-6 of that
    sequence's largest gate gradient, whatever the sequence's scale (dh spans five decades between sequences: nothing in the
    kernel is scaled per launch)."""
    c0, gates, cs, dhs, w_hh = _rows_backward_inputs(b, l, 1000 * b + l)
    got = hip.lstm_rows_backward(c0, gates, cs, dhs, hip.lstm_rows_backward_pack(w_hh))
    want = _backward_through_time_fp64(c0, gates, cs, dhs, w_hh)
    assert bool(torch.isfinite(got).all())
    scale = want.abs().amax(dim=(1, 2, 3), keepdim=True).clamp_min(1e-300)
    assert float(((got.double() - want).abs() / scale).max()) < 2e-6


@pytest.mark.parametrize("b,l,n", [(1, 1, 1), (31, 4, 3), (100, 7, 2), (128, 4, 4), (129, 3, 3), (4101, 2, 1), (33000, 4, 3)])
def test_lstm_rows_backward_with_the_heads_gradient_formed_inside(b, l, n):
    """rl8_lstm_rows_backward_heads_f32: dL/dh_t = dOut x W of the output heads formed in the kernel from four floats per
    row-step, against the fp64 recurrences fed with that product -- the bound of the array form -- and against the array
    form itself; the bound on |dG| it leaves is the maximum of what it wrote; repeated launches agree bit for bit."""
    c0, gates, cs, _, w_hh = _rows_backward_inputs(b, l, 77 * b + l + n)
    g = torch.Generator(device=DEV).manual_seed(b + n)
    dout = (torch.rand(b * l, n, device=DEV, generator=g) * 2 - 1) * torch.exp(-12 * torch.rand(b, 1, device=DEV, generator=g)).repeat_interleave(l, 0) * 1e-3
    w = (torch.rand(n, 256, device=DEV, generator=g) * 2 - 1) / 16
    packed = hip.lstm_rows_backward_pack(w_hh)
    got, bound = hip.lstm_rows_backward(c0, gates, cs, None, packed, with_bound=True, heads=(dout, w))
    dhs64 = (dout.double() @ w.double()).view(b, l, 256)
    want = _backward_through_time_fp64(c0, gates, cs, dhs64, w_hh)
    assert bool(torch.isfinite(got).all())
    scale = want.abs().amax(dim=(1, 2, 3), keepdim=True).clamp_min(1e-300)
    assert float(((got.double() - want).abs() / scale).max()) < 2e-6
    array_form = hip.lstm_rows_backward(c0, gates, cs, dhs64.float(), packed)
    assert float(((got.double() - array_form.double()).abs() / scale).max()) < 1e-6
    assert float(bound) == float(got.abs().max())
    assert torch.equal(got, hip.lstm_rows_backward(c0, gates, cs, None, packed, heads=(dout, w)))
    with pytest.raises(ValueError, match="either dhs or heads"):
        hip.lstm_rows_backward(c0, gates, cs, array_form[:, :, 0], packed, heads=(dout, w))


@pytest.mark.parametrize("case", ["plain", "large_cell_states", "zero_and_tiny_rows", "huge_gradients"])
@pytest.mark.parametrize("b,l,n", [(129, 4, 3), (4101, 3, 2), (1000, 7, 1)])
def test_lstm_rows_backward_heads_form_on_fp16_and_bf16_planes(b, l, n, case, monkeypatch):
    """Round 6: the HEADS form runs the recurrent product dG x W_hh on TWO fp16 planes per operand (three plane products)
    with dG scaled by a power of two per sequence and step from a bound known before the step's first chunk (|dOut| x
    max|W_heads| + the carried max|dh| + max|dc|, placed at 2^6 so that |c_{t-1}| up to 4 094 cannot overflow a plane)
    -- against the same kernel on the three exact bf16 planes (RL8_AMD_LSTM_BACKWARD_PLANES=bf16, read per call) and the
    fp64 recurrences, at the bars of the bf16 form: 2e-6 of each sequence's largest gate gradient.  Cases: cell states up
    to +-60 (the f gate's gradient carries |c_{t-1}|), sequences whose gradient is zero or ~1e-30, gradients of 1e+20."""
    c0, gates, cs, _, w_hh = _rows_backward_inputs(b, l, 31 * b + l + n)
    g = torch.Generator(device=DEV).manual_seed(b + 5 * n)
    rowscale = torch.exp(-12 * torch.rand(b, 1, device=DEV, generator=g)).repeat_interleave(l, 0) * 1e-3
    dout = (torch.rand(b * l, n, device=DEV, generator=g) * 2 - 1) * rowscale
    if case == "large_cell_states":
        cs, c0 = cs * 40.0, c0 * 40.0
    elif case == "zero_and_tiny_rows":
        dout = dout.view(b, l, n)
        dout[::3] = 0.0
        dout[1::3] *= 1e-20   # (|dG| down to ~1e-30: fp32 still normal, far below 2^-80)
        dout = dout.view(b * l, n).contiguous()
    elif case == "huge_gradients":
        dout = dout * 1e23
    w = (torch.rand(n, 256, device=DEV, generator=g) * 2 - 1) / 16
    packed = hip.lstm_rows_backward_pack(w_hh)
    monkeypatch.delenv("RL8_AMD_LSTM_BACKWARD_PLANES", raising=False)
    f16, f16_bound = hip.lstm_rows_backward(c0, gates, cs, None, packed, with_bound=True, heads=(dout, w))
    monkeypatch.setenv("RL8_AMD_LSTM_BACKWARD_PLANES", "bf16")
    bf16 = hip.lstm_rows_backward(c0, gates, cs, None, packed, heads=(dout, w))
    monkeypatch.delenv("RL8_AMD_LSTM_BACKWARD_PLANES")
    want = _backward_through_time_fp64(c0, gates, cs, (dout.double() @ w.double()).view(b, l, 256), w_hh)
    assert bool(torch.isfinite(f16).all()) and bool(torch.isfinite(bf16).all())
    scale = want.abs().amax(dim=(1, 2, 3), keepdim=True).clamp_min(1e-300)
    err16 = float(((f16.double() - want).abs() / scale).max())
    err_bf = float(((bf16.double() - want).abs() / scale).max())
    # (cell states of +-60: 1 - tanh^2 c cancels in fp32 in BOTH schemes' gate arithmetic -- the same code -- so the bar
    # of that case is the exact planes' own error)
    bar = 2e-5 if case == "large_cell_states" else 2e-6
    assert err16 < bar and err_bf < bar, (err16, err_bf)
    assert err16 <= 3 * err_bf + 2e-7, (err16, err_bf)   # as accurate as the exact planes
    assert float(f16_bound) == float(f16.abs().max())
    # the last step of every sequence has no carried gradient: both plane schemes form the same dG there, bit for bit
    assert torch.equal(f16[:, l - 1], bf16[:, l - 1])
    assert torch.equal(f16, hip.lstm_rows_backward(c0, gates, cs, None, packed, heads=(dout, w)))  # repeatable


def test_recurrent_model_with_and_without_the_fused_heads_node():
    """A training pass of the default recurrent models as one LSTM + heads node (fused_lstm.lstm_heads_forward) and as
    the two nodes it replaces: same outputs bit for bit (the same forward kernels), parameter gradients to 2e-6 of each
    tensor's largest entry -- for the discrete model (3 head outputs) and the continuous one (3), and with a consumer of
    the final hidden state beside the heads (the node then falls back to the array form inside)."""
    from rl8_amd.data import DataKeys
    from rl8_amd.env import ContinuousDummyEnv, DiscreteDummyEnv
    from rl8_amd.models_recurrent import DefaultContinuousRecurrentModel, DefaultDiscreteRecurrentModel
    from rl8_amd.nn import fused_lstm
    from rl8_amd.tensordict import TensorDict

    b, l = 517, 4
    g = torch.Generator(device=DEV).manual_seed(3)
    obs = torch.randn(b, l, 1, device=DEV, generator=g) * 10
    states = TensorDict(
        {DataKeys.HIDDEN_STATES: torch.randn(b, l, 1, 256, device=DEV, generator=g) * 0.3,
         DataKeys.CELL_STATES: torch.randn(b, l, 1, 256, device=DEV, generator=g)}, batch_size=[b, l])
    for env_cls, model_cls in ((DiscreteDummyEnv, DefaultDiscreteRecurrentModel), (ContinuousDummyEnv, DefaultContinuousRecurrentModel)):
        env = env_cls(4, 8, device=DEV)
        torch.manual_seed(5)
        model = model_cls(env.observation_spec, env.action_spec).to(DEV)
        w_value = torch.randn(b * l, 1, device=DEV, generator=g) / (b * l)
        w_state = torch.randn(b, 1, 256, device=DEV, generator=g) / b
        for use_state in (False, True):
            def run(fuse):
                fused_lstm.FUSE_HEADS = fuse
                try:
                    model.zero_grad()
                    feats, new_states = model(TensorDict({DataKeys.OBS: obs}, batch_size=[b, l]), states)
                    loss = sum((v * (i + 1)).sum() for i, v in enumerate(feats.values())) / (b * l) + (model.value_function() * w_value).sum()
                    if use_state:
                        loss = loss + (new_states[DataKeys.HIDDEN_STATES] * w_state).sum()
                    loss.backward()
                    return ([v.detach().clone() for v in feats.values()] + [model.value_function().detach().clone()],
                            {k: p.grad.clone() for k, p in model.named_parameters()})
                finally:
                    fused_lstm.FUSE_HEADS = True

            one, two = run(True), run(False)
            for a, e in zip(one[0], two[0]):
                assert torch.equal(a, e)
            for k in two[1]:
                scale = float(two[1][k].abs().max()) + 1e-30
                assert float((one[1][k] - two[1][k]).abs().max()) / scale < 2e-6, (model_cls.__name__, use_state, k)


@pytest.mark.parametrize("b,l,d_in", [(33, 3, 1), (300, 5, 5), (2000, 8, 1), (4097, 2, 3), (300, 4, 4), (2000, 3, 6), (1000, 4, 7)])
def test_lstm_backward_on_planes_matches_autograd(b, l, d_in):
    """hip.lstm_backward with the rows kernel (what fused_lstm runs by default) against torch's autograd, as
    test_lstm_backward_matches_autograd holds the fp32-MFMA kernel; and the two kernels' gradients beside each other."""
    lstm = reference_lstm(d_in, b + 7 * l)
    g = torch.Generator(device=DEV).manual_seed(b + 1)
    x = torch.randn(b, l, d_in, device=DEV, generator=g) * 3
    h0 = torch.randn(b, 256, device=DEV, generator=g) * 0.5
    c0 = torch.randn(b, 256, device=DEV, generator=g)
    dhs = torch.randn(b, l, 256, device=DEV, generator=g) / (b * l)
    with torch.backends.cudnn.flags(enabled=False):
        out, _ = lstm(x, (h0.unsqueeze(0), c0.unsqueeze(0)))
    out.backward(dhs)
    hs, hn, cn, gates, cs = hip.lstm_forward(x, h0, c0, pack(lstm), save=True)
    grads = hip.lstm_backward(x, h0, c0, hs, gates, cs, dhs, None, split=True,
                              rows_packed=hip.lstm_rows_backward_pack(lstm.weight_hh_l0))
    old = hip.lstm_backward(x, h0, c0, hs, gates, cs, dhs, hip.lstm_pack_transposed(lstm.weight_hh_l0), split=True)
    for name, want in (("w_hh", lstm.weight_hh_l0.grad), ("w_ih", lstm.weight_ih_l0.grad), ("b", lstm.bias_ih_l0.grad)):
        scale = float(want.abs().max()) + 1e-12
        assert float((grads[name] - want).abs().max()) / scale < 2e-5, name
        assert float((grads[name] - old[name]).abs().max()) / scale < 5e-6, name
    with pytest.raises(ValueError, match="rows_packed needs"):
        hip.lstm_backward(x, h0, c0, hs, gates, cs, dhs, None, split=False, rows_packed=hip.lstm_rows_backward_pack(lstm.weight_hh_l0))


@pytest.mark.parametrize("case", ["plain", "one_outlier_sequence", "six_decades"])
def test_lstm_weight_gradient_planes_hold_over_the_dynamic_range(case, monkeypatch):
    """The LSTM's weight gradient runs on two fp16 planes per operand WITHOUT a guard on the data:
    dG is scaled by its true maximum and its low plane is wide.  Entry by entry against fp64 on the kernel's own dG,
    relative to the entry's sum of |terms|: with one sequence 10^6 above all others, or sequences spread over six
    decades, the fp16 planes stay within 3 x the exact bf16 planes' error + 2e-7."""
    b, l, d_in = 3000, 4, 1
    lstm = reference_lstm(d_in, 77)
    g = torch.Generator(device=DEV).manual_seed(9)
    x = torch.randn(b, l, d_in, device=DEV, generator=g) * 3
    h0 = torch.randn(b, 256, device=DEV, generator=g) * 0.5
    c0 = torch.randn(b, 256, device=DEV, generator=g)
    dhs = torch.randn(b, l, 256, device=DEV, generator=g) / (b * l)
    if case == "one_outlier_sequence":
        dhs[b // 3] *= 1e6
    elif case == "six_decades":
        dhs *= 10.0 ** torch.randint(-4, 3, (b, 1, 1), device=DEV, generator=g).float()
    hs, hn, cn, gates, cs = hip.lstm_forward(x, h0, c0, pack(lstm), save=True)
    packed = hip.lstm_rows_backward_pack(lstm.weight_hh_l0)
    run = lambda: hip.lstm_backward(x, h0, c0, hs, gates, cs, dhs, None, split=True, rows_packed=packed)  # noqa: E731
    f16 = run()
    monkeypatch.setenv("RL8_AMD_LSTM_WGRAD_PLANES", "bf16")
    exact = run()
    dg = hip.lstm_rows_backward(c0, gates, cs, dhs, packed).double().reshape(b, l, 1024)
    h_prev = torch.cat([h0[:, None], hs[:, :-1]], 1).double()
    want = torch.einsum("blj,bli->ji", dg, h_prev)
    size = torch.einsum("blj,bli->ji", dg.abs(), h_prev.abs()) + 1e-300
    err16 = float(((f16["w_hh"].double() - want).abs() / size).max())
    err_exact = float(((exact["w_hh"].double() - want).abs() / size).max())
    assert err16 <= 3 * err_exact + 2e-7, (case, err16, err_exact)


def test_lstm_rows_backward_repeats_bit_for_bit():
    c0, gates, cs, dhs, w_hh = _rows_backward_inputs(8192, 4, 5)
    packed = hip.lstm_rows_backward_pack(w_hh)
    first = hip.lstm_rows_backward(c0, gates, cs, dhs, packed)
    for trial in range(20):
        assert torch.equal(first, hip.lstm_rows_backward(c0, gates, cs, dhs, packed)), trial


def test_h0_planes_are_shared_only_on_request():
    """fused_lstm's cache of the initial hidden states' planes: off by default (a raw-pointer write into the same memory
    is invisible to the version counter: every pass must split what is there NOW), on between RecurrentAlgorithm's
    SHARE_H0_PLANES = True and clear_state_cache()."""
    from rl8_amd.nn import fused_lstm

    lstm = reference_lstm(1, 3)
    g = torch.Generator(device=DEV).manual_seed(0)
    x = torch.randn(200, 4, 1, device=DEV, generator=g)
    h0 = torch.rand(200, 256, device=DEV, generator=g) - 0.5
    c0 = torch.randn(200, 256, device=DEV, generator=g)

    def out():
        hs, _, _ = fused_lstm.lstm_forward(lstm, x, h0, c0)
        return hs.detach().clone()

    fused_lstm.clear_state_cache()
    first = out()
    h0_new = torch.rand(200, 256, device=DEV, generator=g) - 0.5
    torch.cuda.synchronize()
    # overwrite h0's memory behind torch's back (as a kernel of the library would)
    import ctypes as C
    C.cdll.LoadLibrary("libamdhip64.so").hipMemcpy(C.c_void_p(h0.data_ptr()), C.c_void_p(h0_new.data_ptr()), C.c_size_t(h0.numel() * 4), 3)
    second = out()
    assert not torch.equal(first, second) and fused_lstm._h0_cache == {}
    fused_lstm.SHARE_H0_PLANES = True
    try:
        third = out()
        assert torch.equal(second, third) and "entry" in fused_lstm._h0_cache
        assert torch.equal(out(), third)
    finally:
        fused_lstm.clear_state_cache()
    assert fused_lstm._h0_cache == {} and not fused_lstm.SHARE_H0_PLANES


@pytest.mark.parametrize("n,with_noise,deterministic", [(1, False, 0), (67, True, 0), (8192, False, 0), (10_001, False, 1), (10_001, True, 0)])
def test_rollout_step_with_the_heads_inside_is_the_three_launches(n, with_noise, deterministic):
    """rl8_rollout_step_dummy_heads_f32 (the recurrent rollout's per-timestep tail) against the launches it replaces --
    rl8_linear_heads_forward_f32 for the logits, for the value, then rl8_rollout_step_dummy_f32 -- bit for bit: actions,
    log-probabilities, values, rewards, next observations, env state, discounted-return recurrence."""
    g = torch.Generator(device=DEV).manual_seed(n)
    h = torch.randn(n, 256, device=DEV, generator=g)
    w_pol, b_pol = torch.randn(2, 256, device=DEV, generator=g) / 16, torch.randn(2, device=DEV, generator=g)
    w_vf, b_vf = torch.randn(1, 256, device=DEV, generator=g) / 16, torch.randn(1, device=DEV, generator=g)
    noise = torch.rand(n, 2, device=DEV, generator=g) + 0.01 if with_noise else None
    state0 = (torch.rand(n, device=DEV, generator=g) * 2 - 1) * 100
    rdr_t = torch.randn(n, device=DEV, generator=g)
    lib = hip.load()

    def outputs():
        return dict(state=state0.clone(), action=torch.zeros(n, dtype=torch.int64, device=DEV),
                    logp=torch.zeros(n, device=DEV), value=torch.zeros(n, device=DEV), reward=torch.zeros(n, device=DEV),
                    obs=torch.zeros(n, device=DEV), rdr=torch.zeros(n, device=DEV))

    a, b = outputs(), outputs()
    tail = lambda o: (hip._ptr(o["state"]), hip._ptr(o["action"]), hip._ptr(o["logp"]), hip._ptr(o["value"]),  # noqa: E731
                      hip._ptr(o["reward"]), hip._ptr(o["obs"]), hip._ptr(rdr_t), hip._ptr(o["rdr"]), 0.95, n, 1234, 7, 5,
                      deterministic, hip._stream())
    hip._check(lib.rl8_rollout_step_dummy_heads_f32(hip._ptr(h), hip._ptr(w_pol), hip._ptr(b_pol), hip._ptr(w_vf), hip._ptr(b_vf),
                                                    hip._ptr(noise), *tail(a)), "rl8_rollout_step_dummy_heads_f32")
    logits, value = hip.linear_heads_forward(h, w_pol, b_pol), hip.linear_heads_forward(h, w_vf, b_vf)
    hip._check(lib.rl8_rollout_step_dummy_f32(1, 0, hip._ptr(logits), None, hip._ptr(value), hip._ptr(noise), *tail(b)),
               "rl8_rollout_step_dummy_f32")
    for k in a:
        assert torch.equal(a[k], b[k]), k
    assert int(a["action"].min()) >= 0 and int(a["action"].max()) <= 1
