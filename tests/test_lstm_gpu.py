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
