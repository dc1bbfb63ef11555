"""N1, second generation: the tower kernels whose 256x256 products run as bf16-plane
MFMAs (exact 3-way split of both fp32 operands, six of nine plane products, fp32
accumulate: rl8_amd/csrc/mlp_split_kernels.hip). The claim under test is "fp32
accuracy": every result is compared with an fp64 evaluation of the same op AND
with torch's own fp32 path / the fp32-MFMA kernels at the same inputs, and must
be as close to fp64 as those are. Run-to-run bit equality is checked for every
kernel (fixed summation orders; it is also what exposes a synchronisation bug).
"""

import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from rl8_amd import hip  # noqa: E402

DEV = "cuda:0"


def _params(g, d_in, n_out):
    return {
        "w1": torch.randn(256, d_in, device=DEV, generator=g) * 0.5,
        "b1": torch.randn(256, device=DEV, generator=g) * 0.1,
        "w2": torch.randn(256, 256, device=DEV, generator=g) / 16,
        "b2": torch.randn(256, device=DEV, generator=g) * 0.1,
        "w3": torch.randn(n_out, 256, device=DEV, generator=g) / 16,
        "b3": torch.randn(n_out, device=DEV, generator=g),
    }


def _tower(x, p):
    h1 = torch.relu(x @ p["w1"].T + p["b1"])
    h2 = torch.relu(h1 @ p["w2"].T + p["b2"])
    return h2 @ p["w3"].T + p["b3"], h1, h2


def _rel(got, want):
    return float((got.double() - want).abs().max()) / (float(want.abs().max()) + 1e-12)


@pytest.mark.parametrize("d_in", [1, 2, 3, 4, 5, 6, 8])
@pytest.mark.parametrize("where", [0, 63, 64, 191, 255])
def test_forward_row_scale_sees_the_largest_layer_one_entry_wherever_it_sits(d_in, where):
    """The row factor of the fp16 planes comes from max |b1| and max_k |w1[k][i]| -- a workgroup reduction over the 256
    units (one per thread: lane exchange within a wave, LDS across the four waves).  One unit with a bias and weights a
    thousand times the others', at the first / last lane of a wave and of the workgroup: missed, its h1 would overflow
    the fp16 planes (inf / NaN in the output); found, the forward stays at fp32 accuracy."""
    g = torch.Generator(device=DEV).manual_seed(100 * d_in + where)
    m, n_out = 777, 2
    x = torch.randn(m, d_in, device=DEV, generator=g) * 20
    p = _params(g, d_in, n_out)
    p["b1"][where] = 300.0
    p["w1"][where] = 150.0 * torch.sign(torch.randn(d_in, device=DEV, generator=g))
    out = hip.mlp_tower_forward_split(x, p["w1"], p["b1"], hip.mlp_pack_w2_f16(p["w2"]), p["b2"], p["w3"], p["b3"])[0]
    want = _tower(x.double(), {k: v.double() for k, v in p.items()})[0]
    assert torch.isfinite(out).all()
    assert _rel(out, want) < 5e-6


@pytest.mark.parametrize("m,d_in,n_out", [(1, 1, 2), (127, 1, 1), (128, 1, 2), (129, 2, 3), (1000, 5, 3), (4097, 1, 3),
                                          (5000, 3, 2), (70_001, 1, 2), (33_333, 5, 1), (20_000, 2, 2),
                                          # round 5: run-time widths inside the compiled classes (d_in <= 8, n_out <= 8; d_in <= 16, n_out <= 4 below)
                                          (1000, 4, 4), (5000, 4, 1), (3001, 6, 5), (777, 7, 3), (2000, 8, 8), (129, 8, 2),
                                          (4100, 2, 7), (33_000, 5, 6), (1, 8, 8),
                                          # class 16 (two chained layer-1 products): d_in 9..16, n_out <= 4
                                          (1000, 9, 1), (4097, 12, 2), (33_000, 16, 4), (129, 13, 3), (1, 16, 1), (70_001, 10, 2)])
@pytest.mark.parametrize("scheme", ["f16x2"])
def test_forward_split_is_fp32_accurate(m, d_in, n_out, scheme):
    """Both generations of the plane-product forward against the SAME bars: six
    bf16 plane products per 16 k ("bf16x3") and three scaled fp16 ones ("f16x2")."""
    assert hip.mlp_forward_f16_supports(d_in, n_out)
    g = torch.Generator(device=DEV).manual_seed(m + d_in)
    x = torch.randn(m, d_in, device=DEV, generator=g) * 30
    p = _params(g, d_in, n_out)
    want, h1w, h2w = _tower(x.double(), {k: v.double() for k, v in p.items()})
    packed = hip.mlp_pack_w2_f16(p["w2"])
    out, h1, h2 = hip.mlp_tower_forward_split(x, p["w1"], p["b1"], packed, p["b2"], p["w3"], p["b3"], save=True)
    assert _rel(out, want) < 4e-6 and _rel(h2, h2w) < 2e-6 and _rel(h1, h1w) < 1e-6
    # d_in <= 3: the same layer-1 fma chain as the fp32-MFMA kernel, h1 bit for bit.  d_in = 4..8 (class 8, round 5):
    # layer 1 on the matrix pipe -- all four fp16 plane products of x and W1 in one MFMA -- as close to fp64 as that chain
    out32, h1_32, h2_32 = hip.mlp_tower_forward(x, p["w1"], p["b1"], hip.mlp_pack_w2(p["w2"]), p["b2"], p["w3"], p["b3"],
                                                save=True)
    if d_in <= 3:
        assert torch.equal(h1, h1_32)
    else:
        scale1 = float(h1w.abs().max())
        assert float((h1.double() - h1w).abs().max()) <= 2 * float((h1_32.double() - h1w).abs().max()) + 2e-7 * scale1
        # ... and its ReLU gate is the fma chain's wherever the pre-activation is not within rounding of zero
        z1 = x.double() @ p["w1"].double().T + p["b1"].double()
        bound = (x.double().abs() @ p["w1"].double().abs().T + p["b1"].double().abs()) * 2.0 ** -20
        clear = z1.abs() > bound
        assert bool(((h1 > 0) == (z1 > 0))[clear].all()) and float(clear.double().mean()) > 0.999
    # as close to fp64 as the fp32-MFMA kernel and as torch's fp32 path
    ref32, _, h2_t = _tower(x, p)
    scale = float(want.abs().max()) + 1e-6
    err = float((out.double() - want).abs().max())
    assert err <= 3 * float((out32.double() - want).abs().max()) + 1e-6 * scale
    assert err <= 4 * float((ref32.double() - want).abs().max()) + 1e-6 * scale
    assert float((h2.double() - h2w).abs().max()) <= 3 * float((h2_32.double() - h2w).abs().max()) + 1e-6 * float(h2w.abs().max())
    # inference variant and run-to-run bit equality
    out2, n1, n2 = hip.mlp_tower_forward_split(x, p["w1"], p["b1"], packed, p["b2"], p["w3"], p["b3"])
    assert n1 is None and n2 is None and torch.equal(out, out2)
    out3, h1b, h2b = hip.mlp_tower_forward_split(x, p["w1"], p["b1"], packed, p["b2"], p["w3"], p["b3"], save=True)
    assert torch.equal(out, out3) and torch.equal(h1, h1b) and torch.equal(h2, h2b)
    # h1 is optional (the bf16-plane backward recomputes it)
    out4, h1c, h2c = hip.mlp_tower_forward_split(x, p["w1"], p["b1"], packed, p["b2"], p["w3"], p["b3"], save=True,
                                                 save_h1=False)
    assert h1c is None and torch.equal(out, out4) and torch.equal(h2, h2c)
    # the ReLU gate of h2 as bits: bit j of row s = h2[s, j] > 0
    out5, _, h2d, gate = hip.mlp_tower_forward_split(x, p["w1"], p["b1"], packed, p["b2"], p["w3"], p["b3"], save=True,
                                                     save_gate=True)
    want_bits = (h2d > 0).view(m, 8, 32).to(torch.int64)
    want_words = (want_bits << torch.arange(32, device=DEV)).sum(-1)
    assert torch.equal(out, out5) and torch.equal(gate.to(torch.int64) & 0xFFFFFFFF, want_words)


def test_f16_planes_reconstruct_the_scaled_weights_to_22_bits():
    g = torch.Generator(device=DEV).manual_seed(5)
    w = torch.randn(256, 256, device=DEV, generator=g) / 16
    w[3, 7] = 0.0
    for transposed in (False, True):
        packed = hip.mlp_pack_w2_f16(w, transposed=transposed)
        scale, inv = packed[-16:].view(torch.float32)[:2].tolist()
        top = float(w.abs().max()) * scale
        assert scale * inv == 1.0 and 2.0**13 <= top < 2.0**14   # a power of two placing max |w| below 2^14
        planes = packed[:-16].view(torch.float16).view(16, 8, 2, 64, 8).double().sum(2)  # [step][col tile][lane][e]
        lane = torch.arange(64, device=DEV)
        step, tile, e = torch.arange(16, device=DEV), torch.arange(8, device=DEV), torch.arange(8, device=DEV)
        if transposed:  # the data-gradient kernel's operand: 32x32x16 fragments, unit [k step][column tile][plane][lane]
            col = (32 * tile[None, :, None, None] + (lane & 31)[None, None, :, None]).expand(16, 8, 64, 8)
            k = (16 * step[:, None, None, None] + 8 * (lane >> 5)[None, None, :, None] + e[None, None, None, :]).expand(16, 8, 64, 8)
        else:           # the forward kernel's: 16x16x32 fragments, unit [half-step = 2 k-block + column half][local tile][plane][lane]
            col = (16 * (8 * (step & 1)[:, None, None, None] + tile[None, :, None, None])
                   + (lane & 15)[None, None, :, None]).expand(16, 8, 64, 8)
            k = (32 * (step >> 1)[:, None, None, None] + 8 * (lane >> 4)[None, None, :, None] + e[None, None, None, :]).expand(16, 8, 64, 8)
        want = (w[k, col] if transposed else w[col, k]).double() * scale
        # hi = fp16(v), lo = fp16(v - hi): 22 significand bits, or the fp16 subnormal quantum
        assert bool(((planes - want).abs() <= want.abs() * 2.0**-22 + 2.0**-25).all())


@pytest.mark.parametrize("case", ["rows_of_mixed_magnitude", "outlier_weights", "zero_rows", "tiny_everything", "huge_inputs"])
def test_forward_f16_scaling_holds_over_the_dynamic_range(case):
    """The per-row / per-matrix powers of two are what keeps the fp16 planes in
    range: rows 10^6 apart in one tile, a W2 with a few entries 1000x the rest,
    all-zero rows and weights, and everything near the fp16 underflow / overflow."""
    m, d_in, n_out = 3000, 3, 2
    g = torch.Generator(device=DEV).manual_seed(17)
    x = torch.randn(m, d_in, device=DEV, generator=g)
    p = _params(g, d_in, n_out)
    if case == "rows_of_mixed_magnitude":
        x *= 10.0 ** torch.randint(-3, 4, (m, 1), device=DEV, generator=g).float()
    elif case == "outlier_weights":
        p["w2"][torch.randint(0, 256, (20,), device=DEV, generator=g), torch.randint(0, 256, (20,), device=DEV, generator=g)] = 60.0
    elif case == "zero_rows":
        x[::3] = 0.0
        p["b1"].zero_()
        p["w2"][:, ::2] = 0.0
    elif case == "tiny_everything":
        x *= 1e-12
        p["b1"] *= 1e-12
        p["w2"] *= 1e-9
    elif case == "huge_inputs":
        x *= 1e12
        p["w2"] *= 1e6
    want, h1w, h2w = _tower(x.double(), {k: v.double() for k, v in p.items()})
    out, _, h2 = hip.mlp_tower_forward_split(x, p["w1"], p["b1"], hip.mlp_pack_w2_f16(p["w2"]), p["b2"], p["w3"], p["b3"],
                                             save=True)
    # yardstick: the fp32-MFMA kernel (the six-product bf16-plane forward of rounds 1-2 is gone)
    out6, _, h2_6 = hip.mlp_tower_forward(x, p["w1"], p["b1"], hip.mlp_pack_w2(p["w2"]), p["b2"], p["w3"], p["b3"], save=True)
    assert bool(torch.isfinite(out).all()) and bool(torch.isfinite(h2).all())
    # per ROW against that row's own magnitude (a global relative error would hide the small rows)
    row = h2w.abs().amax(1, keepdim=True) + p["b2"].abs().max().double()
    err, err6 = ((h2.double() - h2w).abs() / row).max(), ((h2_6.double() - h2w).abs() / row).max()
    assert float(err) <= 2e-6 and float(err) <= 3 * float(err6) + 1e-7, (float(err), float(err6))
    orow = want.abs().amax(1, keepdim=True) + (h2w.abs() @ p["w3"].double().abs().T).amax(1, keepdim=True) + 1e-30
    oerr, oerr6 = ((out.double() - want).abs() / orow).max(), ((out6.double() - want).abs() / orow).max()
    assert float(oerr) <= 2e-6 and float(oerr) <= 3 * float(oerr6) + 1e-7, (float(oerr), float(oerr6))


@pytest.mark.parametrize("m,d_in,n_out", [(1, 1, 2), (129, 1, 1), (1000, 5, 3), (4097, 3, 2), (40_000, 1, 2), (9000, 2, 1),
                                          (33_000, 1, 2), (33_000, 5, 3), (20_000, 3, 1), (16_500, 2, 3),
                                          # round 5: four-wide observations, four-way heads
                                          (1000, 4, 4), (4097, 4, 1), (9000, 5, 4), (3000, 2, 4), (33_000, 4, 3),
                                          # round 6: six and seven observations (class 8 data gradients; the weight
                                          # gradients' scalar loads made in front of their wait)
                                          (1000, 6, 3), (4097, 7, 3), (9000, 6, 4), (3000, 7, 1), (33_000, 7, 2),
                                          (129, 6, 1), (20_000, 6, 2), (1, 7, 3)])
@pytest.mark.parametrize("scheme", ["f16x2"])
def test_backward_split_matches_fp64(m, d_in, n_out, scheme):
    """Against an fp64 evaluation of the backward formulas on the SAVED activations
    (the ReLU gates are part of the input of a backward pass: an autograd run in
    fp64 flips the gates of pre-activations within fp32 rounding of zero, which
    says nothing about these kernels).  Both plane schemes against the same bars."""
    assert hip.mlp_backward_f16_supports(d_in, n_out)
    g = torch.Generator(device=DEV).manual_seed(7 * m + d_in)
    x = torch.randn(m, d_in, device=DEV, generator=g) * 3
    p = _params(g, d_in, n_out)
    dout = torch.randn(m, n_out, device=DEV, generator=g) / m
    out, h1, h2, gate = hip.mlp_tower_forward_split(x, p["w1"], p["b1"], hip.mlp_pack_w2_f16(p["w2"]), p["b2"], p["w3"],
                                                    p["b3"], save=True, save_gate=True)
    d, a1, a2 = dout.double(), h1.double(), h2.double()
    dz2 = (d @ p["w3"].double()) * (a2 > 0)
    dz1 = (dz2 @ p["w2"].double()) * (a1 > 0)
    want = {"w1": dz1.T @ x.double(), "b1": dz1.sum(0), "w2": dz2.T @ a1, "b2": dz2.sum(0), "w3": d.T @ a2, "b3": d.sum(0)}
    f16 = scheme == "f16x2"
    w2t = hip.mlp_pack_w2_f16(p["w2"], transposed=True)
    grads = hip.mlp_tower_backward(x, h1, h2, dout, w2t, p["w3"], p["w1"], p["b1"], gate2=gate if f16 else None)
    grads32 = hip.mlp_tower_backward(x, h1, h2, dout, hip.mlp_pack_w2(p["w2"], transposed=True), p["w3"])
    for k in p:
        err, err32 = _rel(grads[k], want[k]), _rel(grads32[k], want[k])
        if k == "b3":  # a sum of +- terms that may nearly cancel: measure against sum |terms|
            err = float((grads[k].double() - want[k]).abs().max()) / float(d.abs().sum(0).max())
            assert err < 1e-6, (k, err)
            continue
        assert err < 2e-5, (k, err)
        assert err <= 8 * err32 + 2e-6, (k, err, err32)  # as accurate as the fp32-MFMA kernels
    again = hip.mlp_tower_backward(x, None, h2, dout, w2t, p["w3"], p["w1"], p["b1"], gate2=gate if f16 else None)  # h1 is not read
    for k in p:
        assert torch.equal(grads[k], again[k]), k  # fixed summation order, no race
    if f16:  # (fused mode only: the gate bits are required)
        with pytest.raises(ValueError):
            hip.mlp_tower_backward(x, None, h2, dout, w2t, p["w3"], p["w1"], p["b1"])
        return
    # gate bits instead of h2 reads in the data-gradient kernel: the same decisions, so the same bits out
    bits = hip.mlp_tower_backward(x, None, h2, dout, w2t, p["w3"], p["w1"], p["b1"], gate2=gate)
    for k in p:
        assert torch.equal(grads[k], bits[k]), k
    with pytest.raises(ValueError):
        hip.mlp_tower_backward(x, None, h2, dout, hip.mlp_pack_w2(p["w2"], transposed=True), p["w3"])


@pytest.mark.parametrize("case", ["rows_of_mixed_magnitude", "one_outlier_row", "clipped_rows", "tiny_gradients",
                                  "outlier_weights"])
@pytest.mark.parametrize("n_out", [2, 1])
def test_backward_f16_scaling_holds_over_the_dynamic_range(case, n_out):
    """dOut as PPO produces it is heavy-tailed (clipped samples contribute exactly zero, a few
    samples carry most of the gradient): the data-gradient kernel scales per ROW, the
    weight-gradient kernel per output COLUMN (its sum runs over the rows).  Every gradient against
    fp64 on the saved activations, entry by entry relative to the sum of the magnitudes of the
    entry's terms, beside the fp32-MFMA kernels' own error.  dW2 is the one place where the fp16 planes
    give up something against fp32 products: a term 2^17 below its column's bound no longer carries
    22 bits, so an entry made only of small rows keeps ~1e-5 of its own size (round 3, unguarded: 1.6e-5 /
    4.7e-5 with rows six decades apart; fp32 MFMAs 6e-7 / 2e-6).  Since round 4 a guard looks at the spread of
    |dOut| in every call (wgrad_tail_kernel) and sends such calls to the exact bf16 planes."""
    m, d_in = 20_000, 3
    g = torch.Generator(device=DEV).manual_seed(23)
    x = torch.randn(m, d_in, device=DEV, generator=g) * 3
    p = _params(g, d_in, n_out)
    dout = torch.randn(m, n_out, device=DEV, generator=g) / m
    if case == "rows_of_mixed_magnitude":
        dout *= 10.0 ** torch.randint(-4, 3, (m, 1), device=DEV, generator=g).float()
        x *= 10.0 ** torch.randint(-2, 2, (m, 1), device=DEV, generator=g).float()
    elif case == "one_outlier_row":
        dout[777] *= 1e6
    elif case == "clipped_rows":
        dout[torch.rand(m, device=DEV, generator=g) < 0.7] = 0.0
    elif case == "tiny_gradients":
        dout *= 1e-20
    elif case == "outlier_weights":
        p["w2"][torch.randint(0, 256, (20,), device=DEV, generator=g), torch.randint(0, 256, (20,), device=DEV, generator=g)] = 40.0
        p["w3"][0, 5] = 30.0
    _, h1, h2, gate = hip.mlp_tower_forward_split(x, p["w1"], p["b1"], hip.mlp_pack_w2_f16(p["w2"]), p["b2"], p["w3"],
                                                  p["b3"], save=True, save_gate=True)
    d, a1, a2 = dout.double(), h1.double(), h2.double()
    dz2 = (d @ p["w3"].double()) * (a2 > 0)
    dz1 = (dz2 @ p["w2"].double()) * (a1 > 0)
    want = {"w1": dz1.T @ x.double(), "b1": dz1.sum(0), "w2": dz2.T @ a1, "b2": dz2.sum(0), "w3": d.T @ a2}
    # the scale of a gradient entry: the sum of the magnitudes of its terms (what an fp32 sum is accurate against)
    size = {"w1": dz1.abs().T @ x.double().abs(), "b1": dz1.abs().sum(0), "w2": dz2.abs().T @ a1, "b2": dz2.abs().sum(0),
            "w3": d.abs().T @ a2}
    consulted0, fired0 = hip.wgrad_guard_counts()
    got = hip.mlp_tower_backward(x, None, h2, dout, hip.mlp_pack_w2_f16(p["w2"], transposed=True), p["w3"], p["w1"], p["b1"],
                                 gate2=gate)
    consulted, fired = (a - b for a, b in zip(hip.wgrad_guard_counts(), (consulted0, fired0)))
    # Round 4: the guard.  dOut with rows decades apart goes to the exact bf16 planes by itself (device-side flag);
    # what it lets through the fp16 planes holds the fp32 kernels' bar ENTRY BY ENTRY again.  (One output with h2
    # given runs the gate kernel on exact bf16 planes: nothing to guard.)
    if n_out == 2:
        assert consulted == 1 and fired == int(case == "one_outlier_row"), (case, consulted, fired)
    else:
        assert consulted == 0
    # yardstick: the fp32-MFMA generation on the same saved activations
    got6 = hip.mlp_tower_backward(x, h1, h2, dout, hip.mlp_pack_w2(p["w2"], transposed=True), p["w3"])
    for k in want:
        assert bool(torch.isfinite(got[k]).all()), k
        # dW2's planes resolve 2^-50 of a column's bound (wide low plane): entries whose whole sum of |terms| lies 2^40 below
        # the largest entry's -- a single row near its ReLU kink, tens of binades down -- are measured against that floor
        floor = size[k].max() * (2.0 ** -40 if k == "w2" else 1e-30) + 1e-300
        err = float(((got[k].double() - want[k]).abs() / (size[k] + floor)).max())
        err6 = float(((got6[k].double() - want[k]).abs() / (size[k] + floor)).max())
        # (the absolute bar is the fp32 accumulators' own: with one row 10^6 above the rest both generations sit at 5e-6 .. 3e-5)
        assert err < 1e-4, (k, err, err6)
        assert err <= 3 * err6 + 2e-7, (case, k, err, err6)


@pytest.mark.parametrize("case", ["plain", "rows_of_mixed_magnitude", "one_outlier_row", "clipped_rows", "outlier_weights"])
@pytest.mark.parametrize("m,d_in,n_out", [(1, 1, 1), (129, 1, 2), (1000, 5, 1), (4097, 3, 2), (40_000, 1, 1), (9000, 2, 2),
                                          (20_000, 3, 1), (33_000, 5, 2), (1000, 4, 1), (4097, 4, 2), (3000, 6, 1), (5000, 7, 2)])
def test_gate_mode_data_gradient(m, d_in, n_out, case):
    """Heads whose dZ2 is gate * d[s] * w3e[k] (one output; two outputs with exactly opposite gradients):
    the data-gradient kernel takes the ReLU gate itself as its A operand and the planes of w3e[k] W2[k][i] as B.
    dW1 / db1 (what that kernel produces) against fp64 on the saved activations and against the general
    fp16-plane kernel's own error, over the dynamic-range cases of the general kernel's test."""
    g = torch.Generator(device=DEV).manual_seed(13 * m + d_in + n_out)
    x = torch.randn(m, d_in, device=DEV, generator=g) * 3
    p = _params(g, d_in, n_out)
    g0 = torch.randn(m, device=DEV, generator=g) / m
    if case == "rows_of_mixed_magnitude":
        g0 *= 10.0 ** torch.randint(-4, 3, (m,), device=DEV, generator=g).float()
        x *= 10.0 ** torch.randint(-2, 2, (m, 1), device=DEV, generator=g).float()
    elif case == "one_outlier_row":
        g0[m // 3] *= 1e6
    elif case == "clipped_rows":
        g0[torch.rand(m, device=DEV, generator=g) < 0.7] = 0.0
    elif case == "outlier_weights":
        p["w2"][torch.randint(0, 256, (20,), device=DEV, generator=g), torch.randint(0, 256, (20,), device=DEV, generator=g)] = 40.0
        p["w3"][0, 5] = 30.0
    dout = (torch.stack([g0, -g0], 1) if n_out == 2 else g0[:, None]).contiguous()
    # (h1 as the forward kernel formed it: the backward recomputes ITS pre-activations, bit for bit -- at d_in >= 4 on the
    # matrix pipe -- so the two never disagree on a gate; torch's own fp32 product may, where z1 rounds to about zero)
    _, h1, h2, gate = hip.mlp_tower_forward_split(x, p["w1"], p["b1"], hip.mlp_pack_w2_f16(p["w2"]), p["b2"], p["w3"], p["b3"],
                                                  save=True, save_gate=True)
    d, a1, a2 = dout.double(), h1.double(), h2.double()
    dz2 = (d @ p["w3"].double()) * (a2 > 0)
    dz1 = (dz2 @ p["w2"].double()) * (a1 > 0)
    want = {"w1": dz1.T @ x.double(), "b1": dz1.sum(0), "w2": dz2.T @ a1, "b2": dz2.sum(0), "w3": d.T @ a2}
    # the yardstick of dW1 / db1: the sum of the magnitudes of ALL terms, those of the inner products
    # dH1[s][i] = sum_k dZ2[s][k] W2[k][i] included (an inner product that cancels is not the kernel's error;
    # profiles/r02_gate_dgrad_inner_error.txt has the three kernels on it: 1.3e-7 gate, 2.4e-7 general, 3.4e-7 bf16)
    inner = (dz2.abs() @ p["w2"].double().abs()) * (a1 > 0)
    size = {"w1": inner.T @ x.double().abs(), "b1": inner.sum(0)}
    w2t = hip.mlp_pack_w2_f16(p["w2"], transposed=True)

    def run(gate_pack):
        hip.timer.reset()
        hip.timer.enabled = True
        try:
            grads = hip.mlp_tower_backward(x, None, h2, dout, w2t, p["w3"], p["w1"], p["b1"], gate2=gate, gate_pack=gate_pack)
            return grads, set(hip.timer.summary())
        finally:
            hip.timer.enabled = False

    got, launched = run(lambda: hip.mlp_pack_w2_f16_gate(p["w2"], p["w3"]))
    ref, launched_ref = run(None)
    assert "mlp_tower_backward_gate" in launched and "mlp_tower_backward" in launched_ref
    for k in want:
        assert bool(torch.isfinite(got[k]).all()), k
        assert _rel(got[k], want[k]) < 2e-5, k
        if k not in ("w1", "b1"):  # (the weight-gradient kernel's outputs: the same kernel in both runs)
            assert torch.equal(got[k], ref[k]), k
            continue
        floor = size[k].max() * 1e-30 + 1e-300
        err = float(((got[k].double() - want[k]).abs() / (size[k] + floor)).max())
        err3 = float(((ref[k].double() - want[k]).abs() / (size[k] + floor)).max())
        assert err < 2e-6 and err <= 3 * err3 + 2e-7, (k, err, err3)
    again, _ = run(lambda: hip.mlp_pack_w2_f16_gate(p["w2"], p["w3"]))
    for k in got:
        assert torch.equal(got[k], again[k]), k


@pytest.mark.parametrize("x_scale", [3.0, 100.0])
@pytest.mark.parametrize("m,d_in,n_out", [(1, 1, 1), (129, 1, 2), (1000, 5, 1), (4097, 3, 2), (40_000, 1, 2), (9000, 2, 1),
                                          (100_000, 1, 1), (1000, 4, 1), (4097, 4, 2), (3000, 6, 2), (5000, 7, 1), (20_000, 7, 2)])
def test_rank_one_backward_from_the_gate_bits_alone(m, d_in, n_out, x_scale):
    """A forward that keeps ONLY the gate bits of h2 (32 bytes per row) and a backward that never sees h2: the gate
    modes of the data and weight gradients, with dW3 = sum_i W2[.][i] M[.][i] + b2 sum_s G dOut taken from the sums M
    the weight-gradient kernel forms anyway.  Every gradient against fp64 on the activations a full forward saves,
    and against the h2-reading kernels on the same inputs; dW3 -- the reformulated one -- differs from the direct sum
    by the rounding the forward's own dot products put into h2 (bar: the direct kernel's error x 10 + 2e-6 of the
    largest entry; observations up to +-300 as the dummy envs produce them)."""
    g = torch.Generator(device=DEV).manual_seed(17 * m + d_in + n_out)
    x = torch.randn(m, d_in, device=DEV, generator=g) * x_scale
    p = _params(g, d_in, n_out)
    g0 = torch.randn(m, device=DEV, generator=g) / m
    g0[torch.rand(m, device=DEV, generator=g) < 0.3] = 0.0
    dout = (torch.stack([g0, -g0], 1) if n_out == 2 else g0[:, None]).contiguous()
    w2p, w2t = hip.mlp_pack_w2_f16(p["w2"]), hip.mlp_pack_w2_f16(p["w2"], transposed=True)
    args = (x, p["w1"], p["b1"], w2p, p["b2"], p["w3"], p["b3"])
    out, _, h2, gate = hip.mlp_tower_forward_split(*args, save=True, save_h1=False, save_gate=True)
    out_b, h1_b, h2_b, gate_b = hip.mlp_tower_forward_split(*args, save=True, save_gate=True, save_h2=False)
    assert h1_b is None and h2_b is None and torch.equal(out_b, out) and torch.equal(gate_b, gate)
    h1 = hip.mlp_tower_forward_split(*args, save=True)[1]  # (the forward's own: see test_gate_mode_data_gradient)
    d, a1, a2 = dout.double(), h1.double(), h2.double()
    dz2 = (d @ p["w3"].double()) * (a2 > 0)
    dz1 = (dz2 @ p["w2"].double()) * (a1 > 0)
    want = {"w1": dz1.T @ x.double(), "b1": dz1.sum(0), "w2": dz2.T @ a1, "b2": dz2.sum(0), "w3": d.T @ a2, "b3": d.sum(0)}
    gate_pack = lambda: hip.mlp_pack_w2_f16_gate(p["w2"], p["w3"])  # noqa: E731
    info = {}
    got = hip.mlp_tower_backward(x, None, None, dout, w2t, p["w3"], p["w1"], p["b1"], gate2=gate, gate_pack=gate_pack,
                                 w2=p["w2"], b2=p["b2"], info=info)
    ref = hip.mlp_tower_backward(x, None, h2, dout, w2t, p["w3"], p["w1"], p["b1"], gate2=gate, gate_pack=gate_pack)
    assert info["rank_one"]
    for k in want:
        if k == "b3":
            continue
        err, err_ref = _rel(got[k], want[k]), _rel(ref[k], want[k])
        assert err < 2e-5, (k, err)
        if k == "w3":
            assert err <= 10 * err_ref + 2e-6, (k, err, err_ref)
        else:  # the same kernels on the same operands (dW2: same sums, W3 applied behind the slab reduction instead of per slab)
            assert err <= 3 * err_ref + 2e-6, (k, err, err_ref)
    if n_out == 2:
        assert torch.equal(got["w3"][1], -got["w3"][0])
    again = hip.mlp_tower_backward(x, None, None, dout, w2t, p["w3"], p["w1"], p["b1"], gate2=gate, gate_pack=gate_pack,
                                   w2=p["w2"], b2=p["b2"])
    for k in got:
        assert torch.equal(got[k], again[k]), k
    if n_out == 2:  # not a pair after all: h2 has to be supplied, and the general kernels run
        broken = dout.clone()
        broken[m // 2, 1] += 1e-3 / m
        with pytest.raises(ValueError):
            hip.mlp_tower_backward(x, None, None, broken, w2t, p["w3"], p["w1"], p["b1"], gate2=gate, gate_pack=gate_pack,
                                   w2=p["w2"], b2=p["b2"])
        calls = []
        late = hip.mlp_tower_backward(x, None, None, broken, w2t, p["w3"], p["w1"], p["b1"], gate2=gate, gate_pack=gate_pack,
                                      w2=p["w2"], b2=p["b2"], h2_fn=lambda: calls.append(1) or h2, info=info)
        general = hip.mlp_tower_backward(x, None, h2, broken, w2t, p["w3"], p["w1"], p["b1"], gate2=gate)
        assert calls == [1] and not info["rank_one"]
        for k in late:
            assert torch.equal(late[k], general[k]), k


@pytest.mark.parametrize("n_out", [1, 2])
@pytest.mark.parametrize("case,fires", [("plain", False), ("clipped_rows", False), ("rows_of_mixed_magnitude", False),
                                         ("one_outlier_row", True), ("many_small_rows", False), ("two_populations", True),
                                         # round 5: calls small enough to be sampled whole -- the maximum is
                                         # then IN the sample, and must not count towards the mean it is compared with
                                         ("small_outlier", True), ("small_plain", False), ("small_all_equal", False)])
def test_guard_picks_the_planes_of_the_gate_bits_weight_gradient_from_the_data(case, fires, n_out, monkeypatch):
    """rl8_mlp_wgrad_gate_bits_f32 (the headline's weight gradient) under its guard: the call's dOut is sampled on the
    device; when its largest entry stands more than 2^12 above the mean of the non-zero ones (one outlier row; a few
    rows that dwarf all others) the sums are formed on the exact bf16 planes -- bit for bit what
    RL8_WGRAD_GATE_PLANES=bf16 gives -- otherwise on the two fp16 planes, bit for bit the unguarded kernel; no host
    round trip either way, and the lifetime counters say which way it went.  A long TAIL of small rows (PPO's
    converged policies: a fifth of the rows 2^-12 below the largest, profiles/r04_wgrad_planes_real_ppo.json) does not
    trip it: the wide low plane keeps 22 bits of a term down to 2^-27 of its column's bound.  Entry by entry against
    fp64, relative to the entry's own sum of |terms|: what the guard lets through stays within 3x the exact planes'
    error + 2e-7."""
    m, d_in = (3000 if case.startswith("small") else 150_000), 1
    g = torch.Generator(device=DEV).manual_seed(3 + n_out)
    x = torch.randn(m, d_in, device=DEV, generator=g) * 40
    p = _params(g, d_in, n_out)
    g0 = torch.randn(m, device=DEV, generator=g) / m
    if case == "small_outlier":
        g0[m // 3] *= 1e6
    elif case == "small_all_equal":
        g0 = torch.full((m,), 1.0 / m, device=DEV) * torch.sign(g0)
    elif case == "clipped_rows":
        g0[torch.rand(m, device=DEV, generator=g) < 0.7] = 0.0
    elif case == "rows_of_mixed_magnitude":
        g0 *= 10.0 ** torch.randint(-4, 3, (m,), device=DEV, generator=g).float()
    elif case == "one_outlier_row":
        g0[m // 3] *= 1e6
    elif case == "many_small_rows":    # a third of the rows 2^-20 down: a long tail, the mean hardly moves
        g0[torch.rand(m, device=DEV, generator=g) < 0.33] *= 2.0 ** -20
    elif case == "two_populations":    # one row in ten thousand 2^20 above all others: the bound belongs to them
        g0[torch.rand(m, device=DEV, generator=g) < 0.0001] *= 2.0 ** 20
    dout = (torch.stack([g0, -g0], 1) if n_out == 2 else g0[:, None]).contiguous()
    w2p, w2t = hip.mlp_pack_w2_f16(p["w2"]), hip.mlp_pack_w2_f16(p["w2"], transposed=True)
    _, h1, h2, gate = hip.mlp_tower_forward_split(x, p["w1"], p["b1"], w2p, p["b2"], p["w3"], p["b3"], save=True,
                                                  save_gate=True)  # (h1 as the kernels form it: one fp32 fma chain)
    gate_pack = lambda: hip.mlp_pack_w2_f16_gate(p["w2"], p["w3"])  # noqa: E731

    def backward(mode):
        if mode is None:
            monkeypatch.delenv("RL8_WGRAD_GATE_PLANES", raising=False)
        else:
            monkeypatch.setenv("RL8_WGRAD_GATE_PLANES", mode)
        before = hip.wgrad_guard_counts()
        out = hip.mlp_tower_backward(x, None, None, dout, w2t, p["w3"], p["w1"], p["b1"], gate2=gate, gate_pack=gate_pack,
                                     w2=p["w2"], b2=p["b2"])
        return out, tuple(a - b for a, b in zip(hip.wgrad_guard_counts(), before))

    shipped, counts = backward(None)
    unguarded, counts_f16 = backward("f16!")
    exact, counts_bf16 = backward("bf16")
    assert counts == (1, int(fires)) and counts_f16 == (0, 0) and counts_bf16 == (0, 0)
    for k in shipped:
        assert torch.equal(shipped[k], (exact if fires else unguarded)[k]), (case, k)
    dz2 = (dout.double() @ p["w3"].double()) * (h2 > 0)
    want, size = dz2.T @ h1.double(), dz2.abs().T @ h1.double()
    # (entries whose whole sum of |terms| lies 2^40 below the largest entry's -- one row near its ReLU kink, tens of
    # binades down -- are measured against that floor: the planes resolve 2^-50 of a column's bound)
    floor = size.max() * 2.0 ** -40 + 1e-300
    err, err_exact = (float(((t["w2"].double() - want).abs() / (size + floor)).max()) for t in (shipped, exact))
    assert err <= 3 * err_exact + 2e-7, (case, err, err_exact)


@pytest.mark.parametrize("m,d_in", [(1, 1), (129, 1), (1000, 5), (4097, 3), (40_000, 1), (9000, 2), (3000, 6), (5000, 7)])
def test_pair_weight_gradient_of_a_two_way_head(m, d_in, monkeypatch):
    """dOut[s][1] == -dOut[s][0] exactly (what the categorical loss kernel emits for two actions):
    dZ2 = gate * dOut[s][0] * (W3[0] - W3[1]), so the weight gradient runs with the gate as its
    (binary, exact) first operand.  Chosen on the DATA (rl8_mlp_dout_pair_check); against fp64, against
    the six-product kernel on the same inputs, and not chosen when one row breaks the property."""
    g = torch.Generator(device=DEV).manual_seed(31 * m + d_in)
    x = torch.randn(m, d_in, device=DEV, generator=g) * 3
    p = _params(g, d_in, 2)
    g0 = torch.randn(m, device=DEV, generator=g) / m
    g0[torch.rand(m, device=DEV, generator=g) < 0.3] = 0.0   # clipped samples: exact zeros (of either sign below)
    dout = torch.stack([g0, -g0], 1).contiguous()
    # (h1 as the forward kernel formed it: the backward recomputes ITS pre-activations, bit for bit -- at d_in >= 4 on the
    # matrix pipe -- so the two never disagree on a gate; torch's own fp32 product may, where z1 rounds to about zero)
    _, h1, h2, gate = hip.mlp_tower_forward_split(x, p["w1"], p["b1"], hip.mlp_pack_w2_f16(p["w2"]), p["b2"], p["w3"], p["b3"],
                                                  save=True, save_gate=True)
    d, a1, a2 = dout.double(), h1.double(), h2.double()
    dz2 = (d @ p["w3"].double()) * (a2 > 0)
    dz1 = (dz2 @ p["w2"].double()) * (a1 > 0)
    want = {"w1": dz1.T @ x.double(), "b1": dz1.sum(0), "w2": dz2.T @ a1, "b2": dz2.sum(0), "w3": d.T @ a2}
    w2t = hip.mlp_pack_w2_f16(p["w2"], transposed=True)

    def run(data):
        hip.timer.reset()
        hip.timer.enabled = True
        try:
            grads = hip.mlp_tower_backward(x, None, h2, data, w2t, p["w3"], p["w1"], p["b1"], gate2=gate)
            return grads, set(hip.timer.summary())
        finally:
            hip.timer.enabled = False

    got, launched = run(dout)
    assert "mlp_wgrad_gate" in launched and "mlp_wgrad" not in launched
    monkeypatch.setenv("RL8_WGRAD_GATE_OFF", "1")
    ref6, launched6 = run(dout)
    monkeypatch.delenv("RL8_WGRAD_GATE_OFF")
    assert "mlp_wgrad" in launched6 and "mlp_wgrad_gate" not in launched6
    for k in want:
        err, err6 = _rel(got[k], want[k]), _rel(ref6[k], want[k])
        assert err < 2e-5 and err <= 3 * err6 + 2e-6, (k, err, err6)
    assert torch.equal(got["w3"][1], -got["w3"][0])
    again, _ = run(dout)
    for k in got:
        assert torch.equal(got[k], again[k]), k  # fixed summation order
    # one row off by an ulp: the property no longer holds, the general kernel runs
    broken = dout.clone()
    broken[m // 2, 1] = torch.nextafter(broken[m // 2, 1] + 1e-30, torch.tensor(1.0, device=DEV))
    _, launched_b = run(broken)
    assert "mlp_wgrad" in launched_b and "mlp_wgrad_gate" not in launched_b


@pytest.mark.parametrize("m,d_in,n_out", [(1, 1, 2), (13, 1, 1), (4099, 2, 3), (20_003, 5, 3), (33_001, 3, 2), (9, 5, 1),
                                          (4099, 4, 4), (131, 4, 2), (4099, 7, 3), (131, 6, 4), (9, 7, 1)])
@pytest.mark.parametrize("scheme", ["f16x2"])
def test_backward_split_never_uses_rows_past_the_end(m, d_in, n_out, scheme):
    """The kernels fetch whole windows of rows (eight samples of x / dOut through scalar
    buffer descriptors, 16-sample chunks of h2, 128-row tiles) and rely on descriptors
    that end at row m for the ragged tail.  Here every input is a view of a larger
    allocation whose rows past m hold NaN: one such row read and used -- even
    multiplied by zero -- would poison the gradients."""
    g = torch.Generator(device=DEV).manual_seed(11 * m + n_out)
    pad = 160

    def view_of(t):
        big = torch.full((m + pad, t.shape[1]), float("nan"), device=DEV)
        big[:m] = t
        return big[:m]

    x = torch.randn(m, d_in, device=DEV, generator=g)
    p = _params(g, d_in, n_out)
    dout = torch.randn(m, n_out, device=DEV, generator=g) / m
    pack = hip.mlp_pack_w2_f16
    w2p, w2t = pack(p["w2"]), pack(p["w2"], transposed=True)
    _, h1, h2, gate = hip.mlp_tower_forward_split(x, p["w1"], p["b1"], w2p, p["b2"], p["w3"], p["b3"], save=True, save_gate=True)
    want = hip.mlp_tower_backward(x, None, h2, dout, w2t, p["w3"], p["w1"], p["b1"], gate2=gate)
    xv, h2v, dv = view_of(x), view_of(h2), view_of(dout)
    outv, _, h2_again, gate_again = hip.mlp_tower_forward_split(xv, p["w1"], p["b1"], w2p, p["b2"], p["w3"], p["b3"], save=True,
                                                               save_gate=True)
    assert torch.equal(h2_again, h2) and torch.equal(gate_again, gate) and bool(torch.isfinite(outv).all())
    for gate2 in ((gate,) if scheme == "f16x2" else (None, gate)):
        got = hip.mlp_tower_backward(xv, None, h2v, dv, w2t, p["w3"], p["w1"], p["b1"], gate2=gate2)
        for k in p:
            assert bool(torch.isfinite(got[k]).all()), k
            assert torch.equal(got[k], want[k]), k
    dz2 = ((dout @ p["w3"]) * (h2 > 0)).contiguous()
    assert torch.equal(hip.mlp_wgrad_split(view_of(dz2), xv, p["w1"], p["b1"]), hip.mlp_wgrad_split(dz2, x, p["w1"], p["b1"]))


def test_unsupported_widths_are_refused_not_miscomputed():
    """Only widths whose kernels compile without scratch are offered
    (tests/test_kernel_resources.py); anything else must fail loudly."""
    assert not hip.mlp_backward_f16_supports(8, 2) and not hip.mlp_backward_f16_supports(1, 5)
    assert hip.mlp_backward_f16_supports(4, 4) and hip.mlp_forward_f16_supports(8, 8)  # round 5
    # round 6: six and seven observations; 7 x 4 is not compiled (88 scalar registers per step in the exact-plane kernel)
    assert hip.mlp_backward_f16_supports(7, 3) and hip.mlp_backward_f16_supports(6, 4) and not hip.mlp_backward_f16_supports(7, 4)
    assert hip.mlp_forward_f16_supports(9, 1) and hip.mlp_forward_f16_supports(16, 4)  # round 5: class 16
    assert not hip.mlp_forward_f16_supports(17, 1) and not hip.mlp_forward_f16_supports(1, 9) and not hip.mlp_forward_f16_supports(12, 6)
    x = torch.zeros(256, 17, device=DEV)
    h = torch.zeros(256, 256, device=DEV)
    w7, b = torch.zeros(256, 17, device=DEV), torch.zeros(256, device=DEV)
    w3, b3 = torch.zeros(3, 256, device=DEV), torch.zeros(3, device=DEV)
    w2 = torch.zeros(256, 256, device=DEV)
    with pytest.raises(ValueError):
        hip.mlp_tower_backward(x, h, h, torch.zeros(256, 3, device=DEV), hip.mlp_pack_w2_f16(w2, transposed=True), w3, w7, b)
    with pytest.raises(ValueError):
        hip.mlp_tower_forward_split(x, w7, b, hip.mlp_pack_w2_f16(w2), b, w3, b3)


def test_wider_towers_mix_fp32_and_split_kernels():
    """CartPole's tower (5 -> 256 -> 256 -> 3) and a 4 -> 4 one through the fused autograd function (plane kernels
    throughout; round 6: 7 -> 2 and 6 -> 4 too) -- and widths whose backward has no plane kernel (7 -> 4, 8 -> 5:
    fp16-plane forward with h1 and h2 stored, fp32-MFMA data gradient, bf16-plane weight gradient; 12 -> 2, 16 -> 4 the
    same through class 16; 12 -> 6: fp32-MFMA forward too); all must match eager."""
    from rl8_amd.nn import fused_mlp

    torch.manual_seed(6)
    for d_in, n_out in ((5, 3), (4, 4), (7, 2), (6, 4), (7, 4), (8, 5), (3, 6), (12, 2), (16, 4), (12, 6)):
        mlp = torch.nn.Sequential(torch.nn.Linear(d_in, 256), torch.nn.ReLU(), torch.nn.Linear(256, 256)).to(DEV)
        trunk = torch.nn.Sequential(mlp, torch.nn.ReLU()).to(DEV)
        head = torch.nn.Linear(256, n_out).to(DEV)
        x = torch.randn(3000, d_in, device=DEV)
        wts = torch.linspace(-1, 1, n_out, device=DEV)
        ((fused_mlp.tower_forward(trunk, [head], x) * wts).sum() / 3000).backward()
        got = {n: p.grad.clone() for n, p in list(trunk.named_parameters()) + list(head.named_parameters())}
        for p in list(trunk.parameters()) + list(head.parameters()):
            p.grad = None
        ((head(trunk(x)) * wts).sum() / 3000).backward()
        for (n, p) in list(trunk.named_parameters()) + list(head.named_parameters()):
            assert _rel(got[n], p.grad.double()) < 2e-5, (d_in, n)


@pytest.mark.parametrize("m,d_in", [(1, 1), (15, 1), (16, 2), (17, 5), (1000, 3), (16384 + 7, 1), (300_000, 1), (5000, 9)])
def test_wgrad_split_matches_torch(m, d_in):
    g = torch.Generator(device=DEV).manual_seed(m)
    dz2 = torch.randn(m, 256, device=DEV, generator=g)
    x = torch.randn(m, d_in, device=DEV, generator=g) * 3
    w1 = torch.randn(256, d_in, device=DEV, generator=g) * 0.5
    b1 = torch.randn(256, device=DEV, generator=g) * 0.1
    # h1 exactly as the forward kernels form it (fma chain in input order)
    _, h1, _ = hip.mlp_tower_forward(x, w1, b1, hip.mlp_pack_w2(torch.zeros(256, 256, device=DEV)),
                                     torch.zeros(256, device=DEV), torch.zeros(1, 256, device=DEV),
                                     torch.zeros(1, device=DEV), save=True)
    want = dz2.double().t() @ h1.double()
    got = hip.mlp_wgrad_split(dz2, x, w1, b1)
    scale = float(want.abs().max()) + 1e-9
    err = float((got.double() - want).abs().max()) / scale
    err32 = float((hip.mlp_wgrad(dz2, h1).double() - want).abs().max()) / scale
    assert err < 5e-6 and err <= 4 * err32 + 1e-6
    assert torch.equal(got, hip.mlp_wgrad_split(dz2, x, w1, b1))  # fixed summation order


def test_fused_tower_autograd_uses_the_split_kernels_and_matches_eager():
    from rl8_amd.nn import fused_mlp

    assert fused_mlp.FORWARD_GEMM == "f16" and fused_mlp.BACKWARD_GEMM == "f16"  # the shipped defaults
    torch.manual_seed(5)
    mlp = torch.nn.Sequential(torch.nn.Linear(1, 256), torch.nn.ReLU(), torch.nn.Linear(256, 256)).to(DEV)
    trunk = torch.nn.Sequential(mlp, torch.nn.ReLU()).to(DEV)
    head = torch.nn.Linear(256, 2).to(DEV)
    x = torch.randn(3000, 1, device=DEV)
    hip.timer.reset()
    hip.timer.enabled = True
    try:
        out = fused_mlp.tower_forward(trunk, [head], x)
        loss = (out * torch.linspace(-1, 1, 2, device=DEV)).sum() / 3000
        loss.backward()
        launched = set(hip.timer.summary())
    finally:
        hip.timer.enabled = False
    # (this loss gives the two outputs exactly opposite gradients: the gate-mode backward kernels run)
    assert {"mlp_tower_forward_save", "mlp_tower_backward_gate", "mlp_wgrad_gate"} <= launched
    assert mlp[2].__dict__["_rl8_rank_one"] is True  # found on the data: the next forward keeps the gate bits only
    got = {n: p.grad.clone() for n, p in list(trunk.named_parameters()) + list(head.named_parameters())}
    for p in list(trunk.parameters()) + list(head.parameters()):
        p.grad = None
    ((head(trunk(x)) * torch.linspace(-1, 1, 2, device=DEV)).sum() / 3000).backward()
    for (n, p) in list(trunk.named_parameters()) + list(head.named_parameters()):
        assert _rel(got[n], p.grad.double()) < 2e-5, n


def test_fused_tower_keeps_only_gate_bits_for_rank_one_heads_and_recovers_when_wrong():
    """The autograd wrapper's bookkeeping: a single-output tower never stores h2; a two-output tower stores it until a
    backward has found its gradients to be exact negatives, then keeps the gate bits only; and when a later backward
    finds a gradient that is NOT a pair the forward is re-run for h2 and the general kernels produce the same
    gradients as eager PyTorch."""
    from rl8_amd.nn import fused_mlp

    torch.manual_seed(11)
    def tower(n_out):
        mlp = torch.nn.Sequential(torch.nn.Linear(2, 256), torch.nn.ReLU(), torch.nn.Linear(256, 256)).to(DEV)
        return torch.nn.Sequential(mlp, torch.nn.ReLU()).to(DEV), torch.nn.Linear(256, n_out).to(DEV)

    x = torch.randn(5000, 2, device=DEV)
    saved = []
    real = hip.mlp_tower_forward_split

    def spy(*a, **k):
        out = real(*a, **k)
        if k.get("save"):
            saved.append(out[2] is not None)  # was h2 stored?
        return out

    def check(trunk, head, weights):
        params = list(trunk.parameters()) + list(head.parameters())
        for q in params:
            q.grad = None
        (fused_mlp.tower_forward(trunk, [head], x) * weights).sum().backward()
        got = [q.grad.clone() for q in params]
        for q in params:
            q.grad = None
        ((head(trunk(x))) * weights).sum().backward()
        for a, q in zip(got, params):
            assert _rel(a, q.grad.double()) < 2e-5

    hip.mlp_tower_forward_split = spy
    fused_mlp.hip.mlp_tower_forward_split = spy
    try:
        trunk1, head1 = tower(1)
        check(trunk1, head1, torch.tensor([0.7], device=DEV) / 5000)
        assert saved == [False]                       # one output: gate bits only from the start
        saved.clear()
        trunk2, head2 = tower(2)
        pair = torch.tensor([1.0, -1.0], device=DEV) / 5000
        check(trunk2, head2, pair)                    # h2 stored; the backward finds a pair
        check(trunk2, head2, pair)                    # gate bits only
        check(trunk2, head2, torch.tensor([1.0, 0.5], device=DEV) / 5000)   # not a pair: forward re-run inside backward
        check(trunk2, head2, pair)                    # (the flag was dropped: h2 stored again)
        assert saved == [True, False, False, True, True], saved  # third entry: the gate-only forward; fourth: its re-run for h2
    finally:
        hip.mlp_tower_forward_split = real
        fused_mlp.hip.mlp_tower_forward_split = real


def test_fused_tower_refuses_a_backward_after_an_in_place_weight_change():
    """The wrapper saves W2 itself (not only its packed copies), so autograd's version check fires exactly as it
    would for the eager modules: a drop-in must fail loudly, not differentiate through the new weights."""
    from rl8_amd.nn import fused_mlp

    torch.manual_seed(2)
    mlp = torch.nn.Sequential(torch.nn.Linear(1, 256), torch.nn.ReLU(), torch.nn.Linear(256, 256)).to(DEV)
    trunk = torch.nn.Sequential(mlp, torch.nn.ReLU()).to(DEV)
    head = torch.nn.Linear(256, 2).to(DEV)
    x = torch.randn(500, 1, device=DEV)
    for target in (mlp[2].weight, mlp[0].weight, head.weight):
        out = fused_mlp.tower_forward(trunk, [head], x)
        with torch.no_grad():
            target.mul_(1.5)
        with pytest.raises(RuntimeError, match="modified by an inplace operation"):
            out.sum().backward()


def test_fused_tower_takes_the_pair_decision_from_the_caller():
    """`pair_gradients=True` (what Algorithm passes for a two-way Categorical under the fused loss): no h2 from the
    FIRST forward of a fresh tower; the device-side check stays the guard -- a gradient that is not a pair re-runs the
    forward for h2 and the general kernels give eager PyTorch's gradients; `False` always stores h2."""
    from rl8_amd.nn import fused_mlp

    torch.manual_seed(13)
    mlp = torch.nn.Sequential(torch.nn.Linear(1, 256), torch.nn.ReLU(), torch.nn.Linear(256, 256)).to(DEV)
    trunk = torch.nn.Sequential(mlp, torch.nn.ReLU()).to(DEV)
    head = torch.nn.Linear(256, 2).to(DEV)
    x = torch.randn(4000, 1, device=DEV) * 20
    saved = []
    real = hip.mlp_tower_forward_split

    def spy(*a, **k):
        out = real(*a, **k)
        if k.get("save"):
            saved.append(out[2] is not None)
        return out

    def check(weights, hint):
        params = list(trunk.parameters()) + list(head.parameters())
        for q in params:
            q.grad = None
        hip.timer.reset()
        hip.timer.enabled = True
        try:
            (fused_mlp.tower_forward(trunk, [head], x, pair_gradients=hint) * weights).sum().backward()
            launched = set(hip.timer.summary())
        finally:
            hip.timer.enabled = False
        got = [q.grad.clone() for q in params]
        for q in params:
            q.grad = None
        (head(trunk(x)) * weights).sum().backward()
        for a, q in zip(got, params):
            assert _rel(a, q.grad.double()) < 2e-5
        return launched

    fused_mlp.hip.mlp_tower_forward_split = spy
    try:
        pair = torch.tensor([1.0, -1.0], device=DEV) / 4000
        launched = check(pair, True)
        assert saved == [False] and {"mlp_tower_backward_gate", "mlp_wgrad_gate"} <= launched
        saved.clear()
        launched = check(torch.tensor([1.0, 0.25], device=DEV) / 4000, True)   # the promise broken: forward re-run
        assert saved == [False, True] and "mlp_tower_backward_gate" not in launched
        saved.clear()
        check(pair, False)
        assert saved == [True]
        saved.clear()
        with fused_mlp.expect_pair_gradients():
            assert fused_mlp.pair_hint()
        assert not fused_mlp.pair_hint()
    finally:
        fused_mlp.hip.mlp_tower_forward_split = real


def test_gate_pack_follows_the_head_parameters_not_the_concatenated_temporary():
    """Two heads: w3 is a fresh torch.cat every forward (version 0, recycled address); the cached W2*w3e pack must be
    re-made when a HEAD parameter changes while W2 does not."""
    from rl8_amd.nn import fused_mlp

    torch.manual_seed(17)
    mlp = torch.nn.Sequential(torch.nn.Linear(1, 256), torch.nn.ReLU(), torch.nn.Linear(256, 256)).to(DEV)
    trunk = torch.nn.Sequential(mlp, torch.nn.ReLU()).to(DEV)
    heads = [torch.nn.Linear(256, 1).to(DEV), torch.nn.Linear(256, 1).to(DEV)]
    x = torch.randn(3000, 1, device=DEV) * 10
    pair = torch.tensor([1.0, -1.0], device=DEV) / 3000
    params = list(trunk.parameters()) + [q for h in heads for q in h.parameters()]
    for it in range(3):
        for q in params:
            q.grad = None
        (fused_mlp.tower_forward(trunk, heads, x, pair_gradients=True) * pair).sum().backward()
        got = [q.grad.clone() for q in params]
        for q in params:
            q.grad = None
        (torch.cat([h(trunk(x)) for h in heads], -1) * pair).sum().backward()
        for a, q in zip(got, params):
            assert _rel(a, q.grad.double()) < 2e-5, it
        with torch.no_grad():  # only the heads move (a frozen trunk): W2's version stays
            heads[0].weight.add_(torch.randn_like(heads[0].weight) * 0.05)
            heads[1].weight.mul_(0.5)


@pytest.mark.parametrize("scheme", ["f16x2", "f16x2-gate"])
@pytest.mark.parametrize("m,parts", [(1 << 23, 8), (1 << 25, 4)])
def test_full_size_launch_equals_its_chunks(m, parts, scheme):
    """BASELINE's training launch (2^25 rows: `Algorithm.step` feeds the towers the whole
    33.5 M-sample batch in one pass; 32 GiB of h2, offsets past 2^32 bytes) and a 2^23-row one
    against the same rows in several launches: a row's outputs depend on nothing but that
    row (fixed k order), so forward results must be bit-identical however the rows are
    cut into launches; gradients are sums over rows, so they must agree to fp32 rounding."""
    g = torch.Generator(device=DEV).manual_seed(99)
    x = torch.empty(m, 1, device=DEV).uniform_(-100, 100, generator=g)  # DiscreteDummyEnv observations
    p = _params(g, 1, 2)
    dout = torch.randn(m, 2, device=DEV, generator=g) / m
    gate_pack = None
    if scheme == "f16x2-gate":  # the headline's own kernels: a two-way categorical's gradients, gate-mode backward
        dout[:, 1] = -dout[:, 0]
        gate_pack = lambda: hip.mlp_pack_w2_f16_gate(p["w2"], p["w3"])  # noqa: E731
    pack = hip.mlp_pack_w2_f16
    w2s, w2ts = pack(p["w2"]), pack(p["w2"], transposed=True)
    out, _, h2, gate = hip.mlp_tower_forward_split(x, p["w1"], p["b1"], w2s, p["b2"], p["w3"], p["b3"], save=True,
                                                   save_h1=False, save_gate=True)
    hip.timer.reset()
    hip.timer.enabled = True
    try:
        full = hip.mlp_tower_backward(x, None, h2, dout, w2ts, p["w3"], p["w1"], p["b1"], gate2=gate, gate_pack=gate_pack)
        launched = set(hip.timer.summary())
    finally:
        hip.timer.enabled = False
    assert ("mlp_tower_backward_gate" in launched and "mlp_wgrad_gate" in launched) == (scheme == "f16x2-gate")
    acc = {k: torch.zeros_like(v, dtype=torch.float64) for k, v in full.items()}
    step = m // parts
    for i in range(parts):
        sl = slice(i * step, (i + 1) * step)
        o, _, h2c, gc = hip.mlp_tower_forward_split(x[sl], p["w1"], p["b1"], w2s, p["b2"], p["w3"], p["b3"], save=True,
                                                    save_h1=False, save_gate=True)
        assert torch.equal(o, out[sl]) and torch.equal(h2c, h2[sl]) and torch.equal(gc, gate[sl])
        part = hip.mlp_tower_backward(x[sl], None, h2c, dout[sl].contiguous(), w2ts, p["w3"], p["w1"], p["b1"], gate2=gc,
                                      gate_pack=gate_pack)
        for k in acc:
            acc[k] += part[k].double()
        del o, h2c, gc, part
    # (the same bound at both sizes: the weight-gradient launcher sums 2^23-row segments, so
    # no fp32 accumulation chain is longer in the 2^25-row launch than in a 2^23-row one)
    for k in acc:
        assert _rel(full[k], acc[k]) < 2e-5, (k, _rel(full[k], acc[k]))
    # inference launch of the rollout (2^20 rows) against the training launch's rows
    o_inf, _, _ = hip.mlp_tower_forward_split(x[: 1 << 20], p["w1"], p["b1"], w2s, p["b2"], p["w3"], p["b3"])
    assert torch.equal(o_inf, out[: 1 << 20])


@pytest.mark.parametrize("scheme", ["f16x2"])
def test_split_kernels_are_deterministic_and_head_gradients_exact(scheme):
    """Run-to-run determinism of every bf16- / fp16-plane kernel at the rollout launch size (a
    property the fixed summation orders promise), and the head gradients against fp64.

    Not a hazard hunt: the round-1 wrong-dW3 event is closed from its instruction
    listing (a packed fp32 op's source overwritten within two VALU slots beside bf16
    MFMAs, tools/check_inflight_regs.py::packed_war); tests/test_kernel_resources.py
    proves on the shipped ISA that the pair cannot form and that no hand-issued load's
    destination is touched before its wait. Three runs state the determinism property."""
    m = 1 << 20
    g = torch.Generator(device=DEV).manual_seed(5)
    x = torch.empty(m, 1, device=DEV).uniform_(-3, 3, generator=g)
    p = _params(g, 1, 2)
    dout = torch.randn(m, 2, device=DEV, generator=g) / m
    pack = hip.mlp_pack_w2_f16
    w2s, w2ts = pack(p["w2"]), pack(p["w2"], transposed=True)

    def run():
        out, _, h2, gate = hip.mlp_tower_forward_split(x, p["w1"], p["b1"], w2s, p["b2"], p["w3"], p["b3"], save=True,
                                                       save_h1=False, save_gate=True)
        grads = hip.mlp_tower_backward(x, None, h2, dout, w2ts, p["w3"], p["w1"], p["b1"], gate2=gate)
        return [out, h2, gate] + [grads[k] for k in ("w1", "b1", "w2", "b2", "w3", "b3")]

    first = run()
    want_w3 = dout.double().T @ first[1].double()
    assert _rel(first[7], want_w3) < 5e-6
    for trial in range(3):
        again = run()
        for a, b in zip(first, again):
            assert torch.equal(a, b), trial


def test_fp32_mfma_kernels_are_deterministic():
    """The same property for the fp32-MFMA generation (which uses packed fp32 ops by
    design, beside fp32 MFMAs that do not co-issue with the VALU)."""
    m = 1 << 19
    g = torch.Generator(device=DEV).manual_seed(6)
    x = torch.empty(m, 5, device=DEV).uniform_(-3, 3, generator=g)
    p = _params(g, 5, 3)
    dout = torch.randn(m, 3, device=DEV, generator=g) / m
    w2p, w2tp = hip.mlp_pack_w2(p["w2"]), hip.mlp_pack_w2(p["w2"], transposed=True)

    def run():
        out, h1, h2 = hip.mlp_tower_forward(x, p["w1"], p["b1"], w2p, p["b2"], p["w3"], p["b3"], save=True)
        grads = hip.mlp_tower_backward(x, h1, h2, dout, w2tp, p["w3"], p["w1"], p["b1"], wgrad_split=True)
        return [out, h1, h2] + [grads[k] for k in ("w1", "b1", "w2", "b2", "w3", "b3")]

    first = run()
    for trial in range(3):
        for a, b in zip(first, run()):
            assert torch.equal(a, b), trial


@pytest.mark.parametrize("planes", ["f16"])
def test_algorithm_agrees_between_the_tower_generations(planes):
    """collect() + step() with the bf16-plane / fp16-plane towers against the same seeded
    run on the fp32-MFMA towers: same actions, same statistics, losses to 1e-5 (the
    tolerance north_star sets against the reference's CPU path)."""
    from rl8_amd import AlgorithmConfig
    from rl8_amd.env import DiscreteDummyEnv
    from rl8_amd.nn import fused_mlp

    def run(gemm):
        old = fused_mlp.FORWARD_GEMM, fused_mlp.BACKWARD_GEMM
        fused_mlp.FORWARD_GEMM = fused_mlp.BACKWARD_GEMM = gemm
        try:
            torch.manual_seed(7)
            algo = AlgorithmConfig(num_envs=4096, horizon=32).build(DiscreteDummyEnv)
            c = algo.collect()
            actions = algo.buffer["actions"].clone()
            s = algo.step()
            return c, actions, s
        finally:
            fused_mlp.FORWARD_GEMM, fused_mlp.BACKWARD_GEMM = old

    c_split, a_split, s_split = run(planes)
    c_f32, a_f32, s_f32 = run("f32")
    assert torch.equal(a_split, a_f32)  # bit-exact action indices
    for k in c_f32:
        if not k.startswith("profiling"):
            assert c_split[k] == pytest.approx(c_f32[k], rel=1e-6, abs=1e-6), k
    for k in ("losses/policy", "losses/vf", "losses/total"):
        assert s_split[k] == pytest.approx(s_f32[k], rel=1e-5, abs=1e-5), k


def test_guard_sends_a_call_with_a_nan_to_the_exact_planes():
    """a NaN in dOut is not a "small entry" -- it used to be skipped by the guard's sample (every comparison
    with NaN is false).  The call now goes to the exact bf16 planes, and the NaN reaches the gradients as it would
    through torch's own backward."""
    m, d_in, n_out = 2048, 1, 1
    g = torch.Generator(device=DEV).manual_seed(8)
    x = torch.randn(m, d_in, device=DEV, generator=g) * 3
    p = _params(g, d_in, n_out)
    dout = torch.randn(m, n_out, device=DEV, generator=g) / m
    dout[m // 2, 0] = float("nan")
    w2p, w2t = hip.mlp_pack_w2_f16(p["w2"]), hip.mlp_pack_w2_f16(p["w2"], transposed=True)
    _, _, _, gate = hip.mlp_tower_forward_split(x, p["w1"], p["b1"], w2p, p["b2"], p["w3"], p["b3"], save=True, save_gate=True)
    before = hip.wgrad_guard_counts()
    out = hip.mlp_tower_backward(x, None, None, dout, w2t, p["w3"], p["w1"], p["b1"], gate2=gate,
                                 gate_pack=lambda: hip.mlp_pack_w2_f16_gate(p["w2"], p["w3"]), w2=p["w2"], b2=p["b2"])
    calls, fires = (a - b for a, b in zip(hip.wgrad_guard_counts(), before))
    assert (calls, fires) == (1, 1)
    assert bool(torch.isnan(out["w2"]).any()) and bool(torch.isnan(out["b3"]).any())


@pytest.mark.parametrize("m,d_in,n_out", [(1, 1, 2), (127, 2, 3), (128, 3, 4), (4097, 1, 3), (33_000, 3, 2), (70_001, 2, 4)])
def test_general_data_gradient_rows_and_tile_kernels_agree(m, d_in, n_out, monkeypatch):
    """Round 5: the general-head data gradient in the rows shape (mlp_rows_backward_general_kernel: d_in <= 3, n_out 2..4)
    beside the tile kernel it replaced there (RL8_MLP_DGRAD_TILE=1; still what d_in 4, 5 run): dW1 / db1 of both
    against fp64 on the saved activations, the new one no further off than 3x the old one + 1e-6, everything else of the
    backward (the weight-gradient kernel's outputs) bit for bit the same."""
    g = torch.Generator(device=DEV).manual_seed(31 * m + d_in + n_out)
    x = torch.randn(m, d_in, device=DEV, generator=g) * 3
    p = _params(g, d_in, n_out)
    dout = torch.randn(m, n_out, device=DEV, generator=g) / m
    dout *= 10.0 ** torch.randint(-3, 2, (m, 1), device=DEV, generator=g).float()   # rows of mixed magnitude
    w2p, w2t = hip.mlp_pack_w2_f16(p["w2"]), hip.mlp_pack_w2_f16(p["w2"], transposed=True)
    _, h1, h2, gate = hip.mlp_tower_forward_split(x, p["w1"], p["b1"], w2p, p["b2"], p["w3"], p["b3"], save=True, save_gate=True)
    dz2 = (dout.double() @ p["w3"].double()) * (h2 > 0)
    dz1 = (dz2 @ p["w2"].double()) * (h1 > 0)
    want = {"w1": dz1.T @ x.double(), "b1": dz1.sum(0)}

    def run(tile):
        monkeypatch.setenv("RL8_MLP_DGRAD_TILE", "1" if tile else "0")
        return hip.mlp_tower_backward(x, None, h2, dout, w2t, p["w3"], p["w1"], p["b1"], gate2=gate, assume_general=True)

    rows, tile = run(False), run(True)
    for k in ("w1", "b1"):
        err_rows, err_tile = _rel(rows[k], want[k]), _rel(tile[k], want[k])
        assert err_rows < 2e-5 and err_rows <= 3 * err_tile + 1e-6, (k, err_rows, err_tile)
    for k in ("w2", "b2", "w3", "b3"):
        assert torch.equal(rows[k], tile[k]), k
    again = run(False)
    for k in rows:
        assert torch.equal(rows[k], again[k]), k  # fixed summation order, no race


@pytest.mark.parametrize("m,d_in,n_out,pair", [(1, 4, 1, False), (127, 5, 2, True), (4097, 4, 3, False), (33_000, 5, 3, False),
                                              (70_001, 5, 4, False), (5000, 4, 2, False), (9000, 5, 1, False)])
def test_wide_data_gradients_follow_the_forwards_gates(m, d_in, n_out, pair, monkeypatch):
    """Round 5, d_in = 4, 5: the rows-shape data gradients with layer 1 on the matrix pipe (class 8: the gate of h1 is the
    forward kernel's own z1 again, dW1 / db1 are MFMAs over the wave's rows, one running-sums array per workgroup updated
    in wave order) beside the tile kernel they replaced (RL8_MLP_DGRAD_TILE=1, general mode, which recomputes the gate
    with an fp32 fma chain).  Reference: fp64 on the forward's saved activations -- h1 included, so a gate the backward opens where
    the forward closed it shows as an error of the size of one row's term: the class-8 kernels stay at rounding level
    (2e-6 of the sum of the magnitudes of all terms), the tile kernels within one such term."""
    g = torch.Generator(device=DEV).manual_seed(11 * m + d_in + n_out)
    x = torch.randn(m, d_in, device=DEV, generator=g) * 3
    p = _params(g, d_in, n_out)
    if pair:
        g0 = torch.randn(m, device=DEV, generator=g) / m
        dout = torch.stack([g0, -g0], 1).contiguous()
    else:
        dout = torch.randn(m, n_out, device=DEV, generator=g) / m
    dout *= 10.0 ** torch.randint(-3, 2, (m, 1), device=DEV, generator=g).float()
    w2t = hip.mlp_pack_w2_f16(p["w2"], transposed=True)
    _, h1, h2, gate = hip.mlp_tower_forward_split(x, p["w1"], p["b1"], hip.mlp_pack_w2_f16(p["w2"]), p["b2"], p["w3"], p["b3"],
                                                  save=True, save_gate=True)
    dz2 = (dout.double() @ p["w3"].double()) * (h2 > 0)
    dz1 = (dz2 @ p["w2"].double()) * (h1 > 0)
    want = {"w1": dz1.T @ x.double(), "b1": dz1.sum(0)}
    inner = (dz2.abs() @ p["w2"].double().abs()) * (h1 > 0)
    size = {"w1": inner.T @ x.double().abs(), "b1": inner.sum(0)}
    gate_mode = n_out == 1 or pair

    def run(tile):  # (tile: the reference kernel -- general mode, whatever the head)
        monkeypatch.setenv("RL8_MLP_DGRAD_TILE", "1" if tile else "0")
        gate_pack = (lambda: hip.mlp_pack_w2_f16_gate(p["w2"], p["w3"])) if gate_mode and not tile else None
        return hip.mlp_tower_backward(x, None, h2, dout, w2t, p["w3"], p["w1"], p["b1"], gate2=gate, gate_pack=gate_pack,
                                      assume_general=n_out == 2 and (tile or not gate_mode))

    rows, tile = run(False), run(True)
    for k in ("w1", "b1"):
        floor = size[k].max() * 1e-30 + 1e-300
        err_rows = float(((rows[k].double() - want[k]).abs() / (size[k] + floor)).max())
        err_tile = float(((tile[k].double() - want[k]).abs() / (size[k] + floor)).max())
        assert err_rows < 2e-6, (k, err_rows, err_tile)
        assert err_tile < 2e-6 + 4.0 / m, (k, err_tile)
    if not gate_mode:
        for k in ("w2", "b2", "w3", "b3"):
            assert torch.equal(rows[k], tile[k]), k  # (the weight-gradient kernel's: the same launch both times)
    again = run(False)
    for k in rows:
        assert torch.equal(rows[k], again[k]), k  # the chain of waves adds in a fixed order: bit for bit


@pytest.mark.parametrize("x_scale", [1e-12, 1e-3, 1.0, 3e5])
@pytest.mark.parametrize("d_in,n_out", [(5, 3), (4, 1), (8, 2), (11, 2), (16, 4)])
def test_class8_scales_over_observation_magnitudes(d_in, n_out, x_scale):
    """Class 8 carries three powers of two per row (x for z1, h1 for layer 2, dZ1 / x~ for dW1): observations from 1e-12
    to 3e5, rows of exact zeros and rows a million times the others beside each other, forward against fp64 and (where the
    plane backward serves the width) dW1 / db1 against fp64 on the forward's own activations."""
    m = 3000
    g = torch.Generator(device=DEV).manual_seed(int(d_in * 100 + n_out + abs(math.log10(x_scale)) * 7))
    x = torch.randn(m, d_in, device=DEV, generator=g) * x_scale
    x[::7] = 0.0
    x[5::11] *= 1e6 if x_scale < 1.0 else 1e-6
    x[3::13, 1] = 0.0
    p = _params(g, d_in, n_out)
    want, h1w, h2w = _tower(x.double(), {k: v.double() for k, v in p.items()})
    packed = hip.mlp_pack_w2_f16(p["w2"])
    out, h1, h2, gate = hip.mlp_tower_forward_split(x, p["w1"], p["b1"], packed, p["b2"], p["w3"], p["b3"], save=True, save_gate=True)
    assert bool(torch.isfinite(out).all())
    assert _rel(out, want) < 5e-6 and _rel(h1, h1w) < 1e-6 and _rel(h2, h2w) < 3e-6
    # rows of zeros: h1 = relu(b1) exactly
    assert torch.equal(h1[::7], torch.relu(p["b1"]).expand_as(h1[::7]))
    if not hip.mlp_backward_f16_supports(d_in, n_out):
        return
    dout = torch.randn(m, n_out, device=DEV, generator=g) / m
    dz2 = (dout.double() @ p["w3"].double()) * (h2 > 0)
    dz1 = (dz2 @ p["w2"].double()) * (h1 > 0)
    want_g = {"w1": dz1.T @ x.double(), "b1": dz1.sum(0)}
    inner = (dz2.abs() @ p["w2"].double().abs()) * (h1 > 0)
    size = {"w1": inner.T @ x.double().abs(), "b1": inner.sum(0)}
    w2t = hip.mlp_pack_w2_f16(p["w2"], transposed=True)
    got = hip.mlp_tower_backward(x, None, h2, dout, w2t, p["w3"], p["w1"], p["b1"], gate2=gate,
                                 gate_pack=(lambda: hip.mlp_pack_w2_f16_gate(p["w2"], p["w3"])) if n_out == 1 else None)
    for k in ("w1", "b1"):
        assert bool(torch.isfinite(got[k]).all()), k
        floor = size[k].max() * 1e-30 + 1e-300
        assert float(((got[k].double() - want_g[k]).abs() / (size[k] + floor)).max()) < 2e-6, k


_CHAIN_CHILD = r"""
import sys, torch
sys.path.insert(0, {root!r})
from rl8_amd import hip
DEV = "cuda"
def params(g, d_in, n_out):
    r = lambda *s, k=1.0: (torch.rand(*s, device=DEV, generator=g) * 2 - 1) * k
    return dict(w1=r(256, d_in, k=1.0), b1=r(256, k=0.5), w2=r(256, 256, k=0.0625), b2=r(256, k=0.0625),
                w3=r(n_out, 256, k=0.0625), b3=r(n_out, k=0.0625))
worst = 0.0
for d_in, n_out in ((5, 3), (5, 1), (4, 2), (4, 1)):
    for m in (1, 17, 31, 33, 100, 128 * 37 + 5):
        g = torch.Generator(device=DEV).manual_seed(m * 7 + d_in + n_out)
        x = torch.randn(m, d_in, device=DEV, generator=g) * 2
        p = params(g, d_in, n_out)
        dout = torch.randn(m, n_out, device=DEV, generator=g)
        _, h1, h2, gate = hip.mlp_tower_forward_split(x, p["w1"], p["b1"], hip.mlp_pack_w2_f16(p["w2"]), p["b2"], p["w3"], p["b3"],
                                                      save=True, save_gate=True)
        w2t = hip.mlp_pack_w2_f16(p["w2"], transposed=True)
        gate_pack = (lambda: hip.mlp_pack_w2_f16_gate(p["w2"], p["w3"])) if n_out == 1 else None
        runs = [hip.mlp_tower_backward(x, None, h2, dout, w2t, p["w3"], p["w1"], p["b1"], gate2=gate, gate_pack=gate_pack,
                                       assume_general=n_out == 2) for _ in range(2)]
        torch.cuda.synchronize()
        for k in runs[0]:
            assert torch.equal(runs[0][k], runs[1][k]), (d_in, n_out, m, k)
        dz2 = (dout.double() @ p["w3"].double()) * (h2 > 0)
        open1 = h1 > 0
        dz1 = (dz2 @ p["w2"].double()) * open1
        inner = (dz2.abs() @ p["w2"].double().abs()) * open1
        want = dict(w1=dz1.T @ x.double(), b1=dz1.sum(0))
        size = dict(w1=inner.T @ x.double().abs(), b1=inner.sum(0))
        for k in ("w1", "b1"):
            err = float(((runs[0][k].double() - want[k]).abs() / (size[k] + 1e-300 + 1e-30 * float(size[k].max()))).max())
            assert err < 2e-6, (d_in, n_out, m, k, err)
            worst = max(worst, err)
print("class8 chain ok", worst)
"""


def test_class8_chain_with_empty_waves_and_one_workgroup():
    """the class-8 data gradients order the four waves' dW1 / db1 updates with an unbounded LDS poll of the
    predecessor wave's counter (mlp_rows_kernels.hip, rows8_epilogue) -- correct only while every wave runs the epilogue
    once per tile.  Exercised where that is most at risk: fewer rows than one wave's 32 (three EMPTY waves), m not a
    multiple of 128 (ragged last tile), and ONE workgroup for the whole launch (``RL8_MLP_GRID_CAP=1``, read once per
    process, hence a child interpreter: 38 tiles, counters up to 608) -- dW1 / db1 against fp64 on the forward's own
    activations, twice, bit for bit.  A broken invariant shows as the child's timeout, not as a wrong number."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RL8_MLP_GRID_CAP="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-c", _CHAIN_CHILD.format(root=root)], capture_output=True, text=True, timeout=240,
                         env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    assert "class8 chain ok" in out.stdout


@pytest.mark.parametrize("d_in,n_out", [(5, 3), (5, 1), (4, 2)])
def test_class8_data_gradient_over_many_tiles(d_in, n_out):
    """2^20 + 77 rows: every workgroup runs 16-17 tiles, so the waves' chain of running-sums updates (one array per
    workgroup, a counter per wave that only grows) is exercised across tiles, with a ragged last tile.  dW1 / db1 against fp64
    on the forward's activations; twice, bit for bit."""
    m = (1 << 20) + 77
    g = torch.Generator(device=DEV).manual_seed(d_in * 10 + n_out)
    x = torch.randn(m, d_in, device=DEV, generator=g) * 2
    p = _params(g, d_in, n_out)
    dout = torch.randn(m, n_out, device=DEV, generator=g) / m
    _, h1, h2, gate = hip.mlp_tower_forward_split(x, p["w1"], p["b1"], hip.mlp_pack_w2_f16(p["w2"]), p["b2"], p["w3"], p["b3"],
                                                  save=True, save_gate=True)
    w2t = hip.mlp_pack_w2_f16(p["w2"], transposed=True)
    gate_pack = (lambda: hip.mlp_pack_w2_f16_gate(p["w2"], p["w3"])) if n_out == 1 else None
    got = hip.mlp_tower_backward(x, None, h2, dout, w2t, p["w3"], p["w1"], p["b1"], gate2=gate, gate_pack=gate_pack,
                                 assume_general=n_out == 2)
    again = hip.mlp_tower_backward(x, None, h2, dout, w2t, p["w3"], p["w1"], p["b1"], gate2=gate, gate_pack=gate_pack,
                                   assume_general=n_out == 2)
    for k in got:
        assert torch.equal(got[k], again[k]), k
    want = {"w1": torch.zeros(256, d_in, dtype=torch.float64, device=DEV), "b1": torch.zeros(256, dtype=torch.float64, device=DEV)}
    size = {"w1": torch.zeros(256, d_in, dtype=torch.float64, device=DEV), "b1": torch.zeros(256, dtype=torch.float64, device=DEV)}
    w2d, w3d = p["w2"].double(), p["w3"].double()
    for lo in range(0, m, 1 << 18):  # (fp64 in slices: the whole dZ1 would be 2 GiB)
        sl = slice(lo, min(m, lo + (1 << 18)))
        dz2 = (dout[sl].double() @ w3d) * (h2[sl] > 0)
        open1 = h1[sl] > 0
        dz1 = (dz2 @ w2d) * open1
        inner = (dz2.abs() @ w2d.abs()) * open1
        xd = x[sl].double()
        want["w1"] += dz1.T @ xd
        want["b1"] += dz1.sum(0)
        size["w1"] += inner.T @ xd.abs()
        size["b1"] += inner.sum(0)
    for k in ("w1", "b1"):
        assert float(((got[k].double() - want[k]).abs() / (size[k] + 1e-300)).max()) < 2e-6, k
