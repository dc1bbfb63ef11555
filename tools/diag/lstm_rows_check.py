"""rl8_lstm_rows_backward_f32 against an fp64 model of the same recurrences, and its
time beside the fp32-MFMA kernel's on the recurrent bench's shape.

    python tools/diag/lstm_rows_check.py [--time]
"""
from __future__ import annotations

import argparse
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))

from rl8_amd import hip  # noqa: E402

H = 256


def model(c0, gates, cs, dhs, w_hh):
    """fp64: dgates [B, L, 4, 256]."""
    b, l = dhs.shape[:2]
    c0, gates, cs, dhs, w = (t.double() for t in (c0, gates, cs, dhs, w_hh))
    dg = torch.zeros(b, l, 4, H, dtype=torch.float64, device=dhs.device)
    dh_carry = torch.zeros(b, H, dtype=torch.float64, device=dhs.device)
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
        dh_carry = dg[:, t].reshape(b, 4 * H) @ w
    return dg


def inputs(b, l, dev, seed):
    gen = torch.Generator(device=dev).manual_seed(seed)
    r = lambda *s: torch.rand(*s, device=dev, generator=gen)  # noqa: E731
    gates = r(b, l, 4, H)
    gates[:, :, 2] = gates[:, :, 2] * 2 - 1
    cs = (r(b, l, H) * 2 - 1) * 1.5
    c0 = (r(b, H) * 2 - 1) * 1.5
    # gradients of a mean loss: tiny, and of very different size from row to row
    dhs = (r(b, l, H) * 2 - 1) * torch.exp(-12 * r(b, 1, 1)) * 1e-3
    w_hh = (r(4 * H, H) * 2 - 1) / 16
    return c0, gates, cs, dhs, w_hh


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--time", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    failures = 0
    for b in (1, 31, 32, 100, 128, 129, 257, 1000, 4101, 33000):
        for l in (1, 2, 4, 7):
            c0, gates, cs, dhs, w_hh = inputs(b, l, dev, 1000 * b + l)
            packed = hip.lstm_rows_backward_pack(w_hh)
            got = hip.lstm_rows_backward(c0, gates, cs, dhs, packed).double()
            want = model(c0, gates, cs, dhs, w_hh)
            # per sequence, relative to that sequence's largest gradient entry
            scale = want.abs().amax(dim=(1, 2, 3), keepdim=True).clamp_min(1e-300)
            err = float(((got - want).abs() / scale).max())
            ok = err < 2e-6 and bool(torch.isfinite(got).all())
            failures += not ok
            print(f"b={b:6d} l={l} max err / row max {err:.2e} {'ok' if ok else 'FAIL'}", flush=True)
    print("FAILURES", failures)
    if args.time:
        b, l = 1 << 19, 4
        c0, gates, cs, dhs, w_hh = inputs(b, l, dev, 7)
        packed = hip.lstm_rows_backward_pack(w_hh)
        whht = hip.lstm_pack_transposed(w_hh)
        x = torch.zeros(b, l, 1, device=dev)
        h0 = torch.zeros(b, H, device=dev)
        hs = torch.zeros(b, l, H, device=dev)

        def timed(fn, rounds=5):
            fn()
            torch.cuda.synchronize()
            a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(rounds):
                fn()
            e.record()
            torch.cuda.synchronize()
            return a.elapsed_time(e) / rounds

        new = timed(lambda: hip.lstm_rows_backward(c0, gates, cs, dhs, packed))
        lib = hip.load()
        dgates = torch.empty(b, l, 4, H, device=dev)
        import ctypes as C
        rows = C.c_int(0)

        def old():
            hip._check(lib.rl8_lstm_backward_f32(hip._ptr(x), b, l, 1, hip._ptr(c0), hip._ptr(gates), hip._ptr(cs), hip._ptr(dhs),
                                                 hip._ptr(whht), hip._ptr(dgates), None, C.byref(rows), hip._stream()),
                       "rl8_lstm_backward_f32")

        prev = timed(old)
        print(f"2^19 sequences x 4 steps: rows kernel {new:.2f} ms, fp32-MFMA kernel {prev:.2f} ms")
        # the HEADS form (the product's: dL/dh_t from four floats per row-step), fp16 planes (round 6) and bf16 planes
        import os
        gen = torch.Generator(device=dev).manual_seed(3)
        dout = (torch.rand(b * l, 3, device=dev, generator=gen) * 2 - 1) * 1e-3
        w = (torch.rand(3, H, device=dev, generator=gen) * 2 - 1) / 16
        for planes in ("f16", "bf16", "f16", "bf16"):
            os.environ["RL8_AMD_LSTM_BACKWARD_PLANES"] = planes
            t = timed(lambda: hip.lstm_rows_backward(c0, gates, cs, None, packed, heads=(dout, w)))
            print(f"heads form, {planes} planes: {t:.2f} ms per 2^21 row-steps", flush=True)
        os.environ.pop("RL8_AMD_LSTM_BACKWARD_PLANES", None)
    sys.exit(1 if failures else 0)


if __name__ == "__main__":
    main()
