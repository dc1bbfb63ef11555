"""Time per 2^20 rows of the tower kernels over observation / head widths (round 5: width classes of the plane kernels).

    python tools/diag/tower_width_sweep.py [--rows 1048576] [--reps 10]

Prints one line per (d_in, n_out): forward (inference / gate bits only / with h2), gate-mode and general data gradient,
gate-bits and general weight gradient, in microseconds (torch events on the launch stream, median of the repetitions).
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

import torch

from rl8_amd import hip

p = argparse.ArgumentParser()
p.add_argument("--rows", type=int, default=1 << 20)
p.add_argument("--reps", type=int, default=10)
p.add_argument("--widths", default="1x1,1x2,2x2,3x1,4x1,4x4,5x1,5x3,6x2,8x1,8x4,8x8,3x8,9x1,12x2,16x4")
args = p.parse_args()
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
m = args.rows


def timed(fn):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(args.reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    return sorted(ts)[len(ts) // 2]


print(f"rows {m}: microseconds per call")
print(f"{'d x n':>7} {'fwd':>8} {'fwd+bits':>9} {'fwd+h2':>8} {'dgrad_gate':>11} {'dgrad':>8} {'wgrad_gate':>11} {'wgrad':>8}")
for spec in args.widths.split(","):
    d, n = (int(v) for v in spec.split("x"))
    x = torch.randn(m, d, device=dev, generator=g) * 3
    w1 = torch.randn(256, d, device=dev, generator=g) * 0.5
    b1 = torch.randn(256, device=dev, generator=g) * 0.1
    w2 = torch.randn(256, 256, device=dev, generator=g) / 16
    b2 = torch.randn(256, device=dev, generator=g) * 0.1
    w3 = torch.randn(n, 256, device=dev, generator=g) / 16
    b3 = torch.randn(n, device=dev, generator=g)
    dout = torch.randn(m, n, device=dev, generator=g) / m
    if n == 2:
        dout[:, 1] = -dout[:, 0]
    pack = hip.mlp_pack_w2_f16(w2)
    cells = {}
    if hip.mlp_forward_f16_supports(d, n):
        cells["fwd"] = timed(lambda: hip.mlp_tower_forward_split(x, w1, b1, pack, b2, w3, b3))
        cells["fwd+bits"] = timed(lambda: hip.mlp_tower_forward_split(x, w1, b1, pack, b2, w3, b3, save=True, save_h1=False,
                                                                     save_gate=True, save_h2=False))
        cells["fwd+h2"] = timed(lambda: hip.mlp_tower_forward_split(x, w1, b1, pack, b2, w3, b3, save=True, save_h1=False,
                                                                   save_gate=True))
    if hip.mlp_backward_f16_supports(d, n):
        _, _, h2, gate = hip.mlp_tower_forward_split(x, w1, b1, pack, b2, w3, b3, save=True, save_h1=False, save_gate=True)
        w2t = hip.mlp_pack_w2_f16(w2, transposed=True)
        hip.timer.reset()
        hip.timer.enabled = True
        for _ in range(args.reps):
            if n <= 2:
                gp = hip.mlp_pack_w2_f16_gate(w2, w3)
                hip.mlp_tower_backward(x, None, None, dout, w2t, w3, w1, b1, gate2=gate, gate_pack=lambda: gp, w2=w2, b2=b2,
                                       assume_pair=n == 2)
            hip.mlp_tower_backward(x, None, h2, dout if n != 2 else torch.randn_like(dout) / m, w2t, w3, w1, b1, gate2=gate,
                                   assume_general=True)
        torch.cuda.synchronize()
        s = hip.timer.summary()
        hip.timer.enabled = False
        for key, name in (("dgrad_gate", "mlp_tower_backward_gate"), ("dgrad", "mlp_tower_backward"),
                          ("wgrad_gate", "mlp_wgrad_gate"), ("wgrad", "mlp_wgrad")):
            if name in s:
                cells[key] = s[name]["avg_ms"] * 1e3
        del h2, gate
    fmt = lambda k: f"{cells[k]:.0f}" if k in cells else "-"  # noqa: E731
    print(f"{spec:>7} {fmt('fwd'):>8} {fmt('fwd+bits'):>9} {fmt('fwd+h2'):>8} {fmt('dgrad_gate'):>11} {fmt('dgrad'):>8}"
          f" {fmt('wgrad_gate'):>11} {fmt('wgrad'):>8}", flush=True)
