"""PROTOTYPE, not product: a two-layer ReLU tower of a SCALAR observation

    out(x) = W3 relu(W2 relu(w1 x + b1) + b2) + b3,        x in R,

is a piecewise-linear function of x with at most 256 layer-1 kinks (x = -b1_i / w1_i) and, inside each of the 257
segments between them, at most 256 layer-2 kinks (where a pre-activation A_jk x + B_jk of that segment changes sign:
A_j = W2 (w1 * m_j), B_j = W2 (b1 * m_j) + b2 with m_j the layer-1 gate pattern of segment j).  So the whole tower is a
table of <= 66 049 intervals, each with one slope and one intercept per output, rebuilt in fp64 whenever the weights
change (17 MFLOP), and a forward pass is a search of x in the sorted breakpoints plus one fma per output: ~20 bytes and a
handful of instructions per row where the matrix kernels execute 3 x 131 072 FLOP per row.  Exact algebra, special to
d_in = 1 (the dummy envs of BASELINE configs 2 and 4); it must never replace the general kernels.

This script measures, on a GPU, with torch ops only (sort / searchsorted / gather: plumbing for the measurement):
  * accuracy of the table against an fp64 evaluation of the tower and against the shipped fp16-plane kernel,
  * time of the table build, of a forward over 2^20 unsorted rows, and of the same over rows sorted once,
beside the shipped kernel's time on the same rows.  The backward pass is sketched in DESIGN.md section 9 (with rows sorted by
x once per step(), the gate of unit k inside segment j is "x beyond tau_jk", so every sum the weight gradient needs is a
difference of two prefix sums of dOut x and dOut over the sorted rows, taken at 66 049 thresholds).

    python tools/proto/piecewise_tower.py [--rows 1048576] [--out profiles/r04_piecewise_tower_prototype.json]
"""

import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

import torch  # noqa: E402

from rl8_amd import hip  # noqa: E402

p = argparse.ArgumentParser()
p.add_argument("--rows", type=int, default=1 << 20)
p.add_argument("--out", default="profiles/r04_piecewise_tower_prototype.json")
args = p.parse_args()
DEV = "cuda:0"


def build_table(w1, b1, w2, b2, w3, b3):
    """(breakpoints [P] fp64 ascending, left value [P + 1, n_out], slope [P + 1, n_out]) in fp64: interval i is
    (breaks[i - 1], breaks[i]]; out(x) = value[i] + slope[i] * (x - anchor[i]) with anchor = the interval's left end
    (the first interval is anchored at its right end)."""
    w1, b1, w2, b2, w3, b3 = (t.double() for t in (w1, b1, w2, b2, w3, b3))
    w1 = w1[:, 0]
    live = w1 != 0
    k1 = torch.sort(-b1[live] / w1[live]).values                       # layer-1 kinks
    edges = torch.cat([k1.new_tensor([-float("inf")]), k1, k1.new_tensor([float("inf")])])
    mids = torch.cat([k1[:1] - 1.0, 0.5 * (k1[1:] + k1[:-1]), k1[-1:] + 1.0])  # one point inside each segment
    m = (mids[:, None] * w1[None, :] + b1[None, :]) > 0                # [S, 256] layer-1 gates per segment
    a = (m * w1) @ w2.T                                                # [S, 256]: z2 = a x + bb on the segment
    bb = (m * b1) @ w2.T + b2
    tau = -bb / a                                                      # layer-2 kinks, where they fall inside their segment
    inside = (a != 0) & (tau > edges[:-1, None]) & (tau < edges[1:, None])
    breaks = torch.sort(torch.cat([k1, tau[inside]])).values
    breaks = torch.unique_consecutive(breaks)
    # one point inside each final interval -> its segment, its layer-2 gate pattern, its slope / intercept per output
    inner = torch.cat([breaks[:1] - 1.0, 0.5 * (breaks[1:] + breaks[:-1]), breaks[-1:] + 1.0])
    seg = torch.searchsorted(k1, inner)                                # segment index of each interval
    g2 = (a[seg] * inner[:, None] + bb[seg]) > 0                       # [P + 1, 256]
    slope = (g2 * a[seg]) @ w3.T                                       # [P + 1, n_out]
    icpt = (g2 * bb[seg]) @ w3.T + b3
    anchor = torch.cat([breaks[:1], breaks])                           # left end (first interval: its right end)
    value = slope * anchor[:, None] + icpt
    return breaks, anchor, value, slope


def table_forward(x, breaks32, anchor32, value32, slope32):
    idx = torch.searchsorted(breaks32, x[:, 0].contiguous())
    return value32[idx] + slope32[idx] * (x - anchor32[idx][:, None])


def timed(fn, rounds=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


report = {"rows": args.rows, "cases": {}}
for name, n_out, x_scale, trained in (("policy tower at init (n_out 2)", 2, 60.0, False), ("value tower at init", 1, 60.0, False),
                                      ("value tower, weights after 300 Adam steps on a regression", 1, 60.0, True)):
    g = torch.Generator(device=DEV).manual_seed(7 + n_out)
    tower = torch.nn.Sequential(torch.nn.Linear(1, 256), torch.nn.ReLU(), torch.nn.Linear(256, 256), torch.nn.ReLU(),
                                torch.nn.Linear(256, n_out)).to(DEV)
    if trained:  # weights that have moved: kinks spread, units died
        opt = torch.optim.Adam(tower.parameters(), 1e-3)
        for _ in range(300):
            xb = (torch.rand(4096, 1, device=DEV, generator=g) * 2 - 1) * 130
            loss = ((tower(xb) + xb.abs()) ** 2).mean()
            opt.zero_grad()
            loss.backward()
            opt.step()
    w1, b1, w2, b2, w3, b3 = (t.detach() for t in (tower[0].weight, tower[0].bias, tower[2].weight, tower[2].bias,
                                                   tower[4].weight, tower[4].bias))
    x = (torch.rand(args.rows, 1, device=DEV, generator=g) * 2 - 1) * 132          # the dummy env's states: U(-100, 100) + a walk
    want = tower.double()(x.double())
    tower.float()
    table64 = build_table(w1, b1, w2, b2, w3, b3)
    table32 = tuple(t.float().contiguous() for t in table64)
    got = table_forward(x, *table32)
    packed = hip.mlp_pack_w2_f16(w2)
    shipped = hip.mlp_tower_forward_split(x, w1, b1, packed, b2, w3, b3)[0]
    scale = float(want.abs().max())
    case = {
        "intervals": int(table64[0].numel()) + 1,
        "max_abs_err_over_max_abs_out": {"table_fp32": float((got.double() - want).abs().max()) / scale,
                                         "shipped_fp16_planes": float((shipped.double() - want).abs().max()) / scale,
                                         "torch_fp32": float((tower(x).double() - want).abs().max()) / scale},
        "us": {
            "table build (fp64, torch ops)": timed(lambda: build_table(w1, b1, w2, b2, w3, b3), 5),
            "forward, unsorted rows (searchsorted + 3 gathers + fma, torch ops)": timed(lambda: table_forward(x, *table32)),
            "shipped rows-per-wave fp16-plane forward": timed(lambda: hip.mlp_tower_forward_split(x, w1, b1, packed, b2, w3, b3)),
        },
    }
    xs = torch.sort(x[:, 0]).values[:, None].contiguous()
    case["us"]["forward, rows sorted by x beforehand"] = timed(lambda: table_forward(xs, *table32))
    case["us"]["sort of the rows by x (once per step(): the observations do not change between SGD iterations)"] = timed(
        lambda: torch.sort(x[:, 0]), 5)
    report["cases"][name] = case
    print(name, json.dumps(case, indent=1), flush=True)
os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
json.dump(report, open(args.out, "w"), indent=1)
print("wrote", args.out)
