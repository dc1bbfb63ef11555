"""OPT-IN (``RL8_AMD_TOWERS=piecewise``), round-4 prototype: the default towers of a SCALAR observation as exact
piecewise-linear tables (kernels: csrc/piecewise_kernels.hip).

``Linear(1, 256) -> ReLU -> Linear(256, 256) -> ReLU -> Linear(256, n_out)`` (reference
``src/rl8/models/_feedforward.py:336-375`` with a one-dimensional observation: the dummy envs of BASELINE configs 2
and 4) is piecewise linear in its input.  Between two consecutive breakpoints both layers' ReLU gates are constant, so

* the forward pass is ``out = value_p + slope_p (x - anchor_p)`` with ``p`` the interval of ``x``;
* every parameter gradient is a small fp64 product of the intervals' gate patterns with the per-interval sums
  ``S0_p = sum dOut``, ``S1_p = sum dOut x`` over the rows of the interval (``rl8_pw_segment_sums_f32``, exact):

      dW2 = Z1^T (M1 * w1) + Z0^T (M1 * b1),   Z0 = G2 * (S0 W3),  Z1 = G2 * (S1 W3)       (P + 1 rows each)
      db2 = sum_p Z0_p,   dW3 = S1^T (G2 * A) + S0^T (G2 * B),   db3 = sum_p S0_p,
      dW1 = sum_p M1_p * (Z1_p W2),   db1 = sum_p M1_p * (Z0_p W2)

  with ``M1`` / ``G2`` the layer-1 / layer-2 gates of interval ``p`` and ``z2 = A_p x + B_p`` its layer-2
  pre-activations.

The table is rebuilt in fp64 whenever a weight changed (version counters), here with torch ops: a prototype's
plumbing -- a fused build kernel is future work -- and so is the gradient algebra above (a few [P, 256] x [256, 256] fp64
products).  What touches the rows is HIP.  Never on unless asked for, never for d_in != 1, and it steps aside
(returns None) when the table would not fit the kernels' LDS: the general tower kernels are the product.
"""

from __future__ import annotations

import os
from typing import Sequence

import torch
import torch.nn as nn

from .. import hip

ENABLED = os.environ.get("RL8_AMD_TOWERS", "") == "piecewise"

#: towers evaluated from a table / tables built, for tests and the bench line
stats = {"forwards": 0, "backwards": 0, "tables_built": 0, "declined": 0}


class Table:
    """Breakpoints and per-interval data of one tower at one set of weights."""

    def __init__(self, breaks: torch.Tensor, flat: torch.Tensor, m1, g2, a, bb, n_out: int) -> None:
        self.breaks, self.flat, self.p, self.n_out = breaks, flat, int(breaks.numel()), n_out
        self.m1, self.g2, self.a, self.bb = m1, g2, a, bb      # [P + 1, 256] fp64 each


def build_table(w1, b1, w2, b2, w3, b3) -> None | Table:
    w1, b1, w2, b2, w3, b3 = (t.detach().double() for t in (w1, b1, w2, b2, w3, b3))
    w1 = w1[:, 0]
    live = w1 != 0
    k1 = torch.sort(-b1[live] / w1[live]).values                       # layer-1 kinks
    k1 = k1[torch.isfinite(k1)]
    inf = k1.new_tensor([float("inf")])
    if k1.numel() == 0:
        k1 = k1.new_tensor([0.0])                                      # (no kink at all: one artificial breakpoint)
    edges = torch.cat([-inf, k1, inf])
    mids = torch.cat([k1[:1] - 1.0, 0.5 * (k1[1:] + k1[:-1]), k1[-1:] + 1.0])
    m = ((mids[:, None] * w1[None, :] + b1[None, :]) > 0).double()     # [S, 256] layer-1 gates per segment
    a = (m * w1) @ w2.T                                                # z2 = a x + bb on the segment
    bb = (m * b1) @ w2.T + b2
    tau = -bb / a
    inside = (a != 0) & (tau > edges[:-1, None]) & (tau < edges[1:, None])
    breaks = torch.unique_consecutive(torch.sort(torch.cat([k1, tau[inside]])).values)
    # (breakpoints that coincide in fp32 are one breakpoint for an fp32 x)
    breaks32 = torch.unique_consecutive(breaks.float())
    p = int(breaks32.numel())
    n_out = w3.shape[0]
    if p > hip.pw_max_breaks() or (p + 1) * n_out * 32 + p * 4 > 150 * 1024:
        return None
    breaks = breaks32.double()
    inner = torch.cat([breaks[:1] - 1.0, 0.5 * (breaks[1:] + breaks[:-1]), breaks[-1:] + 1.0])
    seg = torch.searchsorted(k1, inner)
    a_p, bb_p, m_p = a[seg], bb[seg], m[seg]
    g2 = ((a_p * inner[:, None] + bb_p) > 0).double()
    slope = (g2 * a_p) @ w3.T
    icpt = (g2 * bb_p) @ w3.T + b3
    anchor = torch.cat([breaks[:1], breaks])
    value = slope * anchor[:, None] + icpt
    flat = torch.cat([breaks32, anchor.float(), value.float().reshape(-1), slope.float().reshape(-1)]).contiguous()
    stats["tables_built"] += 1
    return Table(breaks32.contiguous(), flat, m_p, g2, a_p, bb_p, n_out)


def _table_of(l1: nn.Linear, l2: nn.Linear, heads: Sequence[nn.Linear]) -> None | Table:
    params = [l1.weight, l1.bias, l2.weight, l2.bias, *[h.weight for h in heads], *[h.bias for h in heads]]
    key = tuple((t._version, t.data_ptr()) for t in params)
    hit = l2.__dict__.get("_rl8_pw_table")
    if hit is not None and hit[0] == key:
        return hit[1]
    w3 = heads[0].weight if len(heads) == 1 else torch.cat([h.weight for h in heads], 0)
    b3 = heads[0].bias if len(heads) == 1 else torch.cat([h.bias for h in heads], 0)
    table = build_table(l1.weight, l1.bias, l2.weight, l2.bias, w3, b3)
    l2.__dict__["_rl8_pw_table"] = (key, table)
    return table


class _PiecewiseTower(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, w3, b3, table, grad_mode):  # type: ignore[override]
        out = hip.pw_tower_forward(x, table.flat, table.p, table.n_out)
        stats["forwards"] += 1
        if grad_mode and any(ctx.needs_input_grad[1:7]):
            ctx.table = table
            ctx.save_for_backward(x, w1, b1, w2, b2, w3, b3)
        return out

    @staticmethod
    def backward(ctx, dout):  # type: ignore[override]
        x, w1, b1, w2, b2, w3, b3 = ctx.saved_tensors
        t = ctx.table
        stats["backwards"] += 1
        sums = hip.pw_segment_sums(x, dout.contiguous().float(), t.breaks, t.p)
        s0, s1 = sums[:, 0], sums[:, 1]                              # [P + 1, n_out]
        w3d, w2d = w3.double(), w2.double()
        z0, z1 = t.g2 * (s0 @ w3d), t.g2 * (s1 @ w3d)                # [P + 1, 256]
        dw2 = z1.T @ (t.m1 * w1.double()[:, 0]) + z0.T @ (t.m1 * b1.double())
        dw3 = s1.T @ (t.g2 * t.a) + s0.T @ (t.g2 * t.bb)
        dw1 = (t.m1 * (z1 @ w2d)).sum(0)[:, None]
        db1 = (t.m1 * (z0 @ w2d)).sum(0)
        return (None, dw1.float(), db1.float(), dw2.float(), z0.sum(0).float(), dw3.float(), s0.sum(0).float(), None, None)


def tower_forward(l1: nn.Linear, l2: nn.Linear, heads: Sequence[nn.Linear], x: torch.Tensor) -> None | torch.Tensor:
    """``cat([head(trunk(x))])`` from the tower's table, or None (not a scalar observation, table too large)."""
    if x.shape[1] != 1 or l1.in_features != 1:
        return None
    table = _table_of(l1, l2, heads)
    if table is None:
        stats["declined"] += 1
        return None
    w3 = heads[0].weight if len(heads) == 1 else torch.cat([h.weight for h in heads], 0)
    b3 = heads[0].bias if len(heads) == 1 else torch.cat([h.bias for h in heads], 0)
    return _PiecewiseTower.apply(x.contiguous(), l1.weight, l1.bias, l2.weight, l2.bias, w3, b3, table, torch.is_grad_enabled())
