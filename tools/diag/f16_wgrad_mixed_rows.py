"""Host model (CPU, fp64 accumulation) of a weight gradient dW2 = dZ2^T h1 on scaled fp16
planes with ONE power of two per operand and launch -- the only scaling that can be taken
out of a sum over samples -- against six bf16 plane products, with rows 10^6 apart in
magnitude.  The GPU kernel built on this model (removed; profiles/r02_f16_wgrad_mixed_rows.txt
is its measurement) reproduced the model's worst entry to four digits: 6.06e-5 of the
entry's own sum of |terms|, against 7e-7 for bf16 planes.  Why the weight gradient stays on
bf16 planes while forward and data gradient (per-ROW powers of two) moved to fp16."""
import math

import torch

torch.manual_seed(23)
m, d_in, n_out = 20_000, 3, 2
x = torch.randn(m, d_in) * 3
w1, b1 = torch.randn(256, d_in) * 0.5, torch.randn(256) * 0.1
w2, b2 = torch.randn(256, 256) / 16, torch.randn(256) * 0.1
w3 = torch.randn(n_out, 256) / 16
dout = torch.randn(m, n_out) / m
dout *= 10.0 ** torch.randint(-4, 3, (m, 1)).float()
x *= 10.0 ** torch.randint(-2, 2, (m, 1)).float()
h1 = torch.relu(x @ w1.T + b1)
h2 = torch.relu(h1 @ w2.T + b2)
dz = (dout @ w3) * (h2 > 0)
want = dz.double().T @ h1.double()
size = dz.double().abs().T @ h1.double()


def f16_planes(v):
    hi = v.half()
    return hi.double(), (v - hi.float()).half().double()


def bf16_planes(v):
    out = []
    for _ in range(3):
        p = (v.view(torch.int32) & -65536).view(torch.float32)
        out.append(p.double())
        v = v - p
    return out


bound_dz = (dout.abs().amax(0) * w3.abs().amax(1)).sum()
bound_h = b1.abs().max() + (x.abs().amax(0) * w1.abs().amax(0)).sum()
s_dz, s_h = 2.0 ** (14 - math.frexp(float(bound_dz))[1]), 2.0 ** (14 - math.frexp(float(bound_h))[1])
ah, al = f16_planes(dz * s_dz)
bh, bl = f16_planes(h1 * s_h)
f16 = (ah.T @ bh + ah.T @ bl + al.T @ bh) / (s_dz * s_h)
a0, a1, a2 = bf16_planes(dz)
c0, c1, c2 = bf16_planes(h1)
bf16 = a0.T @ c0 + a0.T @ c1 + a1.T @ c0 + a1.T @ c1 + a0.T @ c2 + a2.T @ c0
for name, got in (("fp16 x2, launch-wide powers of two", f16), ("bf16 x3", bf16)):
    rel = (got - want).abs() / (size + size.max() * 1e-30)
    print(f"{name}: worst entry {rel.max():.2e} of its own sum |terms|; {(got - want).abs().max() / want.abs().max():.2e} of max |dW2|;"
          f" entries worse than 1e-6: {float((rel > 1e-6).double().mean()):.4%}")
