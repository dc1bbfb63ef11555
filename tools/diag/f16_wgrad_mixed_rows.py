"""Host model (CPU, fp64 accumulation) of a weight gradient dW2 = dZ2^T h1 on scaled fp16
planes against six bf16 plane products, with rows 10^6 apart in magnitude.  The sum runs
over samples, so only factors whose product is the same for every sample can be taken out
of it; three placements are modelled: one power of two per operand and launch, per-sample
factors balanced between the operands, and those plus a power of two per column (rank-1
bounds).  Two of them were built as kernels and measured in round 2 (both removed;
profiles/r02_f16_wgrad_mixed_rows.txt): the launch-wide one reproduced this model's worst
entry to four digits.  All three leave entries made of a few small-row terms (nearly dead
units) 10^-5 .. 10^-4 off in RELATIVE terms where bf16 planes hold 3e-7 -- and the kernels
were no faster (operand production on the VALU bounds them).  Why the weight gradient stays
on bf16 planes while forward and data gradient (dot products along a scaled row) moved to fp16."""
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
# balanced per-sample powers of two: a(s) + b(s) = 28 - E, each operand D(s)/2 below fp16's top
ed = torch.tensor([math.frexp(float(v))[1] for v in (dout.abs() * w3.abs().amax(1)).sum(1)])
eh = torch.tensor([math.frexp(float(v))[1] for v in b1.abs().max() + (x.abs() * w1.abs().amax(0)).sum(1)])
top = int((ed + eh).max())
deficit = top - ed - eh
half = deficit // 2
sa, sb = 2.0 ** (14 - ed - half).double(), 2.0 ** (14 - eh - (deficit - half)).double()
ah, al = f16_planes((dz.double() * sa[:, None]).float())
bh, bl = f16_planes((h1.double() * sb[:, None]).float())
f16_balanced = (ah.T @ bh + ah.T @ bl + al.T @ bh) / 2.0 ** (28 - top)
# rank-1 bounds |dZ2[s][j]| <= (sum_q |dOut[s][q]|) max_q |W3[q][j]|, |h1[s][i]| <= (1 + sum_c |x[s][c]|) max(|b1[i]|, max_c |W1[i][c]|):
# a power of two per COLUMN of each operand as well (taken out per row / column of dW2), rows balanced as above
exp_of = lambda t: torch.tensor([math.frexp(float(v))[1] for v in t])
er, erh = exp_of(dout.abs().sum(1)), exp_of(1 + x.abs().sum(1))
ec, ech = exp_of(w3.abs().amax(0)), exp_of(torch.maximum(b1.abs(), w1.abs().amax(1)))
top1 = int((er + erh).max())
deficit = top1 - er - erh
half = deficit // 2
ra, rb = 2.0 ** (14 - er - half).double(), 2.0 ** (14 - erh - (deficit - half)).double()
ah, al = f16_planes((dz.double() * ra[:, None] * 2.0 ** (-ec.double())[None, :]).float())
bh, bl = f16_planes((h1.double() * rb[:, None] * 2.0 ** (-ech.double())[None, :]).float())
f16_rank1 = (ah.T @ bh + ah.T @ bl + al.T @ bh) / 2.0 ** (28 - top1) * 2.0 ** ec.double()[:, None] * 2.0 ** ech.double()[None, :]
a0, a1, a2 = bf16_planes(dz)
c0, c1, c2 = bf16_planes(h1)
bf16 = a0.T @ c0 + a0.T @ c1 + a1.T @ c0 + a1.T @ c1 + a0.T @ c2 + a2.T @ c0
for name, got in (("fp16 x2, launch-wide powers of two", f16), ("fp16 x2, per-sample balanced powers of two", f16_balanced),
                  ("fp16 x2, balanced per sample + per column (rank-1 bounds)", f16_rank1),
                  ("bf16 x3", bf16)):
    rel = (got - want).abs() / (size + size.max() * 1e-30)
    print(f"{name}: worst entry {rel.max():.2e} of its own sum |terms|; {(got - want).abs().max() / want.abs().max():.2e} of max |dW2|;"
          f" entries worse than 1e-6: {float((rel > 1e-6).double().mean()):.4%}")

if __name__ == "__main__" and __import__("os").environ.get("WORST"):
    rel = (f16_rank1 - want).abs() / (size + size.max() * 1e-30)
    j, i = divmod(int(rel.flatten().argmax()), 256)
    terms = dz.double()[:, j] * h1.double()[:, i]
    order = terms.abs().argsort(descending=True)[:6]
    print("worst entry", j, i, float(rel[j, i]), "size/max", float(size[j, i] / size.max()), "ec", int(ec[j]), "ech", int(ech[i]))
    for r in order.tolist():
        print(f" row {r}: term {float(terms[r]):.3e} deficit {int(deficit[r])} dz/bound {float(dz[r, j].abs() / (2.0 ** (er[r] + ec[j]))):.2e}"
              f" h/bound {float(h1[r, i] / (2.0 ** (erh[r] + ech[i]))):.2e} dout scale {float(dout[r].abs().max() * m):.1e} x {float(x[r].abs().max()):.1e}")
