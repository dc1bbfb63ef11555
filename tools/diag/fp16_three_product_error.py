"""would a THREE-product scheme on fp16 planes (hi + lo, lo
scaled by 2^11, error-corrected a la tensor-core SGEMM emulation) meet the accuracy bars the
six-product bf16-plane kernels are held to?  Host-side model, same setting as DESIGN 3.1's
table: K = 256 dot products, activations relu(N(0,1)), weights U(+-1/16), 131 072 outputs,
fp32 accumulation modelled per 16-k block (one MFMA), against fp64.

    python tools/diag/fp16_three_product_error.py
"""
import numpy as np

rng = np.random.default_rng(0)
K, N = 256, 131072
a = np.maximum(rng.standard_normal((N, K)), 0).astype(np.float32)
w = rng.uniform(-1 / 16, 1 / 16, (N, K)).astype(np.float32)
ref = (a.astype(np.float64) * w.astype(np.float64)).sum(1)


def blocks(prod_terms):
    """sum over K of the given per-element product terms, fp32 accumulate per 16-k block (the
    16 products of a block are summed exactly -- the MFMA's internal adder tree is wider than
    fp32 -- and rounded once into the fp32 accumulator)."""
    acc = np.zeros(N, np.float32)
    for k0 in range(0, K, 16):
        s = np.zeros(N, np.float64)
        for t in prod_terms:
            s += t[:, k0:k0 + 16].sum(1)
        acc = (acc.astype(np.float64) + s).astype(np.float32)
    return acc


def report(name, got):
    err = np.abs(got.astype(np.float64) - ref)
    print(f"{name:58s} max abs err {err.max():.2e}   mean abs err {err.mean():.2e}")


# fp32 fma chain (what the fp32 kernels / rocBLAS do)
acc = np.zeros(N, np.float32)
for k in range(K):
    acc = (acc.astype(np.float64) + a[:, k].astype(np.float64) * w[:, k].astype(np.float64)).astype(np.float32)
report("fp32 fma chain", acc)


def bf16_trunc(x):
    return (x.view(np.uint32) & 0xFFFF0000).view(np.float32)


ah = bf16_trunc(a); am = bf16_trunc(a - ah); al = a - ah - am
wh = bf16_trunc(w); wm = bf16_trunc(w - wh); wl = w - wh - wm
d = np.float64
six = [ah.astype(d) * wh, ah.astype(d) * wm, am.astype(d) * wh, ah.astype(d) * wl, am.astype(d) * wm, al.astype(d) * wh]
report("6 bf16 plane products (shipped kernels)", blocks(six))

# fp16 two-plane split, lo scaled by 2^11 (Ootomo & Yokota style), per-matrix power-of-two
# scaling of both operands into fp16's comfortable range
def fp16_split(x, scale):
    xs = x.astype(np.float64) * scale
    hi = xs.astype(np.float16).astype(np.float64)
    lo = ((xs - hi) * 2048.0).astype(np.float16).astype(np.float64)
    return hi, lo


for sa, sw, label in ((1.0, 1.0, "no scaling"), (64.0, 2.0 ** 13, "operands scaled to ~2^9")):
    a_hi, a_lo = fp16_split(a, sa)
    w_hi, w_lo = fp16_split(w, sw)
    main = blocks([a_hi * w_hi])
    corr = blocks([a_hi * w_lo, a_lo * w_hi])
    got = ((main.astype(np.float64) + corr.astype(np.float64) / 2048.0) / (sa * sw)).astype(np.float32)
    report(f"3 fp16 products, separate correction accumulator, {label}", got)
    both = blocks([a_hi * w_hi, (a_hi * w_lo + a_lo * w_hi) / 2048.0])
    report(f"3 fp16 products, ONE accumulator (lo planes pre-divided), {label}", (both.astype(np.float64) / (sa * sw)).astype(np.float32))
