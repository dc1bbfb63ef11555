"""Host-side model of a TWO-plane fp16 scheme for the tower products (3 MFMAs per fp32
product instead of 6): operands scaled by powers of two into fp16's range (rows of the
activation operand by a per-row bound, weights by one per-matrix factor), hi = fp16_rn(x),
lo = fp16_rn(x - hi) (fp16 subnormals kept), products hi*hi + hi*lo + lo*hi accumulated in
fp32 per 16-k block, scaled back at the end.  Distributions of tests/test_mlp_split_gpu.py.
"""
import numpy as np

rng = np.random.default_rng(1)


def split16(x64):
    hi = x64.astype(np.float16).astype(np.float64)
    lo = (x64 - hi).astype(np.float16).astype(np.float64)
    return hi, lo


def bf16_trunc(x):
    return (x.view(np.uint32) & 0xFFFF0000).view(np.float32)


def block_sum(terms, n, k):
    acc = np.zeros(n, np.float32)
    for k0 in range(0, k, 16):
        s = np.zeros(n, np.float64)
        for t in terms:
            s += t[:, k0:k0 + 16].sum(1)
        acc = (acc.astype(np.float64) + s).astype(np.float32)
    return acc


def products(a, b_rows, a_scale_rows, b_scale, label):
    """out[r] = sum_k a[r,k] * b_rows[r,k] (b_rows: the weight column each output row meets, one per row here)."""
    n, k = a.shape
    ref = (a.astype(np.float64) * b_rows.astype(np.float64)).sum(1)
    acc = np.zeros(n, np.float32)
    for j in range(k):
        acc = (acc.astype(np.float64) + a[:, j].astype(np.float64) * b_rows[:, j].astype(np.float64)).astype(np.float32)
    e32 = np.abs(acc - ref)
    ah = bf16_trunc(a); am = bf16_trunc(a - ah); al = a - ah - am
    bh = bf16_trunc(b_rows); bm = bf16_trunc(b_rows - bh); bl = b_rows - bh - bm
    d = np.float64
    six = block_sum([ah.astype(d) * bh, ah.astype(d) * bm, am.astype(d) * bh, ah.astype(d) * bl, am.astype(d) * bm, al.astype(d) * bh], n, k)
    e6 = np.abs(six - ref)
    a_hi, a_lo = split16(a.astype(d) * a_scale_rows[:, None])
    b_hi, b_lo = split16(b_rows.astype(d) * b_scale)
    assert np.isfinite(a_hi).all() and np.isfinite(b_hi).all(), "fp16 overflow"
    three = block_sum([a_hi * b_hi, a_hi * b_lo, a_lo * b_hi], n, k)
    got = (three.astype(d) / (a_scale_rows * b_scale)).astype(np.float32)
    e3 = np.abs(got - ref)
    sc = np.abs(ref).max()
    print(f"{label:34s} rel-to-max: fp32 chain {e32.max()/sc:.2e} (mean {e32.mean()/sc:.2e}) | 6 bf16 {e6.max()/sc:.2e} ({e6.mean()/sc:.2e})"
          f" | 3 fp16 {e3.max()/sc:.2e} ({e3.mean()/sc:.2e})")


def pow2_scale(bound, top=2.0 ** 14):
    return 2.0 ** np.floor(np.log2(top / np.maximum(bound, 1e-30)))


n, K = 65536, 256
for d_in, xs in ((1, 30.0), (5, 30.0), (1, 3.0)):
    x = (rng.standard_normal((n, d_in)) * xs).astype(np.float32)
    w1 = (rng.standard_normal((256, d_in)) * 0.5).astype(np.float32); b1 = (rng.standard_normal(256) * 0.1).astype(np.float32)
    h1 = np.maximum(x @ w1.T + b1, 0).astype(np.float32)
    w2col = (rng.standard_normal((n, K)) / 16).astype(np.float32)     # one weight row per output element
    bound = np.abs(b1).max() + (np.abs(x) * np.abs(w1).max(0)).sum(1)   # what a producer can form per row
    products(h1, w2col, pow2_scale(bound), pow2_scale(np.abs(w2col).max()), f"forward  d_in={d_in} x*{xs:g}")
# dgrad: dz2 = (dout @ w3) * gate, dout ~ N(0,1)/m
m = 33000
dout = (rng.standard_normal((n, 2)) / m).astype(np.float32); w3 = (rng.standard_normal((2, 256)) / 16).astype(np.float32)
gate = rng.random((n, 256)) < 0.5
dz2 = ((dout @ w3) * gate).astype(np.float32)
bound = (np.abs(dout) * np.abs(w3).max(1)).sum(1)
products(dz2, (rng.standard_normal((n, K)) / 16).astype(np.float32), pow2_scale(bound), pow2_scale(1 / 4.0), "dgrad    dz2 x W2")
# wgrad: sum over samples of dz2[s][j] * h1[s][i]: K = samples, scales are global (per launch)
S = 8192
dz = ((rng.standard_normal((S, 2)) / m).astype(np.float32) @ w3 * (rng.random((S, 256)) < 0.5)).astype(np.float32)   # [S][256]
x = (rng.standard_normal((S, 1)) * 3).astype(np.float32)
h = np.maximum(x @ (rng.standard_normal((256, 1)) * 0.5).astype(np.float32).T + 0.05, 0).astype(np.float32)            # [S][256]
A = dz.T[:64].copy(); B = h.T[:64].copy()   # 64 (j, i) pairs: rows = outputs, columns = samples
products(A, B, np.full(64, pow2_scale(np.abs(dz).max())), pow2_scale(np.abs(h).max()), f"wgrad    K = {S} samples")
