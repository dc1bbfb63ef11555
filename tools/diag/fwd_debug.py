"""Diagnostic (GPU): which outputs of the bf16-plane forward kernel differ from fp64, and where."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rl8_amd import hip
DEV = "cuda"
m, d_in, n_out = int(sys.argv[1]) if len(sys.argv) > 1 else 300, 1, 2
g = torch.Generator(device=DEV).manual_seed(1)
x = torch.randn(m, d_in, device=DEV, generator=g) * 3
p = {"w1": torch.randn(256, d_in, device=DEV, generator=g), "b1": torch.randn(256, device=DEV, generator=g),
     "w2": torch.randn(256, 256, device=DEV, generator=g) / 16, "b2": torch.randn(256, device=DEV, generator=g) * .1,
     "w3": torch.randn(n_out, 256, device=DEV, generator=g) / 16, "b3": torch.randn(n_out, device=DEV, generator=g)}
pd = {k: v.double() for k, v in p.items()}
h1 = torch.relu(x.double() @ pd["w1"].T + pd["b1"]); h2 = torch.relu(h1 @ pd["w2"].T + pd["b2"]); out = h2 @ pd["w3"].T + pd["b3"]
w2s = hip.mlp_pack_w2_split(p["w2"])
o, _, h2g, gate = hip.mlp_tower_forward_split(x, p["w1"], p["b1"], w2s, p["b2"], p["w3"], p["b3"], save=True, save_h1=False, save_gate=True)
torch.cuda.synchronize()
e2 = (h2g.double() - h2).abs()
print("h2 max err", float(e2.max()), "rows wrong", int((e2.max(1).values > 1e-4).sum()), "of", m)
bad = (e2 > 1e-4).nonzero()[:10].tolist(); print("  first bad (row, col):", bad)
eo = (o.double() - out).abs(); print("out max err", float(eo.max()), "rows wrong", int((eo.max(1).values > 1e-4).sum()))
print("  first bad rows:", (eo.max(1).values > 1e-4).nonzero().flatten()[:20].tolist())
# head recomputed from the kernel's own h2
out2 = h2g.double() @ pd["w3"].T + pd["b3"]; print("out vs head(own h2) max err", float((o.double() - out2).abs().max()))
want_gate = (h2g > 0)
bits = ((gate.view(m, 8).to(torch.int64)[:, :, None] >> torch.arange(32, device=DEV)) & 1).reshape(m, 256).bool()
print("gate mismatches", int((bits != want_gate).sum()), "first:", (bits != want_gate).nonzero()[:8].tolist())
for r in (0, 1, 31, 32, 33, 64, 127, 128, m - 1):
    if r < m: print(r, o[r].tolist(), out[r].tolist())
# where do the stored values belong?  pre-activation without bias, to test the bias-index hypothesis
pre = h1 @ pd["w2"].T
for (r, c) in [(0, 3), (0, 5), (0, 6), (0, 15), (1, 3), (40, 200), (100, 77)]:
    v = float(h2g[r, c])
    hit = ((h2 - v).abs() < 2e-5).nonzero()[:4].tolist()
    # which bias index would explain it
    diff = v - float(pre[r, c])
    bidx = ((pd["b2"] - diff).abs() < 2e-5).nonzero().flatten().tolist()
    print((r, c), "got", v, "want", float(h2[r, c]), "same value found at", hit, "bias index explaining it", bidx)
