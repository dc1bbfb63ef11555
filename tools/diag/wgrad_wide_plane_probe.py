"""Which terms the fp16-plane gate-bits weight gradient gets right: rows with dOut = 2^-k of the call's largest,
one magnitude class at a time (every class's rows alone carry an entry, so its error is the class's own)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from rl8_amd import hip  # noqa: E402

DEV = "cuda:0"
g = torch.Generator(device=DEV).manual_seed(1)
m, d_in, n_out = 4096, 1, 1
x = torch.randn(m, d_in, device=DEV, generator=g) * 40
p = {"w1": torch.randn(256, d_in, device=DEV, generator=g) * 0.5, "b1": torch.randn(256, device=DEV, generator=g) * 0.1,
     "w2": torch.randn(256, 256, device=DEV, generator=g) / 16, "b2": torch.randn(256, device=DEV, generator=g) * 0.1,
     "w3": torch.randn(n_out, 256, device=DEV, generator=g) / 16, "b3": torch.randn(n_out, device=DEV, generator=g)}
w2p, w2t = hip.mlp_pack_w2_f16(p["w2"]), hip.mlp_pack_w2_f16(p["w2"], transposed=True)
_, h1, h2, gate = hip.mlp_tower_forward_split(x, p["w1"], p["b1"], w2p, p["b2"], p["w3"], p["b3"], save=True, save_gate=True)
gate_pack = lambda: hip.mlp_pack_w2_f16_gate(p["w2"], p["w3"])  # noqa: E731
for k in (0, 4, 8, 12, 16, 20, 24, 28, 32, 36):
    d = torch.zeros(m, 1, device=DEV)
    d[0] = 1.0                       # the row that sets the bound; its gate row is removed from the comparison below
    rows = torch.arange(64, m, device=DEV)
    d[rows, 0] = (torch.rand(len(rows), device=DEV, generator=g) + 0.5) * 2.0 ** -k
    out = {}
    for mode in ("f16!", "bf16"):
        os.environ["RL8_WGRAD_GATE_PLANES"] = mode
        out[mode] = hip.mlp_tower_backward(x, None, None, d, w2t, p["w3"], p["w1"], p["b1"], gate2=gate, gate_pack=gate_pack,
                                           w2=p["w2"], b2=p["b2"])["w2"].double()
    dd = d.double().clone()
    dd[0] = 0.0                      # compare the small rows' share only
    dz2 = (dd @ p["w3"].double()) * (h2 > 0)
    want, size = dz2.T @ h1.double(), dz2.abs().T @ h1.double()
    first = ((d.double()[:1] @ p["w3"].double()) * (h2[:1] > 0)).T @ h1[:1].double()
    for mode, got in out.items():
        err = ((got - first - want).abs() / (size + 1e-300))
        print(f"dOut 2^-{k:2d} of max  {mode:5s}: entrywise error of the small rows' sum  median {float(err.median()):.2e}  max {float(err.max()):.2e}")

# The guard test's own setting: a third of the rows 2^-20 down, m = 150 000 (many steps per workgroup)
m = 150_000
x = torch.randn(m, d_in, device=DEV, generator=g) * 40
_, h1, h2, gate = hip.mlp_tower_forward_split(x, p["w1"], p["b1"], w2p, p["b2"], p["w3"], p["b3"], save=True, save_gate=True)
g0 = torch.randn(m, device=DEV, generator=g) / m
g0[torch.rand(m, device=DEV, generator=g) < 0.33] *= 2.0 ** -20
d = g0[:, None].contiguous()
dz2 = (d.double() @ p["w3"].double()) * (h2 > 0)
want, size = dz2.T @ h1.double(), dz2.abs().T @ h1.double()
for mode in ("f16!", "bf16"):
    os.environ["RL8_WGRAD_GATE_PLANES"] = mode
    got = hip.mlp_tower_backward(x, None, None, d, w2t, p["w3"], p["w1"], p["b1"], gate2=gate, gate_pack=gate_pack,
                                 w2=p["w2"], b2=p["b2"])["w2"].double()
    err = (got - want).abs() / (size + 1e-300)
    at = int(err.argmax())
    j, i = at // 256, at % 256
    print(f"many small rows, m = {m}, {mode}: median {float(err.median()):.2e} p99 {float(err.flatten().quantile(0.99)):.2e} max {float(err.max()):.2e}"
          f" at ({j}, {i}): size / largest size {float(size[j, i] / size.max()):.2e}, rows of the entry"
          f" {int(((h2[:, j] > 0) & (h1[:, i] > 0)).sum())}, w3[j] {float(p['w3'][0, j]):.3e}")
