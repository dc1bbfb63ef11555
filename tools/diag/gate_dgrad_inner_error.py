"""Inner-product accuracy of the data-gradient kernels, general fp16 planes vs gate mode: single-row
batches (dW1[i] = dZ1[0][i] * x, so dZ1 is visible), error of dH1[0][i] = sum_k dZ2[k] W2[k][i] relative
to sum_k |terms| (the conditioning-free yardstick), over seeds."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rl8_amd import hip

DEV = torch.device("cuda:0")
worst = {"general": 0.0, "gate": 0.0, "bf16x3": 0.0}
for seed in range(200):
    g = torch.Generator(device=DEV).manual_seed(seed)
    d_in, n_out = 1, 1
    x = torch.randn(1, d_in, device=DEV, generator=g) * 3
    p = {"w1": torch.randn(256, d_in, device=DEV, generator=g) * 0.5, "b1": torch.randn(256, device=DEV, generator=g) * 0.1,
         "w2": torch.randn(256, 256, device=DEV, generator=g) / 16, "b2": torch.randn(256, device=DEV, generator=g) * 0.1,
         "w3": torch.randn(n_out, 256, device=DEV, generator=g) / 16, "b3": torch.randn(n_out, device=DEV, generator=g)}
    dout = torch.randn(1, 1, device=DEV, generator=g)
    _, _, h2, gate = hip.mlp_tower_forward_split(x, p["w1"], p["b1"], hip.mlp_pack_w2_f16(p["w2"]), p["b2"], p["w3"], p["b3"],
                                                 save=True, save_h1=False, save_gate=True)
    h1 = torch.relu(x @ p["w1"].T + p["b1"])
    dz2 = (dout.double() @ p["w3"].double()) * (h2.double() > 0)          # [1, 256]
    terms = dz2[0][:, None] * p["w2"].double()                             # [k, i]
    want = terms.sum(0) * (h1[0].double() > 0)
    size = terms.abs().sum(0) + 1e-300
    for name, pack, gp in (("general", hip.mlp_pack_w2_f16, None),
                           ("gate", hip.mlp_pack_w2_f16, lambda: hip.mlp_pack_w2_f16_gate(p["w2"], p["w3"]))):
        got = hip.mlp_tower_backward(x, None, h2, dout, pack(p["w2"], transposed=True), p["w3"], p["w1"], p["b1"], gate2=gate,
                                     gate_pack=gp)["b1"].double()          # m = 1: db1 = dZ1 itself
        worst[name] = max(worst[name], float(((got - want).abs() / size).max()))
print({k: f"{v:.2e}" for k, v in worst.items()}, "(fp32 eps 6.0e-08)")
