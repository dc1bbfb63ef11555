"""The rows-per-wave fp16-plane forward (mlp_rows_kernels.hip) against fp64 and against the bf16-plane kernel (an
independent implementation of the same tower): outputs / h2 as close to fp64 as that kernel is (x3), the gate bits equal
to h2 > 0 of the kernel's own h2, h1 bit-identical to the bf16-plane kernel's (same fma chain), all SAVE modes agreeing.

    python tools/diag/rows_forward_check.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from rl8_amd import hip

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(3)
bad = 0
for d_in, n_out in ((1, 2), (1, 1), (1, 3), (2, 2), (2, 1), (3, 1), (3, 3), (5, 3), (5, 1), (5, 2)):
    w1 = torch.randn(256, d_in, device=dev, generator=g)
    b1 = torch.randn(256, device=dev, generator=g)
    w2 = torch.randn(256, 256, device=dev, generator=g) / 16
    b2 = torch.randn(256, device=dev, generator=g)
    w3 = torch.randn(n_out, 256, device=dev, generator=g) / 16
    b3 = torch.randn(n_out, device=dev, generator=g)
    w2h, w2p = hip.mlp_pack_w2_f16(w2), hip.mlp_pack_w2(w2)
    for m in (1, 31, 100, 128, 129, 257, 1000, 4096 + 5, (1 << 17) + 77):
        x = torch.randn(m, d_in, device=dev, generator=g) * 30
        x[::7] *= 1e-3
        ref = hip.mlp_tower_forward(x, w1, b1, w2p, b2, w3, b3, save=True)  # the fp32-MFMA kernel as the yardstick
        h1 = torch.relu(x.double() @ w1.double().T + b1.double())
        h2 = torch.relu(h1 @ w2.double().T + b2.double())
        out64 = h2 @ w3.double().T + b3.double()
        got_inf = hip.mlp_tower_forward_split(x, w1, b1, w2h, b2, w3, b3)[0]
        got_gate = hip.mlp_tower_forward_split(x, w1, b1, w2h, b2, w3, b3, save=True, save_h1=False, save_gate=True, save_h2=False)
        got = hip.mlp_tower_forward_split(x, w1, b1, w2h, b2, w3, b3, save=True, save_h1=True, save_gate=True)
        torch.cuda.synchronize()
        scale, h2s = out64.abs().max().item(), h2.abs().max().item()
        e_ref = (ref[0].double() - out64).abs().max().item() / scale
        errs = [(o.double() - out64).abs().max().item() / scale for o in (got_inf, got_gate[0], got[0])]
        h2_err = (got[2].double() - h2).abs().max().item() / h2s
        h2_ref = (ref[2].double() - h2).abs().max().item() / h2s
        bits = ((got[3].view(torch.int32)[:, :, None] >> torch.arange(32, device=dev)) & 1).reshape(m, 256).bool()
        gate_ok = bool((bits == (got[2] > 0)).all()) and bool((got_gate[3] == got[3]).all())
        h1_ok = bool((got[1] == ref[1]).all())
        same = bool((got_inf == got[0]).all()) and bool((got_gate[0] == got[0]).all())
        ok = gate_ok and h1_ok and same and h2_err <= max(3 * h2_ref, 1e-6) and max(errs) <= max(3 * e_ref, 2e-6)
        bad += not ok
        print(f"d_in={d_in} n_out={n_out} m={m:7d} out err64 {max(errs):.2e} (bf16 planes {e_ref:.2e}) h2 {h2_err:.2e} ({h2_ref:.2e})"
              f" gate={gate_ok} h1={h1_ok} modes_agree={same} {'ok' if ok else 'FAIL'}", flush=True)
print("FAILURES", bad)
sys.exit(1 if bad else 0)
