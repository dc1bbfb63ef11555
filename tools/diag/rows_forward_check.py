"""Rows-per-wave forward (mlp_rows_kernels.hip) against the previous fp16-plane kernel and fp64:
gate bits and h2 must be bit-identical (same products, same accumulation order), the head within rounding.

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
for d_in, n_out in ((1, 2), (1, 1), (1, 3), (2, 2), (3, 1), (5, 3)):
    w1 = torch.randn(256, d_in, device=dev, generator=g)
    b1 = torch.randn(256, device=dev, generator=g)
    w2 = torch.randn(256, 256, device=dev, generator=g) / 16
    b2 = torch.randn(256, device=dev, generator=g)
    w3 = torch.randn(n_out, 256, device=dev, generator=g) / 16
    b3 = torch.randn(n_out, device=dev, generator=g)
    w2h = hip.mlp_pack_w2_f16(w2)
    os.environ["RL8_MLP_PACK_LAYOUT"] = "1"
    w2h16 = hip.mlp_pack_w2_f16(w2)
    os.environ["RL8_MLP_PACK_LAYOUT"] = "0"
    for m in (1, 31, 100, 128, 129, 257, 1000, 4096 + 5, (1 << 17) + 77):
        x = torch.randn(m, d_in, device=dev, generator=g) * 30
        x[::7] *= 1e-3
        os.environ["RL8_MLP_FWD_ROWS"] = "0"
        ref = hip.mlp_tower_forward_split(x, w1, b1, w2h, b2, w3, b3, save=True, save_h1=False, save_gate=True)
        ref_inf = hip.mlp_tower_forward_split(x, w1, b1, w2h, b2, w3, b3)[0]
        h1 = torch.relu(x.double() @ w1.double().T + b1.double())
        h2 = torch.relu(h1 @ w2.double().T + b2.double())
        out64 = h2 @ w3.double().T + b3.double()
        for mode in (1, 2, 16):
            if mode == 2 and not (d_in == 1 and n_out <= 2):
                continue
            if mode == 16 and d_in != 1:
                continue
            os.environ["RL8_MLP_FWD_ROWS"] = str(mode)
            pk = w2h16 if mode == 16 else w2h
            got_inf = hip.mlp_tower_forward_split(x, w1, b1, pk, b2, w3, b3)[0]
            got_gate = hip.mlp_tower_forward_split(x, w1, b1, pk, b2, w3, b3, save=True, save_h1=False, save_gate=True,
                                                   save_h2=False)
            got = hip.mlp_tower_forward_split(x, w1, b1, pk, b2, w3, b3, save=True, save_h1=False, save_gate=True)
            os.environ["RL8_MLP_FWD_ROWS"] = "0"
            torch.cuda.synchronize()
            scale = out64.abs().max().item()
            e_ref = (ref[0].double() - out64).abs().max().item() / scale
            errs = [(o.double() - out64).abs().max().item() / scale for o in (got_inf, got_gate[0], got[0])]
            same_inf = (got_inf - ref_inf).abs().max().item() / scale
            if mode == 16:  # another association of the k sum: compare h2 with fp64, the gate with h2 > 0 of the kernel's own h2
                h2_err = (got[2].double() - h2).abs().max().item() / h2.abs().max().item()
                h2_ref = (ref[2].double() - h2).abs().max().item() / h2.abs().max().item()
                h2_eq = h2_err <= max(3 * h2_ref, 1e-6)
                bits = ((got[3].view(torch.int32)[:, :, None] >> torch.arange(32, device=dev)) & 1).reshape(m, 256).bool()
                gate_eq = bool((bits == (got[2] > 0)).all()) and bool((got_gate[3] == got[3]).all())
            else:
                gate_eq = bool((got[3] == ref[3]).all()) and bool((got_gate[3] == ref[3]).all())
                h2_eq = bool((got[2] == ref[2]).all())
            ok = gate_eq and h2_eq and max(errs) <= max(3 * e_ref, 2e-6) and same_inf < 2e-6
            bad += not ok
            print(f"d_in={d_in} n_out={n_out} m={m:7d} mode={mode} err64 ref {e_ref:.2e} rows {max(errs):.2e} "
                  f"vs-f16 {same_inf:.2e} gate_eq={gate_eq} h2_eq={h2_eq} {'ok' if ok else 'FAIL'}", flush=True)
print("FAILURES", bad)
sys.exit(1 if bad else 0)
