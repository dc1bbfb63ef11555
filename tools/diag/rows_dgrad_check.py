"""Gate-mode data gradient: the rows-per-wave kernel (mlp_rows_kernels.hip) against the previous kernel (same ABI entry,
RL8_MLP_DGRAD_ROWS_OFF=1) and fp64 -- dW1 / db1 of rank-one heads on ragged sizes.

    python tools/diag/rows_dgrad_check.py
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from rl8_amd import hip

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(7)
lib = hip.load()
p = hip._ptr
bad = 0
for d_in, n_out in ((1, 1), (1, 2), (2, 1), (2, 2), (3, 1), (3, 2)):
    w1 = torch.randn(256, d_in, device=dev, generator=g) * 0.5
    b1 = torch.randn(256, device=dev, generator=g) * 0.1
    w2 = torch.randn(256, 256, device=dev, generator=g) / 16
    b2 = torch.randn(256, device=dev, generator=g) * 0.1
    w3 = torch.randn(n_out, 256, device=dev, generator=g) / 16
    b3 = torch.randn(n_out, device=dev, generator=g)
    w2h = hip.mlp_pack_w2_f16(w2)
    gate_pack = hip.mlp_pack_w2_f16_gate(w2, w3)
    width = int(lib.rl8_mlp_backward_partial_floats(d_in, n_out))
    for m in (1, 31, 100, 128, 129, 257, 1000, 4096 + 5, (1 << 17) + 77, (1 << 20) + 3):
        x = torch.randn(m, d_in, device=dev, generator=g) * 3
        dout = torch.randn(m, n_out, device=dev, generator=g) / m
        if n_out == 2:
            dout[:, 1] = -dout[:, 0]
        gate = hip.mlp_tower_forward_split(x, w1, b1, w2h, b2, w3, b3, save=True, save_gate=True, save_h2=False)[3]
        res = {}
        for off in ("1", "0"):
            os.environ["RL8_MLP_DGRAD_ROWS_OFF"] = off
            partials = torch.full((int(lib.rl8_mlp_backward_max_rows()), width), float("nan"), device=dev)
            rows = C.c_int(0)
            st = lib.rl8_mlp_tower_backward_gate_f16_f32(p(x), p(w1), p(b1), p(dout), m, d_in, p(gate_pack), n_out, p(partials),
                                                         C.byref(rows), p(gate), hip._stream())
            assert st == 0, st
            own = min((m + 127) // 128, 512)  # the data-gradient kernel's own rows (the rest belong to the weight-gradient kernel's grid)
            tot = partials[:own, : 256 * d_in + 256].double().sum(0)
            res[off] = (tot[: 256 * d_in].view(256, d_in), tot[256 * d_in:], rows.value, partials[:own])
        os.environ["RL8_MLP_DGRAD_ROWS_OFF"] = "0"
        # fp64 from the kernel's own gate bits
        bits = ((gate.view(torch.int32)[:, :, None] >> torch.arange(32, device=dev)) & 1).reshape(m, 256).double()
        w3e = (w3[0] - w3[1] if n_out == 2 else w3[0]).double()
        dz2 = bits * dout[:, :1].double() * w3e
        dh1 = dz2 @ w2.double()
        pre = x.double() @ w1.double().T + b1.double()
        dz1 = dh1 * (pre.float() > 0)  # (the kernels recompute the fp32 gate of h1)
        want_w, want_b = dz1.T @ x.double(), dz1.sum(0)
        sw, sb = want_w.abs().max().item() + 1e-30, want_b.abs().max().item() + 1e-30
        e_old = max((res["1"][0] - want_w).abs().max().item() / sw, (res["1"][1] - want_b).abs().max().item() / sb)
        e_new = max((res["0"][0] - want_w).abs().max().item() / sw, (res["0"][1] - want_b).abs().max().item() / sb)
        finite = bool(torch.isfinite(res["0"][3][:, : 256 * d_in + 256]).all())
        ok = finite and res["0"][2] == res["1"][2] and e_new <= max(3 * e_old, 2e-6)
        bad += not ok
        print(f"d_in={d_in} n_out={n_out} m={m:8d} rows={res['0'][2]:3d} err64 new {e_new:.2e} old {e_old:.2e} {'ok' if ok else 'FAIL'}", flush=True)
print("FAILURES", bad)
sys.exit(1 if bad else 0)
