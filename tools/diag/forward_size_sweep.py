"""Forward tower kernel time against launch size (rows), per (d_in, n_out, save mode): what a rollout timestep's launch
costs beside the same rows inside a large training launch.  Usage: python tools/diag/forward_size_sweep.py"""
import sys
import numpy as np
import torch

sys.path.insert(0, "/root/repo")
from rl8_amd import hip

dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)


BACK_TO_BACK = 16  # launches between the two events, behind a long kernel that lets the host run ahead


def med(fn, reps=8):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    blocker = torch.empty(1 << 28, device=dev)
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        blocker.zero_()  # ~0.3 ms of GPU work: the launches below are queued before it ends
        a.record()
        for _ in range(BACK_TO_BACK):
            fn()
        b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) * 1e3 / BACK_TO_BACK)
    return float(np.median(ts))


for d_in, n_out, mode in ((1, 2, "gate"), (1, 1, "gate"), (5, 3, "h2"), (5, 1, "gate"), (5, 3, "none"), (1, 2, "none")):
    w1 = torch.randn(256, d_in, device=dev, generator=g) * 0.5
    b1 = torch.randn(256, device=dev, generator=g) * 0.1
    w2 = torch.randn(256, 256, device=dev, generator=g) / 16
    b2 = torch.randn(256, device=dev, generator=g) * 0.1
    w3 = torch.randn(n_out, 256, device=dev, generator=g) / 16
    b3 = torch.zeros(n_out, device=dev)
    pack = hip.mlp_pack_w2_f16(w2)
    line = []
    for logm in (17, 18, 19, 20, 22, 24):
        m = 1 << logm
        x = torch.randn(m, d_in, device=dev, generator=g)
        out = torch.empty(m, n_out, device=dev)
        h2 = torch.empty(m, 256, device=dev) if mode == "h2" else None
        gate = torch.empty(m, 8, dtype=torch.int32, device=dev) if mode != "none" else None
        if mode == "none":
            fn = lambda: hip.mlp_tower_forward_split(x, w1, b1, pack, b2, w3, b3, out=out)
        elif mode == "gate":
            fn = lambda: hip.mlp_tower_forward_split(x, w1, b1, pack, b2, w3, b3, save=True, save_gate=True, save_h2=False, out=out, gate_out=gate)
        else:
            fn = lambda: hip.mlp_tower_forward_split(x, w1, b1, pack, b2, w3, b3, save=True, save_h1=False, save_gate=True, out=out, h2_out=h2, gate_out=gate)
        t = med(fn)
        line.append(f"2^{logm}: {t:8.1f} us ({t * 1e3 / m:.3f} ns/row)")
        del x, out, h2, gate
    print(f"d_in={d_in} n_out={n_out} save={mode:5s} " + "  ".join(line), flush=True)
