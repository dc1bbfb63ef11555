"""In-kernel clock and matrix-pipe occupancy of the rows-per-wave forward (diagnostic build with -DRL8_ROWS_STAMP:
`tools/diag_mlp.sh stamp`).  After >= 2 s of back-to-back launches on random data, each wave's
delta s_memtime / delta s_memrealtime x 100 MHz is the clock the chip held inside the kernel
(MI355X_MICROARCH.md, DVFS give-back item 6); MFMA issue cycles / wave cycles is how busy the SIMD's matrix pipe was.

    RL8_AMD_LIBRARY=build_diag/librl8_amd_stamp.so python tools/diag/rows_clock.py
"""
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from rl8_amd import hip

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
N = 1 << 20
x = torch.randn(N, 1, device=dev, generator=g) * 30
w1 = torch.randn(256, 1, device=dev, generator=g)
b1 = torch.randn(256, device=dev, generator=g)
w2 = torch.randn(256, 256, device=dev, generator=g) / 16
b2 = torch.randn(256, device=dev, generator=g)
w3 = torch.randn(2, 256, device=dev, generator=g) / 16
b3 = torch.randn(2, device=dev, generator=g)
w2h = hip.mlp_pack_w2_f16(w2)
stamps = torch.zeros(512 * 4 * 4, dtype=torch.int64, device=dev)
os.environ["RL8_ROWS_STAMP_PTR"] = str(stamps.data_ptr())
CASES = [(1, {}, "rows1 inference", 0), (2, {}, "rows2 inference", 0),
         (1, dict(save=True, save_gate=True, save_h2=False), "rows1 gate-only", 0),
         (1, dict(save=True, save_h1=False, save_gate=True), "rows1 with h2", 0)]
for mode in (1, 2):  # cut-down variants of the inference kernel (DIAG bits: 1 production, 2 epilogue, 4 matrix work, 8 LDS reads)
    for diag, what in ((1, "no production"), (2, "no epilogue"), (3, "no production, no epilogue"), (4, "no matrix work"),
                       (8, "LDS reads once per step"), (7, "nothing but LDS reads + ring"), (11, "matrix work only (reads once)"),
                       (15, "ring + barriers only")):
        CASES.append((mode, {}, f"rows{mode} {what}", diag))
for mode, kw, label, diag in CASES:
    os.environ["RL8_MLP_FWD_ROWS"] = str(mode)
    os.environ["RL8_ROWS_DIAG"] = str(diag)
    t_end = time.time() + 2.1
    while time.time() < t_end:
        for _ in range(50):
            hip.mlp_tower_forward_split(x, w1, b1, w2h, b2, w3, b3, **kw)
        torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        hip.mlp_tower_forward_split(x, w1, b1, w2h, b2, w3, b3, **kw)
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) / 20 * 1e3
    st = stamps.cpu().view(-1, 4)
    st = st[st[:, 1] > 0]
    clk = [float(c) / float(r) * 100 for c, r, _, _ in st.tolist()]
    mt = mode
    waves_per_simd = 2 if mode == 1 else 1
    # per wave: tiles x 16 steps x 24 MT products x 32 cycles of pipe; the SIMD's pipe serves waves_per_simd such waves
    per_block = 1 if diag & 4 else 3
    busy = [waves_per_simd * t * 16 * 8 * per_block * mt * 32 / float(c) for c, _, t, _ in st.tolist()]
    cyc_tile = [float(c) / t for c, _, t, _ in st.tolist()]
    print(f"{label:44s} {us:7.1f} us  clock median {statistics.median(clk):6.0f} MHz (min {min(clk):.0f} max {max(clk):.0f})"
          f"  wave cycles per tile {statistics.median(cyc_tile):8.0f}  matrix pipe busy {statistics.median(busy):.3f}", flush=True)
os.environ["RL8_MLP_FWD_ROWS"] = "0"
