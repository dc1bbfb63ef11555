"""In-kernel clock and matrix-pipe occupancy of the rows-per-wave forward (mlp_rows_kernels.hip) (diagnostic build with -DRL8_ROWS_STAMP:
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
stamps = torch.zeros(512 * 4 * 4, dtype=torch.int64, device=dev)  # [workgroup][wave][cycles, real time, tiles, -]
os.environ["RL8_ROWS_STAMP_PTR"] = str(stamps.data_ptr())
CASES = [(1, {}, "inference", 0), (1, dict(save=True, save_gate=True, save_h2=False), "gate bits only", 0),
         (1, dict(save=True, save_h1=False, save_gate=True), "with h2", 0)]
# cut-down variants of the inference kernel (DIAG bits: 1 production, 2 epilogue, 4 matrix work)
for diag, what in ((1, "no production"), (2, "no epilogue"), (3, "no production, no epilogue"), (4, "no matrix work"),
                   (7, "ring + fragment reads only")):
    CASES.append((1, {}, what, diag))
for mode, kw, label, diag in CASES:
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
    # per wave: tiles x 16 half-steps x 8 column tiles x 2 row tiles x 3 products x 16 cycles of the SIMD's matrix
    # pipe, which serves two such waves
    per_block = 1 if diag & 4 else 3
    busy = [2 * t * 16 * 8 * 2 * per_block * 16 / float(c) for c, _, t, _ in st.tolist()]
    cyc_tile = [float(c) / t for c, _, t, _ in st.tolist()]
    print(f"{label:44s} {us:7.1f} us  clock median {statistics.median(clk):6.0f} MHz (min {min(clk):.0f} max {max(clk):.0f})"
          f"  wave cycles per tile {statistics.median(cyc_tile):8.0f}  matrix pipe busy {statistics.median(busy):.3f}", flush=True)
