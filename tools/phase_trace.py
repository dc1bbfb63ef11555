"""Kernel-tuning aid: per-phase shader-clock timings of the tower kernels.

Needs the tracing build of the library (phase stamps are compiled out of the
shipped one)::

    tools/diag_mlp.sh trace
    RL8_AMD_LIBRARY=build_diag/librl8_amd_trace.so python tools/phase_trace.py

Prints, per kernel, the mean / p10 / p90 cycles every wave spent between
consecutive phase boundaries over tile iterations 4..7 of all workgroups, and
the tile period. This is where DESIGN.md's statements about VALU phases, barrier
cost and the time both resident workgroups spend outside their matrix loops come
from.
"""
import sys, os, ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from rl8_amd import hip
dev = "cuda:0"
N = 1 << 20
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(N, 1, device=dev, generator=g) * 30
w1 = torch.randn(256, 1, device=dev, generator=g); b1 = torch.randn(256, device=dev, generator=g)
w2 = torch.randn(256, 256, device=dev, generator=g) / 16; b2 = torch.randn(256, device=dev, generator=g)
w3 = torch.randn(2, 256, device=dev, generator=g) / 16; b3 = torch.randn(2, device=dev, generator=g)
w2p = hip.mlp_pack_w2(w2); w2tp = hip.mlp_pack_w2(w2, transposed=True)
dout = torch.randn(N, 2, device=dev, generator=g) / N
lib = hip.load()
def dump(label, names):
    torch.cuda.synchronize()
    buf = np.zeros(512 * 4 * 4 * 12, np.uint64)
    lib.rl8_debug_phase_trace(buf.ctypes.data_as(C.c_void_p))
    t = buf.reshape(512, 4, 4, 12).astype(np.int64)
    t = t[:, :, :, :len(names) + 1]
    d = np.diff(t, axis=-1)          # [wg, wave, tile, phase]
    total = (t[..., -1] - t[..., 0])
    print(f"== {label}: cycles per tile and workgroup: mean {total.mean():.0f} (min {total.min()}, max {total.max()})")
    for i, nm in enumerate(names):
        print(f"   {nm:34s} mean {d[..., i].mean():8.0f}  p10 {np.percentile(d[..., i], 10):8.0f}  p90 {np.percentile(d[..., i], 90):8.0f}")
    # tile period: stamp 0 of consecutive tiles
    period = np.diff(t[..., 0], axis=2)
    print(f"   tile period (top to top) mean {period.mean():.0f}")
for _ in range(3):
    out, h1, h2 = hip.mlp_tower_forward(x, w1, b1, w2p, b2, w3, b3, save=True)
dump("forward save", ["top->barrier(x staged)", "layer1 (+B prefetch, h1 stores)", "barrier", "matrix loop", "barrier", "epilogue (+h2 stores)", "barrier", "head"])
for _ in range(3):
    hip.mlp_tower_forward(x, w1, b1, w2p, b2, w3, b3)
dump("forward inference", ["top->barrier(x staged)", "layer1 (+B prefetch)", "barrier", "matrix loop", "barrier", "epilogue", "barrier", "head"])
for _ in range(3):
    hip.mlp_tower_backward(x, h1, h2, dout, w2tp, w3)
dump("backward", ["stage x/dout + wait h2 DMA", "barrier", "phase1 (+B,h1 issue, dz2 stores)", "barrier", "matrix loop", "barrier", "issue next h2/h1", "phase3"])
