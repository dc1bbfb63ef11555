"""Kernel-tuning aid: where a k-step of the bf16-plane forward kernel spends its cycles.

    make -s -C rl8_amd/csrc BUILD=$PWD/build_diag/objtr OUT=$PWD/build_diag/librl8_amd_splittrace.so FLAGS_EXTRA=-DRL8_SPLIT_TRACE
    RL8_AMD_LIBRARY=build_diag/librl8_amd_splittrace.so python tools/split_phase_trace.py

Stamps (s_memtime, shader cycles) per wave and step of tile iteration 3: 0 top of the
step (behind the previous barrier), 1 first MFMA (operands have arrived, the next
chunk's weights are in), 2 last MFMA issued, 3 behind the step barrier.
"""
import sys, os, ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from rl8_amd import hip
dev = "cuda"
m = 1 << 20
g = torch.Generator(device=dev).manual_seed(1)
x = torch.randn(m, 1, device=dev, generator=g) * 30
w1 = torch.randn(256, 1, device=dev, generator=g); b1 = torch.randn(256, device=dev, generator=g)
w2 = torch.randn(256, 256, device=dev, generator=g) / 16; b2 = torch.randn(256, device=dev, generator=g)
w3 = torch.randn(2, 256, device=dev, generator=g) / 16; b3 = torch.randn(2, device=dev, generator=g)
ws = hip.mlp_pack_w2_split(w2)
for save in (False, True):
    for _ in range(5):
        hip.mlp_tower_forward_split(x, w1, b1, ws, b2, w3, b3, save=save, save_h1=False, save_gate=save)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * (512 * 4 * 16 * 4))()
    lib = hip.load(); lib.rl8_debug_split_trace.argtypes = [C.c_void_p]
    assert lib.rl8_debug_split_trace(buf) == 0
    t = np.frombuffer(buf, dtype=np.uint64).reshape(512, 4, 16, 4).astype(np.int64)
    t0, t1, t2, t3 = t[..., 0], t[..., 1], t[..., 2], t[..., 3]
    # s_memtime ticks at 100 MHz? report in ticks and relative shares
    lds_wait = (t1 - t0); issue = (t2 - t1); barrier = (t3 - t2); step = (t3 - t0)
    tile = t[:, :, 15, 3] - t[:, :, 0, 0]
    print("save" if save else "infer", "ticks/step median", np.median(step), "| top->MFMA start", np.median(lds_wait), "| MFMA issue phase", np.median(issue),
          "| barrier", np.median(barrier), "| 16-step span", np.median(tile))
    print("   per-step medians (step 0..15): wait", [int(np.median(lds_wait[:, :, s])) for s in range(16)])
    print("   barrier:", [int(np.median(barrier[:, :, s])) for s in range(16)])
    print("   issue:", [int(np.median(issue[:, :, s])) for s in range(16)])
    ebuf = (C.c_ulonglong * (512 * 4 * 4))()
    lib.rl8_debug_split_trace_epilogue.argtypes = [C.c_void_p]
    assert lib.rl8_debug_split_trace_epilogue(ebuf) == 0
    e = np.frombuffer(ebuf, dtype=np.uint64).reshape(512, 4, 4).astype(np.int64)
    end15 = t[:, :, 15, 3]
    print("   epilogue (cycles): barrier(15) -> start", int(np.median(e[..., 0] - end15)), "| bias/ReLU/h2 stores/gate + head of the first row tile",
          int(np.median(e[..., 1] - e[..., 0])), "| second row tile's head", int(np.median(e[..., 2] - e[..., 1])), "| barrier", int(np.median(e[..., 3] - e[..., 2])))
