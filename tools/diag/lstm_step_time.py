"""Diagnostic (GPU): time of the bf16-plane LSTM step kernel at config 5's training size."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rl8_amd import hip
DEV = "cuda:0"; b, l, d = 1 << 19, 4, 1
g = torch.Generator(device=DEV).manual_seed(0)
lstm = torch.nn.LSTM(d, 256, batch_first=True).to(DEV)
x = torch.randn(b, l, d, device=DEV, generator=g); h0 = torch.randn(b, 256, device=DEV, generator=g) * .5; c0 = torch.randn(b, 256, device=DEV, generator=g)
packed, wb = hip.lstm_pack_split(lstm.weight_ih_l0, lstm.weight_hh_l0, lstm.bias_ih_l0, lstm.bias_hh_l0)
planes = hip.lstm_state_planes(b, DEV, copies=2)
for save in (False, True):
    for _ in range(2): hip.lstm_forward_split(x, h0, c0, packed, wb, save=save, planes=planes)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): hip.lstm_forward_split(x, h0, c0, packed, wb, save=save, planes=planes)
    e1.record(); torch.cuda.synchronize()
    print("save" if save else "infer", "ms per pass (4 steps incl. state planes):", e0.elapsed_time(e1) / 3)
