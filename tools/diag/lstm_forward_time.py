"""Time of the training forward of the fused LSTM alone (rl8_lstm_split_state + rl8_lstm_step_split_f32 per step, gates
saved) on the recurrent bench's shape: 2^19 sequences x 4 steps, d_in = 1.  Under rocprofv3 --pmc it gives the HBM
traffic of lstm_step_split_kernel<1, true> per launch (algorithmic: 2 KiB read, 7 KiB written per row)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from rl8_amd import hip  # noqa: E402

b, l, d = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 19, int(sys.argv[1]) if len(sys.argv) > 1 else 4, 1
save = (sys.argv[3] if len(sys.argv) > 3 else "save") == "save"  # "rollout": no gates stored (the rollout's launches)
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(3)
lstm = torch.nn.LSTM(d, 256, batch_first=True).to(dev)
x = torch.randn(b, l, d, device=dev, generator=g) * 10
h0 = torch.rand(b, 256, device=dev, generator=g) - 0.5
c0 = torch.randn(b, 256, device=dev, generator=g)
packed, wb = hip.lstm_pack_split(lstm.weight_ih_l0, lstm.weight_hh_l0, lstm.bias_ih_l0, lstm.bias_hh_l0)
planes = hip.lstm_state_planes(b, dev, copies=2)


def run():
    return hip.lstm_forward_split(x, h0, c0, packed, wb, save=save, planes=planes)


run()
torch.cuda.synchronize()
a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
reps = 5 if b * l >= 1 << 20 else 50
for _ in range(reps):
    out = run()
    del out
e.record()
torch.cuda.synchronize()
print(f"{a.elapsed_time(e) / reps:.3f} ms per forward of {b} x {l} row-steps ({a.elapsed_time(e) / reps / l * 1e3:.1f} us per step launch incl. allocation), save={save}")
