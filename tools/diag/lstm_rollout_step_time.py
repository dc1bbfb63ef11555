"""Time of one rollout timestep's LSTM launch (rl8_lstm_step_split_f32 without saved gates) at the recurrent bench's 8 192
environments per GPU, for the diagnostic builds of tools/diag_mlp.sh ls<bits> (RL8_AMD_LIBRARY=...)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from rl8_amd import hip  # noqa: E402

b, d = int(sys.argv[1]) if len(sys.argv) > 1 else 8192, 1
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(3)
lstm = torch.nn.LSTM(d, 256, batch_first=True).to(dev)
x = torch.randn(b, 1, d, device=dev, generator=g) * 10
h0 = torch.rand(b, 256, device=dev, generator=g) - 0.5
c0 = torch.randn(b, 256, device=dev, generator=g)
packed, wb = hip.lstm_pack_split(lstm.weight_ih_l0, lstm.weight_hh_l0, lstm.bias_ih_l0, lstm.bias_hh_l0)
planes = hip.lstm_state_planes(b, dev, copies=2)
half = planes.numel() // 2
hs, cs = torch.empty(b, 256, device=dev), torch.empty(b, 256, device=dev)
lib = hip.load()
hip.lstm_split_state(h0, out=planes)


def step(t):
    p_in, p_out = hip._ptr(planes) + (t & 1) * half, hip._ptr(planes) + ((t + 1) & 1) * half
    hip._check(lib.rl8_lstm_step_split_f32(hip._ptr(x), d, d, p_in, hip._ptr(c0), 256, hip._ptr(packed), hip._ptr(wb), b,
                                           hip._ptr(hs), 256, hip._ptr(cs), 256, None, 0, p_out, hip._stream()), "step")


for t in range(10):
    step(t)
torch.cuda.synchronize()
a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for t in range(400):
    step(t)
e.record()
torch.cuda.synchronize()
print(f"{a.elapsed_time(e) / 400 * 1e3:.2f} us per launch of {b} rows")
