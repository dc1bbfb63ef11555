"""Time of rl8_lstm_rows_backward_f32 alone on the recurrent bench's shape (2^19 sequences x 4 steps), for the
diagnostic builds of tools/diag_mlp.sh lr<bits> (RL8_AMD_LIBRARY=build_diag/librl8_amd_lr<bits>.so)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from rl8_amd import hip  # noqa: E402
sys.path.insert(0, str(Path(__file__).resolve().parent))
from lstm_rows_check import inputs  # noqa: E402

import os  # noqa: E402

b, l = 1 << 19, int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda:0")
c0, gates, cs, dhs, w_hh = inputs(b, l, dev, 7)
packed = hip.lstm_rows_backward_pack(w_hh)
stamps = torch.zeros(16, dtype=torch.int64, device=dev)
os.environ["RL8_LR_STAMP_PTR"] = hex(stamps.data_ptr())  # read by builds with -DRL8_LR_STAMP (tools/diag_mlp.sh lrstamp)
hip.lstm_rows_backward(c0, gates, cs, dhs, packed)
torch.cuda.synchronize()
a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(5):
    hip.lstm_rows_backward(c0, gates, cs, dhs, packed)
e.record()
torch.cuda.synchronize()
print(f"{a.elapsed_time(e) / 5:.2f} ms per {b} x {l} row-steps")
if int(stamps[11]):
    v = stamps.double() / float(stamps[11])
    names = [f"vmcnt before barrier {k}" for k in range(8)] + ["vmcnt before phase B's read", "vmcnt before phase C's read", "barriers", "all", "fragment waits (lgkmcnt)",
             "issue of row loads / parks / requests", "gate arithmetic incl. its waits and stores", "-"]
    for nm, x in zip(names, v.tolist()):
        print(f"  {nm:32s} {100 * x:5.1f} % of the waves' time")
