import sys, json, io, contextlib
sys.path.insert(0, "/root/repo")
sys.argv = ["bench.py", "--recurrent", "--num-envs", "8192", "--horizon", "256", "--steps", "6", "--warmup", "2", "--no-cpu-baseline", "--uninstrumented-steps", "6"]
import bench
from rl8_amd.nn import fused_lstm
for tiled in (True, False, True, False):
    fused_lstm.TILED_SAVED = tiled
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        bench.run(bench.parse_args())
    d = json.loads(buf.getvalue().strip().splitlines()[-1])
    k = d["kernels"]
    print("tiled", tiled, round(d["value"] / 1e6, 2), "M/s", round(d["ms_per_step"], 2), "ms; uninstr", round(d["value_uninstrumented"] / 1e6, 2),
          {n: round(k[n]["avg_ms"], 3) for n in ("lstm_rows_backward", "lstm_step_save", "lstm_wgrad")}, flush=True)
