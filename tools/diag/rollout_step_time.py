import sys, torch, numpy as np
sys.path.insert(0, "/root/repo")
from rl8_amd import hip
n = 1 << 20
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
logits = torch.randn(n, 1, 2, device=dev, generator=g) * 0.5
value = torch.randn(n, 1, device=dev, generator=g)
noise = -torch.log(torch.rand(n, 1, 2, device=dev, generator=g).clamp_min(1e-9))
state = torch.rand(n, 1, device=dev, generator=g) * 200 - 100
cols = {k: torch.empty(n, 1, device=dev) for k in ("logp", "value", "reward", "obs", "rdr1")}
act = torch.empty(n, 1, dtype=torch.int64, device=dev)
rdr0 = torch.randn(n, 1, device=dev, generator=g)
def run(noise_t, det):
    hip.rollout_step_dummy(discrete=True, squashed=False, features=logits, features2=None, value=value, noise=noise_t, state=state,
        action_col=act, logp_col=cols["logp"], value_col=cols["value"], reward_col=cols["reward"], obs_col_next=cols["obs"],
        rdr_t=rdr0, rdr_t1=cols["rdr1"], gamma=0.95, seed=1, step=2, env_offset=0, deterministic=det)
for name, nz, det in (("philox", None, False), ("injected noise", noise, False), ("deterministic (exact path)", None, True)):
    for _ in range(5): run(nz, det)
    torch.cuda.synchronize()
    ts = []
    for _ in range(30):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); run(nz, det); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) * 1e3)
    print(f"{name:30s} median {np.median(ts):.1f} us  min {min(ts):.1f} us")
# a plain copy kernel of the same bytes for scale
x = torch.empty(54_500_000 // 8, device=dev); y = torch.empty_like(x)
for _ in range(5): y.copy_(x)
torch.cuda.synchronize(); ts = []
for _ in range(30):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); y.copy_(x); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) * 1e3)
print(f"torch copy of 27 MB -> 27 MB       median {np.median(ts):.1f} us  min {min(ts):.1f} us")
