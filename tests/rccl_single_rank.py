"""Single-rank RCCL smoke: drive every collective EnvShards issues through the real
nccl backend (world size 1, collectives forced on)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 400))
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
from rl8_amd import AlgorithmConfig
from rl8_amd.env import DiscreteDummyEnv, ContinuousDummyEnv
from rl8_amd import parallel
parallel.EnvShards.active = property(lambda self: True)   # force the collectives with one rank
torch.manual_seed(0)
for env_cls, kw in ((DiscreteDummyEnv, {}), (ContinuousDummyEnv, {"sgd_minibatch_size": 4096})):
    algo = AlgorithmConfig(num_envs=1024, horizon=16, **kw).build(env_cls)
    assert algo.shards.active and dist.get_backend() == "nccl"
    for _ in range(2):
        c = algo.collect(); s = algo.step()
    print(env_cls.__name__, "returns/mean", c["returns/mean"], "loss", s["losses/total"])
# Latency of the step's collectives through RCCL with this one rank (the launch + completion path of the production
# backend; the xGMI hops of N > 1 ranks come on top): DESIGN.md section 7 prices the 8-GPU projection with these.
import json, time
shards = algo.shards
grads = [p for p in algo.policy.model.parameters()]
for p in grads:
    p.grad = torch.zeros_like(p)
sums = [torch.zeros(5, dtype=torch.float64, device="cuda")]
moments = torch.zeros(3, dtype=torch.float64, device="cuda")
raw = torch.zeros(12, dtype=torch.float64, device="cuda")
lat = {}
for name, call in (("sum_gradients_ (flat fp64 [gradient | loss sums], %d floats)" % sum(p.numel() for p in grads),
                    lambda: shards.sum_gradients_(grads, sums)),
                   ("sum_ (fp64 moments)", lambda: shards.sum_(moments)),
                   ("combine_rollout_stats (all-gather of 12 fp64 + host read)", lambda: shards.combine_rollout_stats(raw))):
    for _ in range(5):
        call()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        call()
    torch.cuda.synchronize()
    lat[name] = round((time.perf_counter() - t0) / 50 * 1e6, 1)
print("rccl single-rank latency us:", json.dumps(lat))
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
if os.path.isdir(out):
    with open(os.path.join(out, "rccl_single_rank_latency.json"), "w") as f:
        json.dump({"unit": "us per call, one rank, backend nccl (RCCL)", "latency": lat}, f, indent=1)
dist.barrier(); torch.cuda.synchronize()
dist.destroy_process_group()
print("rccl single-rank collectives ok")
