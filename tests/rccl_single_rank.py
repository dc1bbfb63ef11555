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
dist.barrier(); torch.cuda.synchronize()
dist.destroy_process_group()
print("rccl single-rank collectives ok")
