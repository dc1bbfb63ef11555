"""Headline benchmark: env transitions/sec (+ policy updates/sec) of
``collect(); step()`` on DiscreteDummyEnv, num_envs = 2^20 per GPU, horizon 32,
all ``AlgorithmConfig`` defaults (BASELINE.json configs[1]).

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \\
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = one ``collect()`` (32 timesteps of policy forward + fused sample /
env.step / bookkeeping launch, bootstrap value, stats) + one ``step()`` (GAE,
4 SGD iterations of policy forward, fused PPO loss fwd+bwd, policy backward,
clip, Adam). Everything lives in HBM before the timed region starts.

Weak scaling: every rank owns 2^20 environments (env-sharded, RCCL all-reduce of
moments / loss sums / gradients only).

Besides the contract's fields, the JSON line carries
  roofline      the dominant hand-written kernel (fused PPO loss fwd+bwd, 44 B per
                sample algorithmic) timed with HIP events inside the timed region;
  kernels       the same for every hand kernel that ran;
  cpu_baseline  the CPU restatement (oracle/, kind "port") of the same algorithm
                on the host cores, on a bounded sample (rank 0, N=1 only).
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); measured copy ceiling ~6290
HBM_COPY_CEILING_GBS = 6290.0
MFMA_F32_PEAK_TFLOPS = 157.3  # dense fp32 MFMA (= fp32 vector) peak, MI355X_MICROARCH.md
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA peak (spec), MI355X_MICROARCH.md; 1750 sustained on changing operands
SPLIT_PRODUCTS = 6              # bf16 plane products per fp32 product in the bf16-plane kernels

# MFMA-bound kernels: algorithmic FLOP per unit (row) for the default towers,
# 2 * (256*d_in + 256*256 + 256*n_out); the backward kernel does the data-gradient
# GEMM only (the weight-gradient GEMM is a library call).
def tower_flops_per_row(d_in: int, n_out: int) -> float:
    return 2.0 * (256 * d_in + 256 * 256 + 256 * n_out)


# Algorithmic bytes per unit (DESIGN.md "Kernels").
ALGORITHMIC_BYTES = {
    "ppo_loss_categorical": 44.0,   # logits 8 + value 4 + action 8 + logp 4 + adv 4 + ret 4; grads 8 + 4
    "ppo_loss_normal": 40.0,
    "gae_scan": 16.0 + 8.0 / 32.0,  # r 4 + v 4 + adv 4 + ret 4 per transition, + 8 B/env for column H
    "advantage_normalise": 8.0,
    "rollout_step_dummy": 52.0,     # logits 8 value 4 state 4 rdr 4 | action 8 logp 4 value 4 reward 4 obs 4 state 4 rdr 4
    "rollout_stats": 8.0,
    "gather_minibatch": 56.0,
}


# HBM traffic from the PMC counters (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE
# passes over tools/kernel_microbench.py at config-2 shapes; corrected as
# MI355X_MICROARCH.md prescribes; summary committed under profiles/). Scaled by
# units to the launch size bench.py uses.
PMC_SUMMARY = os.path.join(ROOT, "profiles", "r01_pmc_traffic_microbench.json")
PMC_KERNEL = {  # bench name -> (substring of the profiled kernel name, units in that profiled launch)
    "ppo_loss_categorical": ("ppo_loss_categorical_kernel", 1 << 22),
    "ppo_loss_normal": ("ppo_loss_normal_kernel", 1 << 22),
    "gae_scan": ("gae_scan_time_major_kernel", (1 << 20) * 32),
    "advantage_normalise": ("advantage_normalise_flat_kernel", (1 << 20) * 32),
    "rollout_step_dummy": ("rollout_step_dummy_kernel", 1 << 20),
    "rollout_stats": ("rollout_stats_kernel", (1 << 20) * 32),
    "gather_minibatch": ("gather_minibatch_kernel", 1 << 22),
    # the towers, profiled at 2^20 rows (policy tower of the discrete dummy env)
    "mlp_tower_forward": ("mlp_tower_forward_kernel<1, 2, false>", 1 << 20),
    "mlp_tower_forward_save": ("mlp_tower_forward_kernel<1, 2, true>", 1 << 20),
    "mlp_tower_backward": ("mlp_tower_backward_kernel<1, 2>", 1 << 20),
    "mlp_wgrad": ("mlp_wgrad_kernel", 1 << 20),
}
PMC_KERNEL_SPLIT = {  # the bf16-plane kernels, same profiled shapes
    "mlp_tower_forward": ("mlp_tower_forward_split_kernel<1, 2, false>", 1 << 20),
    "mlp_tower_forward_save": ("mlp_tower_forward_split_kernel<1, 2, true>", 1 << 20),
    "mlp_tower_backward": ("mlp_tower_backward_split_kernel<1, 2>", 1 << 20),
    "mlp_wgrad": ("mlp_wgrad_split_kernel<1, 2>", 1 << 20),
}


def pmc_traffic(name: str, units_per_launch: float, split: bool = False):
    try:
        summary = json.load(open(PMC_SUMMARY))
    except OSError:
        return None
    needle, units = (PMC_KERNEL_SPLIT if split else PMC_KERNEL).get(name, (None, 1))
    for kernel, rec in summary.items():
        if needle and needle in kernel:
            return rec["traffic_bytes_per_launch"] / units * units_per_launch
    return None


def parse_args() -> argparse.Namespace:
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=5)
    p.add_argument("--warmup", type=int, default=2)
    p.add_argument("--num-envs", type=int, default=1 << 20, help="environments PER GPU")
    p.add_argument("--horizon", type=int, default=32)
    p.add_argument("--env", default="discrete", choices=["discrete", "continuous", "cartpole", "mountain_car", "pendulum"])
    p.add_argument("--distribution", default="default", choices=["default", "squashed"])
    p.add_argument("--recurrent", action="store_true", help="RecurrentAlgorithmConfig (LSTM, seq_len 4)")
    p.add_argument("--cpu-baseline-seconds", type=float, default=20.0)
    p.add_argument("--no-cpu-baseline", action="store_true")
    return p.parse_args()


def cpu_baseline(budget_s: float) -> dict:
    """The CPU restatement of the same algorithm (oracle/ppo_cpu.py: C kernels +
    torch-CPU MLP) on BASELINE config 1 (DiscreteDummyEnv, N=8192, H=32,
    defaults), host cores of this box. Bounded: one warm-up iteration, then
    whole iterations until ~budget_s seconds are spent."""
    from oracle.ppo_cpu import OraclePPO

    cores = torch.get_num_threads()
    torch.manual_seed(0)
    algo = OraclePPO("discrete", num_envs=8192, horizon=32)
    algo.collect()
    algo.step()
    iters, t0 = 0, time.perf_counter()
    while True:
        algo.collect()
        algo.step()
        iters += 1
        elapsed = time.perf_counter() - t0
        if elapsed >= budget_s or iters >= 50:
            break
    return {
        "value": 8192 * 32 * iters / elapsed,
        "unit": "env transitions/sec",
        "policy_updates_per_sec": iters / elapsed,
        "cores": cores,
        "host_cpus": os.cpu_count(),
        "kind": "port",
        "sample": f"{iters} x (collect+step), DiscreteDummyEnv num_envs=8192 horizon=32 defaults"
                  f" (BASELINE configs[0]), {elapsed:.1f} s, oracle C kernels + torch-CPU MLP",
    }


def main() -> None:
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    from rl8_amd import AlgorithmConfig, RecurrentAlgorithmConfig, hip
    from rl8_amd.distributions import SquashedNormal
    from rl8_amd.env import ContinuousDummyEnv, DiscreteDummyEnv

    if args.env == "cartpole":
        from rl8_amd.envs.cartpole import CartPole as env_cls
    elif args.env == "mountain_car":
        from rl8_amd.envs import MountainCar as env_cls
    elif args.env == "pendulum":
        from rl8_amd.envs import Pendulum as env_cls
    else:
        env_cls = DiscreteDummyEnv if args.env == "discrete" else ContinuousDummyEnv

    torch.manual_seed(0)
    global_envs = args.num_envs * world
    config_cls = RecurrentAlgorithmConfig if args.recurrent else AlgorithmConfig
    extra = {"distribution_cls": SquashedNormal} if args.distribution == "squashed" else {}
    algo = config_cls(num_envs=global_envs, horizon=args.horizon, **extra).build(env_cls)
    horizon = algo.hparams.horizon

    def barrier() -> None:
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        algo.collect()
        algo.step()

    hip.timer.reset()
    hip.timer.enabled = True
    barrier()
    t0 = time.perf_counter()
    collect_ms = step_ms = 0.0
    for _ in range(args.steps):
        c = algo.collect()
        s = algo.step()
        collect_ms += c["profiling/collect_ms"]
        step_ms += s["profiling/step_ms"]
    barrier()
    elapsed = time.perf_counter() - t0
    hip.timer.enabled = False
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)

    kernels = {}
    obs_dim = int(algo.env.observation_spec.shape[0])
    model = algo.policy.model
    if hasattr(model, "feature_model"):      # default discrete model: logits tower + value tower
        tower_heads = [model.feature_model[2].out_features, 1]
    elif hasattr(model, "action_mean"):      # default continuous model: (mean | log_std) tower + value tower
        tower_heads = [model.action_mean.out_features + model.action_log_std.out_features, 1]
    else:
        tower_heads = []
    from rl8_amd.nn import fused_mlp

    def tower_gemm(name: str) -> str:
        """Which matrix pipe this tower kernel ran its 256x256 product on."""
        if not tower_heads:
            return "f32"
        widths = [(obs_dim, n) for n in tower_heads]
        if name in ("mlp_tower_forward", "mlp_tower_forward_save"):
            ok = fused_mlp.FORWARD_GEMM == "split" and all(hip.mlp_forward_split_supports(d, n) for d, n in widths)
        elif name == "mlp_wgrad":  # the bf16-plane weight-gradient kernel takes any width
            ok = fused_mlp.BACKWARD_GEMM == "split"
        else:
            ok = fused_mlp.BACKWARD_GEMM == "split" and all(hip.mlp_backward_split_supports(d, n) for d, n in widths)
        return "bf16x3-split" if ok else "f32"

    for name, rec in hip.timer.summary().items():
        if name.startswith("mlp_"):
            # n_out differs per tower (policy 2-3, value 1); price both at the mean.
            # mlp_wgrad is the 256x256 weight-gradient product alone.
            per_row = 2.0 * 256 * 256 if name == "mlp_wgrad" else tower_flops_per_row(obs_dim, 1.5)
            flops_per_launch = per_row * rec["units_per_launch"]
            tflops = flops_per_launch / (rec["avg_ms"] * 1e-3) / 1e12
            gemm = tower_gemm(name)
            kernels[name] = {
                "bound": "mfma",
                "gemm": gemm,
                "launches": rec["launches"],
                "avg_ms": round(rec["avg_ms"], 5),
                "total_ms": round(rec["total_ms"], 3),
                "algorithmic_flop_per_launch": flops_per_launch,
                "achieved_TFLOPs": round(tflops, 2),            # fp32-equivalent: algorithmic flops / time
                "frac_of_f32_mfma_peak": round(tflops / MFMA_F32_PEAK_TFLOPS, 4),
                "pmc_traffic_bytes_per_launch": pmc_traffic(name, rec["units_per_launch"], gemm != "f32"),
            }
            if gemm != "f32":
                # every fp32 multiply-add of the 256x256 product is SPLIT_PRODUCTS bf16 ones on the matrix pipe
                executed = SPLIT_PRODUCTS * 2.0 * 256 * 256 * rec["units_per_launch"]
                bf16_tflops = executed / (rec["avg_ms"] * 1e-3) / 1e12
                kernels[name].update({
                    "executed_bf16_flop_per_launch": executed,
                    "executed_bf16_TFLOPs": round(bf16_tflops, 1),
                    "frac_of_bf16_mfma_peak": round(bf16_tflops / MFMA_BF16_PEAK_TFLOPS, 4),
                })
            continue
        bytes_per_launch = ALGORITHMIC_BYTES.get(name, 0.0) * rec["units_per_launch"]
        gbs = bytes_per_launch / (rec["avg_ms"] * 1e-3) / 1e9 if rec["avg_ms"] > 0 else 0.0
        kernels[name] = {
            "bound": "hbm",
            "launches": rec["launches"],
            "avg_ms": round(rec["avg_ms"], 5),
            "total_ms": round(rec["total_ms"], 3),
            "algorithmic_bytes_per_launch": bytes_per_launch,
            "achieved_GBps": round(gbs, 1),
            "frac_of_8TBps": round(gbs / HBM_PEAK_GBS, 4),
            "pmc_traffic_bytes_per_launch": pmc_traffic(name, rec["units_per_launch"]),
        }

    if rank == 0:
        hbm_dominant = "ppo_loss_categorical" if "ppo_loss_categorical" in kernels else "ppo_loss_normal"
        variant = ("Recurrent" if args.recurrent else "") + (" SquashedNormal" if args.distribution == "squashed" else "")
        dom = kernels[hbm_dominant]
        # The kernel the timed region spends most of its time in.
        dominant = max(kernels, key=lambda k: kernels[k]["total_ms"])
        top = kernels[dominant]
        if top["bound"] == "mfma" and top["gemm"] != "f32":
            # bf16-plane kernel: priced in the bf16 multiply-adds the matrix pipe executes
            # (6 per fp32 multiply-add of the algorithm) against the dense bf16 peak
            roofline = {
                "kernel": f"rl8_{dominant}_split_f32",
                "bound": "mfma",
                "achieved": top["executed_bf16_TFLOPs"],
                "peak": MFMA_BF16_PEAK_TFLOPS,
                "unit": "TFLOP/s",
                "frac": top["frac_of_bf16_mfma_peak"],
                "flop_per_launch": top["executed_bf16_flop_per_launch"],
                "flop_definition": "6 bf16 plane products x 2*256*256 per row (fp32 operands split exactly into"
                                   " 3 bf16 planes, fp32 accumulate)",
                "f32_equivalent_TFLOPs": top["achieved_TFLOPs"],   # the algorithm's fp32 FLOP / time
                "f32_mfma_peak_TFLOPs": MFMA_F32_PEAK_TFLOPS,
                "f32_equivalent_frac_of_f32_mfma_peak": top["frac_of_f32_mfma_peak"],  # > 1: beyond the fp32 matrix roofline
                "bf16_sustained_TFLOPs_measured": 1750.0,          # tools/probes/bf16_split_probe.hip, changing operands
                "frac_of_sustained": round(top["executed_bf16_TFLOPs"] / 1750.0, 4),
                "algorithmic_flop_per_launch": top["algorithmic_flop_per_launch"],
                "avg_launch_ms": top["avg_ms"],
                "launches": top["launches"],
                "share_of_step_ms": round(top["total_ms"] / args.steps, 2),
                "traffic": top["pmc_traffic_bytes_per_launch"],
            }
        elif top["bound"] == "mfma":
            roofline = {
                "kernel": f"rl8_{dominant}_f32",
                "bound": "mfma",
                "achieved": top["achieved_TFLOPs"],
                "peak": MFMA_F32_PEAK_TFLOPS,
                "unit": "TFLOP/s",
                "frac": top["frac_of_f32_mfma_peak"],
                "algorithmic_flop_per_launch": top["algorithmic_flop_per_launch"],
                "avg_launch_ms": top["avg_ms"],
                "launches": top["launches"],
                "share_of_step_ms": round(top["total_ms"] / args.steps, 2),
                "traffic": top["pmc_traffic_bytes_per_launch"],
            }
        else:
            roofline = None
        transitions = global_envs * horizon * args.steps
        line = {
            "metric": "env transitions/sec + policy updates/sec, DiscreteDummyEnv num_envs=2^20 h=32",
            "value": transitions / elapsed,
            "unit": "env transitions/sec",
            "policy_updates_per_sec": args.steps / elapsed,
            "optimizer_steps_per_sec": args.steps * algo.hparams.num_sgd_iters * algo.hparams.num_minibatches / elapsed,
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "collect_ms_per_step": collect_ms / args.steps,
            "update_ms_per_step": step_ms / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "gemm": sorted({k["gemm"] for k in kernels.values() if k["bound"] == "mfma"}),
            "data": "synthetic (Philox-reset DiscreteDummyEnv states, random-init default MLP)",
            "config": {
                "workload": f"{env_cls.__name__}{variant} collect()+step(), num_envs={args.num_envs} per GPU"
                            f" ({global_envs} total), horizon={horizon}, AlgorithmConfig defaults"
                            " (4 SGD iters, one full-buffer minibatch, Adam 1e-3)",
                "num_envs_per_gpu": args.num_envs,
                "horizon": horizon,
                "parallelism": f"env-sharded x{world}",
            },
            "roofline": roofline if roofline is not None else {
                "kernel": f"rl8_{hbm_dominant}_fwd_bwd_f32",
                "bound": "hbm",
                "achieved": dom["achieved_GBps"],
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(dom["achieved_GBps"] / HBM_PEAK_GBS, 4),
                "frac_of_measured_copy_ceiling": round(dom["achieved_GBps"] / HBM_COPY_CEILING_GBS, 4),
                "algorithmic_bytes_per_launch": dom["algorithmic_bytes_per_launch"],
                "avg_launch_ms": dom["avg_ms"],
                "launches": dom["launches"],
                "traffic": dom["pmc_traffic_bytes_per_launch"],
            },
            "roofline_hbm": {
                "kernel": f"rl8_{hbm_dominant}_fwd_bwd_f32",
                "bound": "hbm",
                "achieved": dom["achieved_GBps"],
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(dom["achieved_GBps"] / HBM_PEAK_GBS, 4),
                "algorithmic_bytes_per_launch": dom["algorithmic_bytes_per_launch"],
                "avg_launch_ms": dom["avg_ms"],
                "launches": dom["launches"],
                "traffic": dom["pmc_traffic_bytes_per_launch"],
            },
            "kernels": kernels,
            "hand_kernel_ms_per_step": round(sum(k["total_ms"] for k in kernels.values()) / args.steps, 3),
            "fused_towers": True,
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.cpu_baseline_seconds)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
