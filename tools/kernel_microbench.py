"""Times every hand-written kernel at BASELINE config-2 shapes (N = 2^20 envs,
H = 32) with HIP events, several rounds interleaved in one process, and prints
achieved algorithmic GB/s against the 8 TB/s HBM3E peak. Also the target of the
rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE) whose summaries live in profiles/.

    python tools/kernel_microbench.py [--rounds 20] [--envs 1048576] [--only gae_scan,...]
"""

from __future__ import annotations

import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch

from rl8_amd import hip

p = argparse.ArgumentParser()
p.add_argument("--rounds", type=int, default=20)
p.add_argument("--envs", type=int, default=1 << 20)
p.add_argument("--horizon", type=int, default=32)
p.add_argument("--loss-rows", type=int, default=1 << 22)
p.add_argument("--only", default="")
args = p.parse_args()
N, H, M = args.envs, args.horizon, args.loss_rows
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
f32 = lambda gamma: float(np.float32(gamma))  # noqa: E731

# --- operands (time-major buffer, like Algorithm's) --------------------------
state0 = torch.empty(N, 1, device=dev).uniform_(-100, 100, generator=g)
rewards = -(state0.T + torch.randn(H + 1, N, device=dev, generator=g).cumsum(0)).abs().contiguous()
values = torch.randn(H + 1, N, device=dev, generator=g)
adv = torch.empty_like(rewards)
ret = torch.empty_like(rewards)
rdr = torch.randn(H + 1, N, device=dev, generator=g)
rewards_em = rewards.T.contiguous()
values_em = values.T.contiguous()
adv_em = torch.empty_like(rewards_em)
ret_em = torch.empty_like(rewards_em)

logits = torch.randn(M, 1, 2, device=dev, generator=g) * 0.5
value_m = torch.randn(M, 1, device=dev, generator=g)
ret_m = value_m + torch.randn(M, 1, device=dev, generator=g)
action_m = torch.randint(0, 2, (M, 1), device=dev, generator=g)
logp_m = torch.full((M, 1), -0.69, device=dev) + torch.randn(M, 1, device=dev, generator=g) * 0.1
adv_m = torch.randn(M, 1, device=dev, generator=g)
hp = hip.ppo_hparams(clip_param=0.2, dual_clip_param=None, entropy_coeff=0.0, vf_clip_param=5.0, vf_coeff=1.0,
                     grad_scale=1.0 / M)
mean_m = torch.randn(M, 1, device=dev, generator=g)
ls_m = torch.tanh(torch.randn(M, 1, device=dev, generator=g))
act_f = torch.tanh(mean_m + torch.randn(M, 1, device=dev, generator=g))

step_logits = torch.randn(N, 1, 2, device=dev, generator=g) * 1e-3
step_value = torch.randn(N, 1, device=dev, generator=g)
state = state0.clone()
cols = {k: torch.empty(N, 1, device=dev) for k in ("logp", "value", "reward", "obs", "rdr1")}
action_col = torch.empty(N, 1, dtype=torch.int64, device=dev)
rdr0 = torch.randn(N, 1, device=dev, generator=g)

cp_state = torch.randn(4, N, device=dev, generator=g) * 0.05
cp_logits = torch.randn(N, 1, 3, device=dev, generator=g)
cp_obs = torch.empty(N, 5, device=dev)
cp_cfg = hip.CartPoleCfg(5.0, 9.8, 0.5, 0.1, 0.05, 1.1, 0.02, 0)

perm = torch.randperm(N * H, device=dev, generator=g)[:M]
obs_leaf = torch.randn(H + 1, N, 1, device=dev, generator=g).transpose(0, 1)
act_leaf = torch.randint(0, 2, (H + 1, N, 1), device=dev, generator=g).transpose(0, 1)
f_leaf = [torch.randn(H + 1, N, 1, device=dev, generator=g).transpose(0, 1) for _ in range(3)]
moments = torch.tensor([float(N * H), 0.0, float(N * H)], dtype=torch.float64, device=dev)
packed_rows = hip.PackedSamples(H, [obs_leaf, act_leaf, *f_leaf])


def gae_tm():
    return hip.gae_scan(rewards, values, adv, ret, layout=1, n=N, h=H, gamma=f32(0.95), gamma_lambda=f32(0.9025),
                        reward_denominator=f32(30.0), write_scaled_rewards=False)


def gae_em():
    return hip.gae_scan(rewards_em, values_em, adv_em, ret_em, layout=0, n=N, h=H, gamma=f32(0.95),
                        gamma_lambda=f32(0.9025), reward_denominator=f32(30.0), write_scaled_rewards=False)


mlp_x = torch.randn(N, 1, device=dev, generator=g) * 30
mlp_w1 = torch.randn(256, 1, device=dev, generator=g)
mlp_b1 = torch.randn(256, device=dev, generator=g)
mlp_w2 = torch.randn(256, 256, device=dev, generator=g) / 16
mlp_b2 = torch.randn(256, device=dev, generator=g)
mlp_w3 = torch.randn(2, 256, device=dev, generator=g) / 16
mlp_b3 = torch.randn(2, device=dev, generator=g)
mlp_w2p = hip.mlp_pack_w2(mlp_w2)
MLP_FLOP = 2 * N * (256 * 1 + 256 * 256 + 256 * 2)


def torch_tower():
    h = torch.relu(torch.addmm(mlp_b1, mlp_x, mlp_w1.T))
    h = torch.relu(torch.addmm(mlp_b2, h, mlp_w2.T))
    return torch.addmm(mlp_b3, h, mlp_w3.T)


_, mlp_h1, mlp_h2 = hip.mlp_tower_forward(mlp_x, mlp_w1, mlp_b1, mlp_w2p, mlp_b2, mlp_w3, mlp_b3, save=True)
mlp_w2tp = hip.mlp_pack_w2(mlp_w2, transposed=True)
mlp_dout = torch.randn(N, 2, device=dev, generator=g) / N

mlp_dz2 = torch.randn(N, 256, device=dev, generator=g)
mlp_zero = torch.zeros(N, 256, device=dev)

# plane kernels (fp32 operands as scaled fp16 planes: 3 plane products per 16 k; 2 with the ReLU gate as an operand)
import ctypes as C  # noqa: E402

mlp_w2h = hip.mlp_pack_w2_f16(mlp_w2)  # scaled fp16 two-plane pack
mlp_gate = hip.mlp_tower_forward_split(mlp_x, mlp_w1, mlp_b1, mlp_w2h, mlp_b2, mlp_w3, mlp_b3, save=True, save_h1=False,
                                       save_gate=True)[3]
_lib = hip.load()
_partials = torch.empty(int(_lib.rl8_mlp_backward_max_rows()), int(_lib.rl8_mlp_backward_partial_floats(1, 2)), device=dev)
_dw2 = torch.empty(256, 256, device=dev)
_ws = torch.empty(int(_lib.rl8_mlp_wgrad_workspace_bytes()) // 4, device=dev)
_rows = C.c_int(0)
_p = hip._ptr


mlp_w2th = hip.mlp_pack_w2_f16(mlp_w2, transposed=True)


def f16_dgrad():  # fp16-plane data-gradient kernel (fused mode)
    _lib.rl8_mlp_tower_backward_f16_f32(_p(mlp_x), _p(mlp_w1), _p(mlp_b1), _p(mlp_dout), N, 1, _p(mlp_w2th), _p(mlp_w3), 2,
                                        _p(_partials), C.byref(_rows), _p(mlp_gate), hip._stream())


# the value tower (one output): its weight gradient runs the gate-plane kernel
mlp_w3v = torch.randn(1, 256, device=dev, generator=g) / 16
mlp_doutv = torch.randn(N, 1, device=dev, generator=g) / N
_partials_v = torch.empty(int(_lib.rl8_mlp_backward_max_rows()), int(_lib.rl8_mlp_backward_partial_floats(1, 1)), device=dev)


def gate_wgrad_fused():  # single-output tower: gate plane x three planes of dOut * h1 (3 products per 16 samples)
    _lib.rl8_mlp_wgrad_fused_split_f32(_p(mlp_h2), _p(mlp_doutv), _p(mlp_x), _p(mlp_w1), _p(mlp_b1), _p(mlp_w3v), N, 1, 1,
                                       _p(_ws), _p(_dw2), _p(_partials_v), hip._stream())


mlp_w2tg = hip.mlp_pack_w2_f16_gate(mlp_w2, mlp_w3v)  # B of the gate-mode data gradient (value tower)


def f16_dgrad_gate():  # single-output tower: gate plane x two planes of w3 * W2 (2 products per 16 k)
    _lib.rl8_mlp_tower_backward_gate_f16_f32(_p(mlp_x), _p(mlp_w1), _p(mlp_b1), _p(mlp_doutv), N, 1, _p(mlp_w2tg), 1,
                                             _p(_partials_v), C.byref(_rows), _p(mlp_gate), hip._stream())


def f16_forward_gate_only():  # training forward of a rank-one head: gate bits only, no h2
    hip.mlp_tower_forward_split(mlp_x, mlp_w1, mlp_b1, mlp_w2h, mlp_b2, mlp_w3, mlp_b3, save=True, save_gate=True, save_h2=False)


def gate_bits_wgrad():  # weight gradient of a single-output head from the gate bits alone (dW3 from the sums)
    _lib.rl8_mlp_wgrad_gate_bits_f32(_p(mlp_gate), _p(mlp_doutv), _p(mlp_x), _p(mlp_w1), _p(mlp_b1), _p(mlp_w2), _p(mlp_b2),
                                     _p(mlp_w3v), N, 1, 1, _p(_ws), _p(_dw2), _p(_partials_v), hip._stream())


def split_wgrad_fused():  # weight-gradient kernel: re-forms dZ2 and h1, accumulates the head gradients
    _lib.rl8_mlp_wgrad_fused_split_f32(_p(mlp_h2), _p(mlp_dout), _p(mlp_x), _p(mlp_w1), _p(mlp_b1), _p(mlp_w3), N, 1, 2,
                                       _p(_ws), _p(_dw2), _p(_partials), hip._stream())


KERNELS = {
    # bf16-plane kernels: "GB/s" column = fp32-equivalent TFLOP/s (algorithmic FLOP / 1000 as bytes)
    "mlp_tower_forward_f16": (lambda: hip.mlp_tower_forward_split(mlp_x, mlp_w1, mlp_b1, mlp_w2h, mlp_b2, mlp_w3, mlp_b3), MLP_FLOP / 1000),
    "mlp_tower_forward_save_f16": (lambda: hip.mlp_tower_forward_split(mlp_x, mlp_w1, mlp_b1, mlp_w2h, mlp_b2, mlp_w3, mlp_b3, save=True, save_h1=False, save_gate=True), MLP_FLOP / 1000),
    "mlp_tower_backward_f16": (f16_dgrad, MLP_FLOP / 1000),
    "mlp_wgrad_fused_split": (split_wgrad_fused, 2 * N * 65536 / 1000),
    "mlp_wgrad_fused_gate": (gate_wgrad_fused, 2 * N * 65536 / 1000),
    "mlp_wgrad_gate_bits": (gate_bits_wgrad, 2 * N * 65536 / 1000),
    "mlp_tower_forward_gate_only_f16": (f16_forward_gate_only, MLP_FLOP / 1000),
    "mlp_tower_backward_gate_f16": (f16_dgrad_gate, MLP_FLOP / 1000),
    "mlp_wgrad_split": (lambda: hip.mlp_wgrad_split(mlp_dz2, mlp_x, mlp_w1, mlp_b1), 2 * N * 65536 / 1000),
    "mlp_wgrad_fused": (lambda: hip.mlp_wgrad(mlp_dz2, mlp_h1), 2 * N * 65536 / 1000),
    # same launch on all-zero operands: the gap to the line above is clock / power, not the kernel
    "mlp_wgrad_zeros": (lambda: hip.mlp_wgrad(mlp_zero, mlp_zero), 2 * N * 65536 / 1000),
    "mlp_wgrad_torch": (lambda: mlp_dz2.t() @ mlp_h1, 2 * N * 65536 / 1000),
    "mlp_tower_forward_save": (lambda: hip.mlp_tower_forward(mlp_x, mlp_w1, mlp_b1, mlp_w2p, mlp_b2, mlp_w3, mlp_b3, save=True), MLP_FLOP / 1000),
    "mlp_tower_backward_fused": (lambda: hip.mlp_tower_backward(mlp_x, mlp_h1, mlp_h2, mlp_dout, mlp_w2tp, mlp_w3), 2 * MLP_FLOP / 1000),
    # the two MLP entries report TFLOP/s in the GB/s column (algorithmic "bytes" = FLOP / 1000)
    "mlp_tower_forward_fused": (lambda: hip.mlp_tower_forward(mlp_x, mlp_w1, mlp_b1, mlp_w2p, mlp_b2, mlp_w3, mlp_b3), MLP_FLOP / 1000),
    "mlp_tower_forward_torch": (torch_tower, MLP_FLOP / 1000),
    # name: (callable, algorithmic bytes per launch)
    "gae_scan_time_major": (gae_tm, 16 * N * H + 8 * N),
    "gae_scan_env_major_lds": (gae_em, 16 * N * H + 8 * N),
    "advantage_normalise": (lambda: hip.advantage_normalise(adv, layout=1, n=N, h=H, moments=moments), 8 * N * H),
    "ppo_loss_categorical": (lambda: hip.ppo_loss_categorical(logits, value_m, action_m, logp_m, adv_m, ret_m, hp), 44 * M),
    "ppo_loss_squashed_normal": (lambda: hip.ppo_loss_normal(mean_m, ls_m, value_m, act_f, logp_m, adv_m, ret_m, hp, squashed=True), 40 * M),
    "rollout_step_dummy": (lambda: hip.rollout_step_dummy(
        discrete=True, squashed=False, features=step_logits, features2=None, value=step_value, noise=None, state=state,
        action_col=action_col, logp_col=cols["logp"], value_col=cols["value"], reward_col=cols["reward"],
        obs_col_next=cols["obs"], rdr_t=rdr0, rdr_t1=cols["rdr1"], gamma=f32(0.95), seed=1, step=0, env_offset=0,
        deterministic=False), 52 * N),
    "rollout_step_cartpole": (lambda: hip.rollout_step_cartpole(
        logits=cp_logits, value=step_value, noise=None, state=cp_state, cfg=cp_cfg, action_col=action_col,
        logp_col=cols["logp"], value_col=cols["value"], reward_col=cols["reward"], obs_col_next=cp_obs, rdr_t=rdr0,
        rdr_t1=cols["rdr1"], gamma=f32(0.95), seed=1, step=0, env_offset=0, deterministic=False), 88 * N),
    "rollout_stats": (lambda: hip.rollout_stats(rewards.T.unsqueeze(-1), rdr.T.unsqueeze(-1)), 8 * N * H),
    "gather_minibatch": (lambda: hip.gather_minibatch(perm, H, [obs_leaf, act_leaf, *f_leaf]), 56 * M),
    # the same minibatch out of rows packed once per step (24 B read + 32 B written per sample of the buffer)
    "gather_packed": (lambda: packed_rows.gather(perm), 56 * M),
    "pack_samples": (lambda: hip.PackedSamples(H, [obs_leaf, act_leaf, *f_leaf]), 56 * N * H),
}
only = [s for s in args.only.split(",") if s]
names = [k for k in KERNELS if not only or k in only]

REPS = 10  # launches per captured graph
for name in names:  # warm-up (also first-touch of outputs)
    KERNELS[name][0]()
torch.cuda.synchronize()
# Capture REPS back-to-back launches of each kernel into a HIP graph: replaying
# it removes the Python / ctypes launch path from the measurement, so the events
# bracket device execution only.
graphs = {}
side = torch.cuda.Stream()
for name in names:
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        for _ in range(REPS):
            KERNELS[name][0]()
    graphs[name] = graph
torch.cuda.synchronize()
# Clocks: a cold MI355X sits in a low-power state; spin it up with ~2 s of GEMMs
# and a few untimed replays before measuring (the real workload runs hot).
import time

wa = torch.randn(8192, 8192, device=dev)
t_end = time.time() + 2.0
while time.time() < t_end:
    for _ in range(10):
        wa @ wa
    torch.cuda.synchronize()
for _ in range(3):
    for name in names:
        graphs[name].replay()
torch.cuda.synchronize()
times = {k: [] for k in names}
for _ in range(args.rounds):
    for name in names:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        graphs[name].replay()
        b.record()
        times[name].append((a, b))
torch.cuda.synchronize()
out = {}
for name in names:
    ms = sorted(a.elapsed_time(b) / REPS for a, b in times[name])
    med, mn = ms[len(ms) // 2], ms[0]
    nbytes = KERNELS[name][1]
    out[name] = {"median_us": round(med * 1e3, 2), "min_us": round(mn * 1e3, 2), "algorithmic_MB": round(nbytes / 1e6, 1),
                 "GBps_median": round(nbytes / med / 1e6, 1), "frac_8TBps": round(nbytes / med / 1e6 / 8000, 3)}
    print(f"{name:28s} {med*1e3:9.1f} us  {nbytes/1e6:8.1f} MB  {nbytes/med/1e6:8.1f} GB/s  ({nbytes/med/1e6/80:.1f}% of 8 TB/s)")
print(json.dumps(out))
