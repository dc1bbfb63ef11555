"""Headline benchmark: env transitions/sec (+ policy updates/sec) of
``collect(); step()`` on DiscreteDummyEnv, num_envs = 2^20 per GPU, horizon 32,
all ``AlgorithmConfig`` defaults (BASELINE.json configs[1]).

    python bench.py --gpus 1 --steps 5 --warmup 2
    python bench.py --gpus 8                      # starts its own 8 ranks (below)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \\
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = one ``collect()`` (32 timesteps of policy forward + fused sample /
env.step / bookkeeping launch, bootstrap value, stats) + one ``step()`` (GAE,
4 SGD iterations of policy forward, fused PPO loss fwd+bwd, policy backward,
clip, Adam). Everything lives in HBM before the timed region starts.

Ranks. One process per GPU. Under ``torchrun`` the ranks exist already
(``WORLD_SIZE`` in the environment). Called plainly with ``--gpus N > 1`` this
process touches no GPU: it starts N children of this same script with
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT set, waits,
and prints rank 0's JSON line (``launch_ranks``).

Scaling. ``--scaling weak`` (default, what the driver's ``bench.py --gpus N``
runs): every rank owns ``--num-envs`` environments -- 2^20 per GPU, 2^23 in
total at N = 8.  ``--scaling strong``: ``--num-envs`` is the GLOBAL count, split
over the ranks.  BASELINE.json's ``north_star`` quotes ONE problem (num_envs =
2^20 in total) at 1 / 2 / 4 / 8 GPUs: THAT table is

    python bench.py --gpus N --scaling strong --num-envs 1048576

(configs[3]: add ``--env continuous --distribution squashed``; configs[4]:
``--recurrent --num-envs 65536 --horizon 256``).  Every line names both
conventions under ``config.scaling_conventions``.  Either way the environments
are sharded and the only traffic between ranks is the RCCL all-reduce of moments
and of [gradient | loss sums] per optimizer step.

Secondary configurations. With the headline configuration on one GPU (the
defaults) the line also carries ``secondary``: BASELINE configs[2..4] --
CartPole 2^18 x 128, ContinuousDummyEnv + SquashedNormal 2^20 x 32, the
recurrent algorithm at 2^16 x 256 -- measured after the headline's timed region
in the same process, a few steps each under the same timing rules (value,
ms_per_step, dominant kernel, frac / executed_frac per configuration;
``--no-secondary`` skips them, ``--secondary-steps`` sets their length).

Besides the contract's fields, the JSON line carries
  roofline      the kernel the timed region spends most time in (whichever it is
                in this run -- with the defaults one of the three tower kernels:
                MFMA-bound, priced in the 16-bit multiply-adds the matrix pipe
                executes against the 2.5 PF dense peak), timed with HIP events
                inside the timed region; ``roofline_hbm`` the same for the largest
                HBM-bound hand kernel (fused PPO loss fwd+bwd, 44 B per sample);
  kernels       the same for every hand kernel that ran;
  cpu_baseline  the CPU restatement (oracle/, kind "port") of the same algorithm
                on the host cores, best of a thread-count sweep, on a bounded
                sample (rank 0, N=1 only);
  world_size / backend / collectives_per_step   as ``torch.distributed`` saw them;
  rank_ms_per_step       every rank's own time per step (min / max / list): a
                         straggler shows here, ``ms_per_step`` is the max.

Self-launched ranks run in their own sessions with a parent-death signal; the
parent forwards SIGTERM / SIGINT to them, stops everything at ``--launch-timeout``
and, when a rank fails, prints the tail of that rank's (and rank 0's) stderr.
"""

from __future__ import annotations

import argparse
import json
import os
import signal
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# torch is imported inside run(): the launcher parent (launch_ranks) never needs it
# and must not initialise a GPU.

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); measured copy ceiling ~6290
HBM_COPY_CEILING_GBS = 6290.0
MFMA_F32_PEAK_TFLOPS = 157.3  # dense fp32 MFMA (= fp32 vector) peak, MI355X_MICROARCH.md
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense 16-bit MFMA peak (spec), MI355X_MICROARCH.md
# What the chip holds under power with random operands, two waves per SIMD (tools/probes/mfma_shape_probe.hip, round 3:
# 16x16x32 f16 1677-1687 TFLOP/s at 1.82 GHz, 32x32x16 1483-1520 at 1.72 GHz; round 1's 1750 was on constant-ish data)
MFMA_16BIT_SUSTAINED_TFLOPS = 1680.0
PLANE_PRODUCTS = {"bf16x3-split": 6, "f16x2-split": 3, "bf16-gate-x3": 3, "f16-gate-x2": 2, "f16-gatebits-x2": 2}  # 16-bit MFMA products per fp32 product, by scheme

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
    "gather_minibatch": 56.0,       # index 8 + 24 read + 24 written per sample
    "gather_packed": 56.0,          # the same minibatch out of rows packed once per step()
    "pack_samples": 56.0,           # 24 B read + 32 B written per sample of the buffer, once per step()
    # the recurrent models' heads: a row of the LSTM's output (256 floats) in, the logits + value (3 floats) out /
    # their gradient in (these lines printed 0 GB/s for 12 ms of kernels)
    # opt-in piecewise towers (scalar observation): x in, the outputs out / x and dOut in (n_out 1 and 2: 1.5 on average)
    "pw_tower_forward": 4.0 + 6.0,
    "pw_segment_sums": 4.0 + 6.0,
    "linear_heads_forward": 1024.0 + 12.0,
    "linear_heads_backward": 1024.0 + 12.0,
}

# Fabric traffic of the recurrent bench's kernels (rocprofv3 --pmc TCC_EA0_* per launch at 8 192 envs x 256 steps,
# tools/diag/recurrent_bench_pmc.sh -> profiles/r03_cfg5_fabric_traffic.txt): bytes per unit of the launch that was profiled
FABRIC_SUMMARY = next((p for p in (os.path.join(ROOT, "profiles", f"r{r:02d}_cfg5_fabric_traffic.txt") for r in (6, 5, 4, 3))
                       if os.path.exists(p)), "")
FABRIC_KERNEL = {  # bench name -> (substring of the profiled kernel's name, units of that launch)
    "lstm_rows_backward": ("lstm_rows_backward_heads", 1 << 21),   # (heads16_kernel since round 6, heads_kernel before)
    "lstm_step_save": ("lstm_step_split_kernel<1, true>", 1 << 19),
    "lstm_wgrad": ("mlp_wgrad_loadh16_kernel<1>", 1 << 19),        # (four gates per launch; mlp_wgrad_split_kernel<1, 0, true, true> in r03)
    "linear_heads_forward": ("linear_heads_forward_kernel<3>", 1 << 21),
    "linear_heads_backward": ("linear_heads_backward_kernel<3>", 1 << 21),
}


def fabric_traffic(name: str, units_per_launch: float):
    """Bytes one launch of ``units_per_launch`` units moves over the fabric, scaled from the profiled launch."""
    needle, units = FABRIC_KERNEL.get(name, (None, 1))
    if not needle or not FABRIC_SUMMARY:
        return None
    try:
        for line in open(FABRIC_SUMMARY):
            if needle in line and "reads/launch" in line:
                reads = float(line.split("reads/launch")[1].split()[0])
                writes = float(line.split("writes/launch")[1].split()[0])
                return (reads * 128.0 + writes * 64.0) / units * units_per_launch
    except (OSError, ValueError, IndexError):
        pass
    return None


# HBM traffic from the PMC counters (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE
# passes over tools/kernel_microbench.py at config-2 shapes; corrected as
# MI355X_MICROARCH.md prescribes; summary committed under profiles/). Scaled by
# units to the launch size bench.py uses.
PMC_SUMMARY = next((p for p in (os.path.join(ROOT, "profiles", f"r{r:02d}_pmc_traffic_microbench.json") for r in (6, 5, 4, 3, 2, 1))
                    if os.path.exists(p)), os.path.join(ROOT, "profiles", "r05_pmc_traffic_microbench.json"))
PMC_KERNEL = {  # bench name -> (substring of the profiled kernel name, units in that profiled launch)
    "ppo_loss_categorical": ("ppo_loss_categorical_kernel", 1 << 22),
    "ppo_loss_normal": ("ppo_loss_normal1_kernel", 1 << 22),  # (the one-action-dim vector kernel: what configs 4 launches)
    "gae_scan": ("gae_scan_time_major_kernel", (1 << 20) * 32),
    "advantage_normalise": ("advantage_normalise_flat_kernel", (1 << 20) * 32),
    "rollout_step_dummy": ("rollout_step_dummy_kernel", 1 << 20),
    "rollout_stats": ("rollout_stats_kernel", (1 << 20) * 32),
    "gather_minibatch": ("gather_minibatch_kernel", 1 << 22),
    "gather_packed": ("gather_packed_kernel", 1 << 22),
    "pack_samples": ("pack_samples_kernel", (1 << 20) * 32),
    # the towers, profiled at 2^20 rows (policy tower of the discrete dummy env)
    "mlp_tower_forward": ("mlp_tower_forward_kernel<1, 2, false>", 1 << 20),
    "mlp_tower_forward_save": ("mlp_tower_forward_kernel<1, 2, true>", 1 << 20),
    "mlp_tower_backward": ("mlp_tower_backward_kernel<1, 2>", 1 << 20),
    "mlp_wgrad": ("mlp_wgrad_kernel", 1 << 20),
}
PMC_KERNEL_F16 = {  # the fp16-plane kernels (forward, data gradient), same profiled shapes
    # (round 3: the rows-per-wave forward, mlp_rows_kernels.hip; template <d_in, n_out, SAVE, ring, diag>)
    "mlp_tower_forward": ("mlp_rows_forward_kernel<1, 2, 0, 4, 0>", 1 << 20),
    # (training forward of a rank-one head keeps the gate bits only: SAVE mode 2; mode 1 stores h2 as well)
    "mlp_tower_forward_save": ("mlp_rows_forward_kernel<1, 2, 2, 4, 0>", 1 << 20),
    # (round 4: the rollout's launches run in that save mode into the slabs SGD iteration 0 replays: fused_mlp.RolloutRecord)
    "mlp_tower_forward_record": ("mlp_rows_forward_kernel<1, 2, 2, 4, 0>", 1 << 20),
    # (round 5: general heads -- configs[3]'s mean / log_std tower -- run the rows-shape data gradient <d_in, KOUT, ring> and
    # the sixteen-wave weight gradient <d_in, n_out>; "mlp_tower_backward_f16_kernel<1, 2>" in older summaries)
    "mlp_tower_backward": ("mlp_rows_backward_general_kernel<1, 2, 3>", 1 << 20),
    "mlp_wgrad": ("mlp_wgrad_fused16_kernel<1, 2>", 1 << 20),
}
PMC_KERNEL_SPLIT = {  # the six-product bf16-plane weight gradient (RL8_WGRAD_PLANES=bf16), same profiled shape
    # (round 5 dropped the F16 template parameter: "<1, 2, false, false>" in the r04 and older summaries)
    "mlp_wgrad": ("mlp_wgrad_split_kernel<1, 2, false>", 1 << 20),
}
PMC_KERNEL_SPLIT_OLD = {"mlp_wgrad": ("mlp_wgrad_split_kernel<1, 2, false, false>", 1 << 20)}
PMC_KERNEL_GATE = {  # gate-mode kernels (heads whose dZ2 is gate * d * w3e)
    # (template <d_in, PAIR, BITS, F16>: the value tower's kernel, gate bits, two fp16 planes)
    # (round 4: the sixteen-wave kernel, template <d_in, PAIR>; the eight-wave one <d_in, PAIR, BITS, F16> in older summaries)
    "mlp_wgrad_gate": ("mlp_wgrad_gate16_kernel<1, false>", 1 << 20),
    "mlp_tower_backward_gate": ("mlp_rows_backward_gate_kernel<1, 1, 4>", 1 << 20),
}


# Traffic that does NOT scale with the rows of a launch: the weight-gradient kernels end with one 256 x 256 fp32 slab per
# workgroup (one workgroup per CU) which a reduction kernel then reads back; the ABI call cuts its rows into segments of
# 2^23, each its own launch.  (scaling the 2^20-row profile linearly counted these 64 B/row instead of 8.)
WGRAD_SEGMENT_ROWS = 1 << 23
WGRAD_SLAB_BYTES = 256 * 256 * 256 * 4.0   # written by the 256 workgroups of one launch
PMC_FIXED_BYTES = {"mlp_wgrad_gate": WGRAD_SLAB_BYTES, "mlp_wgrad": WGRAD_SLAB_BYTES}


def pmc_traffic(name: str, units_per_launch: float, gemm: str = "f32"):
    """HBM bytes of one ABI call of ``units_per_launch`` rows, from the profiled launch of the same kernel: the part
    that scales with rows is scaled, the per-launch part (PMC_FIXED_BYTES) is counted once per launched segment."""
    try:
        summary = json.load(open(PMC_SUMMARY))
    except OSError:
        return None
    table = {"f16x2-split": PMC_KERNEL_F16, "bf16x3-split": PMC_KERNEL_SPLIT, "bf16-gate-x3": PMC_KERNEL_GATE,
             "f16-gate-x2": PMC_KERNEL_GATE, "f16-gatebits-x2": PMC_KERNEL_GATE}.get(gemm, PMC_KERNEL)
    needle, units = table.get(name, (None, 1))
    if table is PMC_KERNEL_SPLIT and needle and not any(needle in kernel for kernel in summary):
        needle, units = PMC_KERNEL_SPLIT_OLD.get(name, (None, 1))
    for kernel, rec in summary.items():
        if needle and needle in kernel:
            fixed = min(PMC_FIXED_BYTES.get(name, 0.0), rec["traffic_bytes_per_launch"])
            launches = max(1.0, -(-units_per_launch // WGRAD_SEGMENT_ROWS)) if fixed else 1.0
            return (rec["traffic_bytes_per_launch"] - fixed) / units * units_per_launch + fixed * launches
    return None


def parse_args(argv: None | list[str] = None) -> argparse.Namespace:
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=5)
    p.add_argument("--warmup", type=int, default=2)
    p.add_argument("--num-envs", type=int, default=1 << 20,
                   help="environments PER GPU (--scaling weak) or in TOTAL (--scaling strong)")
    p.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    p.add_argument("--horizon", type=int, default=32)
    p.add_argument("--env", default="discrete", choices=["discrete", "continuous", "cartpole", "mountain_car", "pendulum"])
    p.add_argument("--distribution", default="default", choices=["default", "squashed"])
    p.add_argument("--recurrent", action="store_true", help="RecurrentAlgorithmConfig (LSTM, seq_len 4)")
    p.add_argument("--minibatches", type=int, default=1,
                   help="K > 1: sgd_minibatch_size = num_envs * horizon / K per rank, i.e. the SHUFFLED path of step()"
                        " (randperm, pack_samples once, gather_packed per minibatch: reference src/rl8/_utils.py:211-225);"
                        " 1 = the defaults (one full-buffer minibatch, whose mean needs no shuffle)")
    p.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                   help="nccl = RCCL over xGMI (production); gloo stages the tiny messages through the host"
                        " (rehearsals: CPU rendezvous tests, several ranks on one GPU)")
    p.add_argument("--single-device", action="store_true",
                   help="rehearsal: every rank uses cuda:0 (needs --backend gloo; RCCL wants one device per rank)")
    p.add_argument("--launch-timeout", type=float, default=None,
                   help="self-launched ranks (--gpus N without torchrun): seconds before the parent stops them all"
                        " (default: 900 + 120 per step and warm-up step)")
    p.add_argument("--towers", default="matrix", choices=["matrix", "piecewise"],
                   help="matrix (default, the product): the MFMA tower kernels.  piecewise: OPT-IN round-4 prototype for scalar"
                        " observations only (rl8_amd/nn/piecewise_mlp.py: each tower as an exact piecewise-linear table) -- a"
                        " labelled extra line, never the headline")
    p.add_argument("--uninstrumented-steps", type=int, default=10,
                   help="after the timed region: this many more steps with the per-kernel HIP-event timers OFF, reported as"
                        " value_uninstrumented / ms_per_step_uninstrumented beside the headline (0: skip)")
    p.add_argument("--cpu-baseline-seconds", type=float, default=120.0,
                   help="cap on the CPU baseline's wall time (3 sweep points + 10 timed iterations take ~65 s on the GPU hosts)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--secondary", dest="secondary", action="store_true", default=None,
                   help="after the headline (N = 1): BASELINE configs[2..4] on this one device, a few steps each, under the"
                        " line's `secondary` key (default: on when the run IS the headline configuration)")
    p.add_argument("--no-secondary", dest="secondary", action="store_false")
    p.add_argument("--secondary-steps", type=int, default=3)
    p.add_argument("--secondary-warmup", type=int, default=1)
    return p.parse_args(argv)


# --------------------------------------------------------------------------- #
# Self-launch: `python bench.py --gpus N` with no WORLD_SIZE in the environment
# --------------------------------------------------------------------------- #
def free_port() -> int:
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        return sock.getsockname()[1]


def rank_environment(rank: int, world: int, port: int, base: None | dict = None) -> dict:
    """What torchrun would export for local rank ``rank`` of a one-node job."""
    env = dict(os.environ if base is None else base)
    env.update({
        "RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_WORLD_SIZE": str(world),
        "GROUP_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port),
    })
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL needs it on these hosts
    return env


def default_launch_timeout(steps: int, warmup: int) -> float:
    """Finite by default: a rank hung in init_process_group or its first all-reduce must not hold N GPUs until an
    outer driver kills the parent. Generous against the slowest config (first import + build + RCCL bring-up, then
    <= 1 s per step on the headline config, ~10 s on CPU-rehearsal configs)."""
    return 900.0 + 120.0 * (steps + warmup)


def die_with_parent(parent_pid: int) -> None:
    """Called by a self-launched rank as its first action (``RL8_BENCH_PARENT`` in its environment): ask the kernel
    for SIGTERM when the launcher dies (Linux ``PR_SET_PDEATHSIG``), then check that it has not died already -- the
    request only covers deaths from now on.  Done here, in the child's own interpreter, not in a ``preexec_fn``: code
    between fork and exec of a threaded parent (its pump threads) is not safe to run."""
    try:
        import ctypes

        ctypes.CDLL("libc.so.6", use_errno=True).prctl(1, signal.SIGTERM)  # PR_SET_PDEATHSIG
    except Exception:  # noqa: BLE001  (not Linux / no libc: the launcher's signal handlers and its timeout remain)
        pass
    if os.getppid() != parent_pid:
        print(f"bench.py: launcher {parent_pid} is gone (parent is now {os.getppid()}): rank exits", file=sys.stderr, flush=True)
        os._exit(1)


def _stop_ranks(procs: list, grace: float = 20.0) -> None:
    """SIGTERM every live rank's process group, SIGKILL what is left after ``grace`` seconds."""
    live = [p for p in procs if p.poll() is None]
    for p in live:
        try:
            os.killpg(p.pid, signal.SIGTERM)  # (each rank leads its own session: the group is the rank + its helpers)
        except (ProcessLookupError, PermissionError):
            pass
    deadline = time.monotonic() + grace
    for p in live:
        try:
            p.wait(timeout=max(0.0, deadline - time.monotonic()))
        except subprocess.TimeoutExpired:
            try:
                os.killpg(p.pid, signal.SIGKILL)
            except (ProcessLookupError, PermissionError):
                pass
            p.wait()


def _tail(f, lines: int = 25) -> str:
    f.flush()
    f.seek(0)
    return "".join(f.readlines()[-lines:])


def launch_ranks(world: int, argv: list[str], *, script: str = os.path.abspath(__file__),
                 timeout: None | float = None) -> tuple[int, str]:
    """Start ``world`` fresh child processes of ``script`` (one per GPU), wait for
    them, return (exit code, rank 0's stdout). The caller has not touched a GPU
    and does not here: children are spawned (never exec'ed over this process) and
    each initialises its own device. Every rank leads its own session and asks the
    kernel for SIGTERM when this process dies; SIGTERM / SIGINT to this process stop
    the ranks (then exit 128 + signal). If a rank fails the others are stopped and
    the tail of its stderr (and rank 0's) is printed; the first non-zero exit code
    is returned (124 on timeout). Rank stderr is passed through line by line."""
    import tempfile
    import threading

    port = free_port()
    deadline = None if timeout is None else time.monotonic() + timeout
    procs: list = []
    logs = [tempfile.TemporaryFile("w+") for _ in range(world)]

    def pump(rank: int, stream) -> None:  # rank stderr -> ours (live progress) and -> its log (tail on failure)
        for line in stream:
            logs[rank].write(line)
            sys.stderr.write(line if world == 1 else f"[rank {rank}] {line}")
            sys.stderr.flush()

    def on_signal(signum, _frame):
        print(f"bench.py: signal {signum}: stopping {len(procs)} ranks", file=sys.stderr, flush=True)
        _stop_ranks(procs, grace=10.0)
        os._exit(128 + signum)

    previous = {sig: signal.signal(sig, on_signal) for sig in (signal.SIGTERM, signal.SIGINT)} \
        if threading.current_thread() is threading.main_thread() else {}
    with tempfile.TemporaryFile("w+") as rank0_out:
        pumps = []
        try:
            # every rank is forked before the first pump thread exists (fork from a single-threaded parent), leads its
            # own session (start_new_session: setsid in the child by the C runtime, no Python between fork and exec)
            # and arranges its own death with the launcher's (die_with_parent, first thing in main())
            for rank in range(world):
                env = rank_environment(rank, world, port)
                env["RL8_BENCH_PARENT"] = str(os.getpid())
                procs.append(subprocess.Popen([sys.executable, script, *argv], env=env,
                                              stdout=rank0_out if rank == 0 else subprocess.DEVNULL,
                                              stderr=subprocess.PIPE, text=True, start_new_session=True))
            for rank in range(world):
                pumps.append(threading.Thread(target=pump, args=(rank, procs[rank].stderr), daemon=True))
                pumps[-1].start()
            rc, pending, failed = 0, set(range(world)), None
            while pending and rc == 0:
                for rank in sorted(pending):
                    code = procs[rank].poll()
                    if code is None:
                        continue
                    pending.discard(rank)
                    if code != 0 and rc == 0:
                        rc, failed = code, rank
                        print(f"bench.py: rank {rank} exited with {code}; stopping the other ranks", file=sys.stderr)
                if pending and rc == 0:
                    if deadline is not None and time.monotonic() > deadline:
                        rc = 124
                        print(f"bench.py: ranks timed out after {timeout:.0f} s (--launch-timeout); still running:"
                              f" {sorted(pending)}", file=sys.stderr)
                    else:
                        time.sleep(0.05)
        finally:
            _stop_ranks(procs)
            for t in pumps:
                t.join(timeout=5)
            for sig, handler in previous.items():
                signal.signal(sig, handler)
        if rc != 0:
            for rank in sorted({r for r in (failed, 0) if r is not None}):
                print(f"bench.py: ---- last lines of rank {rank}'s stderr ----\n{_tail(logs[rank])}", file=sys.stderr, flush=True)
        for f in logs:
            f.close()
        rank0_out.seek(0)
        return rc, rank0_out.read()


def cpu_model() -> str:
    try:
        with open("/proc/cpuinfo") as f:
            return next(line.split(":", 1)[1].strip() for line in f if line.startswith("model name"))
    except (OSError, StopIteration):
        return "unknown"


def cpu_baseline(budget_s: float) -> dict:
    """The CPU restatement of the same algorithm (oracle/ppo_cpu.py: C kernels +
    torch-CPU MLP) on BASELINE configs[0] (DiscreteDummyEnv, N=8192, H=32,
    defaults), on the host cores of this box -- the best the host does, not one
    arbitrary setting: the 262 144 x 256 GEMMs of that config stop scaling (and
    then lose) well below the core count of a GPU host, so the torch thread count
    is swept (one timed collect()+step() each, after one warm-up iteration) and
    the rest of the budget is spent at the fastest. ``sweep`` keeps every point."""
    import torch

    from oracle.ppo_cpu import OraclePPO

    host_cpus = os.cpu_count() or 1
    usable = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else host_cpus
    candidates = sorted({t for t in (16, 32, 64) if t <= usable} | ({min(usable, 8)} if usable < 16 else set()))
    n = 8192 * 32
    torch.manual_seed(0)
    algo = OraclePPO("discrete", num_envs=8192, horizon=32)

    def iteration() -> float:
        t0 = time.perf_counter()
        algo.collect()
        algo.step()
        return time.perf_counter() - t0

    # warm-ups, then >= 10 timed collect()+step() iterations, median.  The warm-ups double as
    # the thread-count sweep (one iteration per candidate after a first untimed one); the ten timed iterations all run
    # at the fastest candidate.  ``budget_s`` caps the whole thing on a slow host (then fewer iterations, stated).
    t_start = time.perf_counter()
    sweep: dict[int, float] = {}
    order = [t for t in (32, 64, 16) if t in candidates] + [t for t in candidates if t not in (32, 64, 16)]
    torch.set_num_threads(order[0])
    iteration()  # thread pools, allocator, first touch
    for threads in order:
        torch.set_num_threads(threads)
        sweep[threads] = iteration()
    best = min(sweep, key=sweep.get)
    torch.set_num_threads(best)
    warmups = 1 + len(sweep)
    while warmups < 3:
        iteration()
        warmups += 1
    times: list[float] = []
    while len(times) < 10 and (len(times) < 3 or time.perf_counter() - t_start < budget_s):
        times.append(iteration())
    elapsed = sum(times)
    median = sorted(times)[len(times) // 2] if len(times) % 2 else 0.5 * (sorted(times)[len(times) // 2 - 1] + sorted(times)[len(times) // 2])
    return {
        "value": n / median,
        "unit": "env transitions/sec",
        "policy_updates_per_sec": 1.0 / median,
        "mean_value": n * len(times) / elapsed,
        "iterations": len(times),
        "warmup_iterations": warmups,
        "iteration_seconds": {"median": round(median, 4), "min": round(min(times), 4), "max": round(max(times), 4)},
        "cores": best,
        "threads": best,
        "host_cpus": host_cpus,
        "usable_cpus": usable,
        "cpu_model": cpu_model(),
        "sweep_transitions_per_sec": {str(t): round(n / v, 1) for t, v in sorted(sweep.items())},
        "kind": "port",
        "sample": f"median of {len(times)} x (collect+step) after {warmups} warm-ups at {best} torch threads (best of sweep"
                  f" {sorted(sweep)}), DiscreteDummyEnv num_envs=8192 horizon=32 defaults (BASELINE configs[0]),"
                  f" {elapsed:.1f} s timed, oracle C kernels + torch-CPU MLP",
    }


def main() -> None:
    if os.environ.get("RL8_BENCH_PARENT", "").isdigit():  # a rank of launch_ranks()
        die_with_parent(int(os.environ.pop("RL8_BENCH_PARENT")))
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: this process starts the ranks and stays off the GPU
        limit = args.launch_timeout if args.launch_timeout is not None else default_launch_timeout(args.steps, args.warmup)
        rc, out = launch_ranks(args.gpus, sys.argv[1:], timeout=limit)
        # stdout carries the ONE JSON line; anything else rank 0 wrote there (gloo's
        # connection banner) goes to stderr
        for text in out.splitlines():
            is_line = text.startswith("{") and text.rstrip().endswith("}")
            print(text, file=sys.stdout if is_line else sys.stderr, flush=True)
        sys.exit(rc)
    run(args)


def run(args: argparse.Namespace) -> None:
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    if args.single_device and args.backend != "gloo":
        raise SystemExit("bench.py: --single-device needs --backend gloo")
    device_index = 0 if args.single_device else local_rank
    backend = None
    visible = torch.cuda.device_count()  # (counting devices does not initialise one)
    if world > 1 and args.backend == "nccl" and visible < world:
        # fail before any rendezvous: RCCL wants one device per rank, and a rank without one would hang the others
        raise SystemExit(f"bench.py: rank {rank}: --gpus {world} with --backend nccl needs {world} HIP devices,"
                         f" {visible} visible (HIP_VISIBLE_DEVICES={os.environ.get('HIP_VISIBLE_DEVICES')},"
                         f" ROCR_VISIBLE_DEVICES={os.environ.get('ROCR_VISIBLE_DEVICES')})")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            torch.cuda.set_device(device_index)
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{device_index}"))
        else:
            dist.init_process_group("gloo")
        backend = dist.get_backend()
        world = dist.get_world_size()
        print(f"bench.py: rank {dist.get_rank()}/{world} up, backend {backend}", file=sys.stderr, flush=True)
    if torch.cuda.is_available():
        torch.cuda.set_device(device_index)
        try:
            rccl = ".".join(str(v) for v in torch.cuda.nccl.version()) if args.backend == "nccl" else "-"
        except Exception as exc:  # noqa: BLE001
            rccl = f"? ({exc.__class__.__name__})"
        print(f"bench.py: rank {rank}: device {device_index} of {visible} = {torch.cuda.get_device_name(device_index)},"
              f" RCCL {rccl}, torch {torch.__version__}, HSA_ENABLE_IPC_MODE_LEGACY="
              f"{os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY')}", file=sys.stderr, flush=True)
    # no HIP device / no librl8_amd.so: AlgorithmConfig.build() raises HipExtensionError below

    if args.secondary is None:  # default: with the headline configuration itself, on one device
        args.secondary = (world == 1 and args.env == "discrete" and args.distribution == "default" and not args.recurrent
                          and args.num_envs == 1 << 20 and args.horizon == 32 and args.minibatches == 1
                          and args.towers == "matrix")
    m = measure(args, world, backend)
    if rank == 0:
        line = headline(args, m, world, backend)
        if world == 1 and args.secondary:
            line["secondary"] = secondary_lines(args, world, backend)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.cpu_baseline_seconds)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def measure(args: argparse.Namespace, world: int, backend: None | str) -> dict:
    """Build the algorithm ``args`` names, run ``args.warmup`` untimed and exactly ``args.steps`` timed
    ``collect(); step()`` pairs between barriers (per-kernel HIP-event timers on), then ``args.uninstrumented_steps``
    more with the timers off; returns the times and the per-kernel table.  Nothing of the algorithm survives the call
    (buffers, records and caches are released), so several configurations can be measured in one process."""
    import gc

    import torch
    import torch.distributed as dist

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

    if args.towers == "piecewise":
        from rl8_amd.nn import piecewise_mlp

        piecewise_mlp.ENABLED = True
    torch.manual_seed(0)
    if args.scaling == "strong":
        if args.num_envs % world:
            raise SystemExit(f"bench.py: --scaling strong: --num-envs {args.num_envs} is not divisible by {world} ranks")
        global_envs = args.num_envs
    else:
        global_envs = args.num_envs * world
    envs_per_gpu = global_envs // world
    config_cls = RecurrentAlgorithmConfig if args.recurrent else AlgorithmConfig
    extra = {"distribution_cls": SquashedNormal} if args.distribution == "squashed" else {}
    if args.minibatches > 1:
        # the recurrent algorithm counts a minibatch in sequences (seq_len 4 by default), the feed-forward one in samples
        units = global_envs * args.horizon // (RecurrentAlgorithmConfig.seq_len if args.recurrent else 1)
        if units % args.minibatches:
            raise SystemExit(f"bench.py: --minibatches {args.minibatches} does not divide the {units} units of the buffer")
        extra["sgd_minibatch_size"] = units // args.minibatches
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
    # The same loop without the per-kernel HIP-event pairs (~127 timed ABI calls per step above), same process, same
    # barriers: what the instrumentation costs shows as the difference.
    plain_elapsed = None
    if args.uninstrumented_steps > 0:
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.uninstrumented_steps):
            algo.collect()
            algo.step()
        barrier()
        plain_elapsed = time.perf_counter() - t1
        if world > 1:
            t = torch.tensor([plain_elapsed], dtype=torch.float64, device="cpu" if backend == "gloo" else "cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            plain_elapsed = float(t)
    rank_elapsed = [elapsed]
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if backend == "gloo" else "cuda")
        every = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(every, t)   # (outside the timed region: a straggling rank shows in the line)
        rank_elapsed = [float(v) for v in every]
        elapsed = max(rank_elapsed)

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
        if name in ("mlp_tower_forward", "mlp_tower_forward_save", "mlp_tower_forward_record"):
            ok = f16 = fused_mlp.FORWARD_GEMM == "f16" and all(hip.mlp_forward_f16_supports(d, n) for d, n in widths)
        elif name == "mlp_wgrad_gate":  # rank-one heads: gate plane x the planes of dOut * h1 (two fp16, or three bf16)
            return "bf16-gate-x3" if os.environ.get("RL8_WGRAD_GATE_PLANES", "f16").startswith("b") else "f16-gatebits-x2"
        elif name == "mlp_tower_backward_gate":  # rank-one heads: gate plane x two planes of w3e * W2
            return "f16-gate-x2"
        elif name == "mlp_wgrad":  # general heads: both operands as planes (two fp16 each: three products; or three bf16: six)
            ok = fused_mlp.BACKWARD_GEMM == "f16"
            f16 = ok and not os.environ.get("RL8_WGRAD_PLANES", "f16").startswith("b") and all(
                hip.mlp_backward_f16_supports(d, n) for d, n in widths)
        else:
            ok = f16 = fused_mlp.BACKWARD_GEMM == "f16" and all(hip.mlp_backward_f16_supports(d, n) for d, n in widths)
        return "f16x2-split" if f16 else "bf16x3-split" if ok else "f32"

    lstm_gemm = {  # the recurrent models' LSTM (config 5): FLOP per row-step, matrix pipe
        "lstm_step": (2.0 * 256 * 1024, "f16x2-split"), "lstm_step_save": (2.0 * 256 * 1024, "f16x2-split"),
        "lstm_forward": (2.0 * 264 * 1024, "f32"), "lstm_forward_save": (2.0 * 264 * 1024, "f32"),
        "lstm_backward": (2.0 * 1024 * 256, "f32"),
        # backward through time: the recurrent product dG x W_hh of a row-step on two fp16 planes scaled per sequence
        # (three products; round 6, the heads form the default models run) or on three exact bf16 planes (six)
        "lstm_rows_backward": (2.0 * 1024 * 256,
                               "bf16x3-split" if os.environ.get("RL8_AMD_LSTM_BACKWARD_PLANES", "f16").startswith("b") else "f16x2-split"),
        # weight gradient: fp16 planes (three products) behind the rows kernel's bound on |dG|, bf16 planes (six) else
        "lstm_wgrad": (2.0 * 1024 * 256,
                       "f32" if os.environ.get("RL8_AMD_LSTM_GEMM", "split") != "split" else
                       "f16x2-split" if (os.environ.get("RL8_AMD_LSTM_BACKWARD_ROWS", "1") != "0"
                                         and os.environ.get("RL8_AMD_LSTM_WGRAD_PLANES", "f16") != "bf16") else "bf16x3-split"),
    }
    # ... and what the same kernels move per row-step: since the backward through time went to fp16
    # planes (round 6) it is the HBM roofline that binds it, not the matrix pipe
    lstm_bytes = {"lstm_rows_backward": 12.0 * 1024, "lstm_step_save": 9.0 * 1024, "lstm_step": 5.0 * 1024, "lstm_wgrad": 5.0 * 1024}
    for name, rec in hip.timer.summary().items():
        if name in lstm_gemm:
            per_row, gemm = lstm_gemm[name]
            flops_per_launch = per_row * rec["units_per_launch"]
            tflops = flops_per_launch / (rec["avg_ms"] * 1e-3) / 1e12
            kernels[name] = {
                "bound": "mfma", "gemm": gemm, "launches": rec["launches"], "avg_ms": round(rec["avg_ms"], 5),
                "total_ms": round(rec["total_ms"], 3), "algorithmic_flop_per_launch": flops_per_launch,
                "achieved_TFLOPs": round(tflops, 2), "frac_of_f32_mfma_peak": round(tflops / MFMA_F32_PEAK_TFLOPS, 4),
                "pmc_traffic_bytes_per_launch": fabric_traffic(name, rec["units_per_launch"]),
            }
            if name in lstm_bytes:
                moved = lstm_bytes[name] * rec["units_per_launch"]
                gbs = moved / (rec["avg_ms"] * 1e-3) / 1e9 if rec["avg_ms"] > 0 else 0.0
                kernels[name].update({"algorithmic_bytes_per_launch": moved, "achieved_GBps": round(gbs, 1),
                                      "frac_of_8TBps": round(gbs / HBM_PEAK_GBS, 4)})
            if gemm != "f32":
                executed = PLANE_PRODUCTS[gemm] * flops_per_launch
                kernels[name].update({
                    "plane_products": PLANE_PRODUCTS[gemm],
                    "executed_bf16_flop_per_launch": executed,
                    "executed_bf16_TFLOPs": round(executed / (rec["avg_ms"] * 1e-3) / 1e12, 1),
                    "frac_of_bf16_mfma_peak": round(executed / (rec["avg_ms"] * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4),
                })
            continue
        if name.startswith("mlp_"):
            # n_out differs per tower (policy 2-3, value 1); price both at the mean.
            # mlp_wgrad is the 256x256 weight-gradient product alone.
            per_row = 2.0 * 256 * 256 if name in ("mlp_wgrad", "mlp_wgrad_gate") else tower_flops_per_row(obs_dim, 1.5)
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
                "pmc_traffic_bytes_per_launch": pmc_traffic(name, rec["units_per_launch"], gemm),
            }
            if gemm != "f32":
                # every fp32 multiply-add of the 256x256 product is 6 bf16 (3 fp16) ones on the matrix pipe
                executed = PLANE_PRODUCTS[gemm] * 2.0 * 256 * 256 * rec["units_per_launch"]
                bf16_tflops = executed / (rec["avg_ms"] * 1e-3) / 1e12
                kernels[name].update({
                    "plane_products": PLANE_PRODUCTS[gemm],
                    "executed_bf16_flop_per_launch": executed,   # (16-bit MFMA flop: bf16 or fp16 planes, same peak)
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
            "pmc_traffic_bytes_per_launch": (pmc_traffic(name, rec["units_per_launch"]) if name not in FABRIC_KERNEL
                                             else fabric_traffic(name, rec["units_per_launch"])),
        }

    out = {
        "kernels": kernels, "elapsed": elapsed, "plain_elapsed": plain_elapsed, "rank_elapsed": rank_elapsed,
        "collect_ms": collect_ms, "step_ms": step_ms, "global_envs": global_envs, "envs_per_gpu": envs_per_gpu,
        "horizon": horizon, "env_name": env_cls.__name__, "num_sgd_iters": algo.hparams.num_sgd_iters,
        "num_minibatches": algo.hparams.num_minibatches, "collectives": algo.shards.collectives,
        "piecewise": dict(piecewise_mlp.stats) if args.towers == "piecewise" else None,
        "max_memory_allocated_GB": round(torch.cuda.max_memory_allocated() / 1e9, 2),
    }
    # release everything the configuration held on the device (the next one may need most of it)
    algo._release_step_caches()
    del algo, model
    gc.collect()
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats()
    return out


def roofline_of(kernels: dict, steps: int) -> tuple[None | dict, str, str]:
    """The ``roofline`` object of a bench line: the kernel the timed region spent most time in (None when that kernel
    is HBM-bound: the caller then prices the fused loss kernel), its bench name, and the name of the HBM kernel."""
    hbm_dominant = "ppo_loss_categorical" if "ppo_loss_categorical" in kernels else "ppo_loss_normal"
    dominant = max(kernels, key=lambda k: kernels[k]["total_ms"])
    top = kernels[dominant]
    if top["bound"] == "mfma" and top.get("frac_of_8TBps", 0.0) > top.get("frac_of_bf16_mfma_peak", 1.0):
        # a plane kernel that sits closer to the HBM roofline than to the matrix pipe's (the LSTM's backward through time
        # on fp16 planes: 12 KiB per row-step at 4.5+ TB/s against 0.24 executed of the 16-bit MFMA peak)
        roofline = {
            "kernel": {"lstm_rows_backward": "rl8_lstm_rows_backward_heads_f32 (backward through time of the recurrent models' LSTM)",
                       "lstm_wgrad": "rl8_lstm_wgrad_f16_f32", "lstm_step_save": "rl8_lstm_step_split_f32"}.get(dominant, f"rl8_{dominant}_f32"),
            "bound": "hbm",
            "achieved": top["achieved_GBps"],
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": top["frac_of_8TBps"],
            "algorithmic_bytes_per_launch": top["algorithmic_bytes_per_launch"],
            "executed_frac": top.get("frac_of_bf16_mfma_peak"),   # (of the 16-bit MFMA peak, for comparison)
            "plane_products": top.get("plane_products"),
            "avg_launch_ms": top["avg_ms"],
            "launches": top["launches"],
            "share_of_step_ms": round(top["total_ms"] / steps, 2),
            "traffic": top["pmc_traffic_bytes_per_launch"],
        }
    elif top["bound"] == "mfma" and top["gemm"] != "f32":
        # bf16-plane kernel: priced in the bf16 multiply-adds the matrix pipe executes
        # (6 per fp32 multiply-add of the algorithm) against the dense bf16 peak
        roofline = {
            "kernel": {"mlp_wgrad": "rl8_mlp_wgrad_fused_split_f32", "mlp_wgrad_gate": "rl8_mlp_wgrad_gate_bits_f32 (rank-one heads: one output, or a pair of opposite gradients)",
                       "mlp_tower_backward_gate": "rl8_mlp_tower_backward_gate_f16_f32",
                       "lstm_rows_backward": "rl8_lstm_rows_backward_heads_f32 (backward through time of the recurrent models' LSTM)",
                       "lstm_wgrad": "rl8_lstm_wgrad_f16_f32", "lstm_step_save": "rl8_lstm_step_split_f32"}.get(
                dominant, f"rl8_{dominant}_{'f16' if top['gemm'] == 'f16x2-split' else 'split'}_f32"),
            "bound": "mfma",
            # ALGORITHMIC FLOP of the launch (one fp32 product per multiply-add of the reference's
            # arithmetic) / HIP-event time, against the dense 16-bit MFMA peak the kernel's products run on
            "achieved": top["achieved_TFLOPs"],
            "peak": MFMA_BF16_PEAK_TFLOPS,
            "unit": "TFLOP/s",
            "frac": round(top["achieved_TFLOPs"] / MFMA_BF16_PEAK_TFLOPS, 4),
            # what the matrix pipe EXECUTES: plane_products 16-bit products per fp32 product (fp32 accuracy on a
            # 16-bit pipe), against the same peak -- the ceiling of this scheme is 1 / plane_products
            "executed_TFLOPs": top["executed_bf16_TFLOPs"],
            "executed_frac": top["frac_of_bf16_mfma_peak"],
            "plane_products": top["plane_products"],
            "executed_flop_per_launch": top["executed_bf16_flop_per_launch"],
            "flop_definition": (f"{top['plane_products']} 16-bit plane products x 2*1024*256 per row-step (the recurrent product of"
                                " the LSTM: fp32 operands as 3 exact bf16 planes / 2 scaled fp16 planes, fp32 accumulate)"
                                if dominant.startswith("lstm_") else
                                "3 bf16 plane products x 2*256*256 per row (ReLU gate as one exact bf16 plane x the three"
                                " planes of dOut*h1, fp32 accumulate)" if top["gemm"] == "bf16-gate-x3" else
                                "2 fp16 plane products x 2*256*256 per row (ReLU gate as one exact fp16 plane x the two"
                                " planes of dOut*h1 scaled per column, fp32 accumulate)" if top["gemm"] == "f16-gatebits-x2" else
                                "2 fp16 plane products x 2*256*256 per row (ReLU gate as one exact fp16 plane x the two"
                                " planes of w3e*W2, fp32 accumulate)" if top["gemm"] == "f16-gate-x2" else
                                "3 fp16 plane products x 2*256*256 per row (fp32 operands scaled by powers of two and"
                                " split into 2 fp16 planes, fp32 accumulate)" if top["gemm"] == "f16x2-split" else
                                "6 bf16 plane products x 2*256*256 per row (fp32 operands split exactly into"
                                " 3 bf16 planes, fp32 accumulate)"),
            "f32_equivalent_TFLOPs": top["achieved_TFLOPs"],   # the algorithm's fp32 FLOP / time
            "f32_mfma_peak_TFLOPs": MFMA_F32_PEAK_TFLOPS,
            "f32_equivalent_frac_of_f32_mfma_peak": top["frac_of_f32_mfma_peak"],  # > 1: beyond the fp32 matrix roofline
            "sustained_TFLOPs_measured": MFMA_16BIT_SUSTAINED_TFLOPS,  # tools/probes/mfma_shape_probe.hip (power-limited)
            "frac_of_sustained": round(top["executed_bf16_TFLOPs"] / MFMA_16BIT_SUSTAINED_TFLOPS, 4),
            "algorithmic_flop_per_launch": top["algorithmic_flop_per_launch"],
            "avg_launch_ms": top["avg_ms"],
            "launches": top["launches"],
            "share_of_step_ms": round(top["total_ms"] / steps, 2),
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
            "share_of_step_ms": round(top["total_ms"] / steps, 2),
            "traffic": top["pmc_traffic_bytes_per_launch"],
        }
    else:
        roofline = None
    return roofline, dominant, hbm_dominant


def hbm_roofline(kernels: dict, hbm_dominant: str) -> dict:
    dom = kernels[hbm_dominant]
    return {
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
    }


def workload_text(args: argparse.Namespace, m: dict) -> str:
    variant = ("Recurrent" if args.recurrent else "") + (" SquashedNormal" if args.distribution == "squashed" else "")
    return (f"{m['env_name']}{variant} collect()+step(), num_envs={m['envs_per_gpu']} per GPU"
            f" ({m['global_envs']} total), horizon={m['horizon']}, AlgorithmConfig defaults"
            + (" (4 SGD iters, one full-buffer minibatch, Adam 1e-3)" if args.minibatches == 1 else
               f" except sgd_minibatch_size = buffer / {args.minibatches} (4 SGD iters x {args.minibatches}"
               " shuffled minibatches, Adam 1e-3)"))


def headline(args: argparse.Namespace, m: dict, world: int, backend: None | str) -> dict:
    """The one JSON line of the contract from a :func:`measure` result."""
    kernels, elapsed, plain_elapsed, rank_elapsed = m["kernels"], m["elapsed"], m["plain_elapsed"], m["rank_elapsed"]
    global_envs, horizon = m["global_envs"], m["horizon"]
    roofline, _, hbm_dominant = roofline_of(kernels, args.steps)
    hbm = hbm_roofline(kernels, hbm_dominant)
    transitions = global_envs * horizon * args.steps
    line = {
        "metric": "env transitions/sec + policy updates/sec, DiscreteDummyEnv num_envs=2^20 h=32",
        "value": transitions / elapsed,
        "unit": "env transitions/sec",
        "policy_updates_per_sec": args.steps / elapsed,
        "optimizer_steps_per_sec": args.steps * m["num_sgd_iters"] * m["num_minibatches"] / elapsed,
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "value_uninstrumented": (global_envs * horizon * args.uninstrumented_steps / plain_elapsed
                                 if plain_elapsed else None),
        "ms_per_step_uninstrumented": plain_elapsed / args.uninstrumented_steps * 1e3 if plain_elapsed else None,
        "uninstrumented_steps": args.uninstrumented_steps if plain_elapsed else 0,
        "rank_ms_per_step": {"min": min(rank_elapsed) / args.steps * 1e3, "max": max(rank_elapsed) / args.steps * 1e3,
                             "per_rank": [round(v / args.steps * 1e3, 3) for v in rank_elapsed]},
        "collect_ms_per_step": m["collect_ms"] / args.steps,
        "update_ms_per_step": m["step_ms"] / args.steps,
        "higher_is_better": True,
        "scaling": args.scaling,
        "world_size": world,
        "backend": backend,
        "collectives_per_step": m["collectives"] / max(args.steps + args.warmup
                                                       + (args.uninstrumented_steps if plain_elapsed else 0), 1),
        "vs_baseline": None,
        "dtype": "f32",
        "gemm": sorted({k["gemm"] for k in kernels.values() if k["bound"] == "mfma"}),
        "data": "synthetic (Philox-reset DiscreteDummyEnv states, random-init default MLP)",
        "config": {
            "workload": workload_text(args, m),
            "num_envs_per_gpu": m["envs_per_gpu"],
            "num_envs_global": global_envs,
            "horizon": horizon,
            "parallelism": f"env-sharded x{world}",
            # north_star quotes ONE problem (num_envs = 2^20 in total) at 1/2/4/8 GPUs: that table is
            # `--scaling strong --num-envs 1048576 --gpus N`; this line's convention is `scaling` above
            "scaling_conventions": {
                "this_line": {"scaling": args.scaling, "num_envs_global": global_envs, "num_envs_per_gpu": m["envs_per_gpu"]},
                "weak": {"num_envs_global": args.num_envs * world, "num_envs_per_gpu": args.num_envs},
                "strong": ({"num_envs_global": args.num_envs, "num_envs_per_gpu": args.num_envs // world}
                           if args.num_envs % world == 0 else None),
                "north_star_table": "--scaling strong --num-envs 1048576 (2^20 environments in total at every N)",
            },
        },
        "roofline": roofline if roofline is not None else hbm,
        "roofline_hbm": {k: v for k, v in hbm.items() if k != "frac_of_measured_copy_ceiling"},
        "kernels": kernels,
        "kernels_note": "HIP-event times of ABI calls inside the timed region; the recurrent rollout's per-timestep"
                        " launches are sampled (every 16th timestep), so their `launches` / `total_ms` are of the sample",
        "hand_kernel_ms_per_step": round(sum(k["total_ms"] for k in kernels.values()) / args.steps, 3),
        "max_memory_allocated_GB": m["max_memory_allocated_GB"],
        "fused_towers": True,
        "towers": ("matrix kernels (the product path)" if args.towers == "matrix" else
                   "OPT-IN PROTOTYPE: exact piecewise-linear tables of a scalar observation (rl8_amd/nn/piecewise_mlp.py);"
                   " table build and gradient algebra still torch fp64 ops; not the headline"),
    }
    if m["piecewise"] is not None:
        line["piecewise"] = m["piecewise"]
    return line


#: BASELINE.json configs[2..4] on ONE device, measured after the headline in the same process (VERDICT r5 next #1b):
#: (key, what BASELINE calls it, argument overrides).  configs[3] / [4] are quoted on 8 GPUs: here the whole stated
#: problem runs on one.
SECONDARY = (
    ("cfg3", "configs[2]: CartPole, num_envs=2^18, horizon=128",
     {"env": "cartpole", "num_envs": 1 << 18, "horizon": 128}),
    ("cfg4", "configs[3]: ContinuousDummyEnv + SquashedNormal, num_envs=2^20, horizon=32 (whole problem on one device)",
     {"env": "continuous", "distribution": "squashed", "num_envs": 1 << 20, "horizon": 32}),
    ("cfg5", "configs[4]: RecurrentAlgorithmConfig on DiscreteDummyEnv, num_envs=2^16, horizon=256 (whole problem on one device)",
     {"env": "discrete", "recurrent": True, "num_envs": 1 << 16, "horizon": 256}),
)


def secondary_lines(args: argparse.Namespace, world: int, backend: None | str) -> dict:
    """The other single-device configurations of BASELINE.json, a few steps each, same timing rules as the headline
    (warm-up, barrier + synchronize on both sides, HIP-event timers on, then the same steps with the timers off)."""
    out: dict = {"note": "same process, after the headline's timed region; each: --secondary-warmup untimed +"
                         " --secondary-steps timed collect()+step() with the kernel timers on, then as many without"}
    for key, title, overrides in SECONDARY:
        a = argparse.Namespace(**{**vars(args), "env": "discrete", "distribution": "default", "recurrent": False,
                                  "minibatches": 1, "towers": "matrix", "scaling": "weak",
                                  "steps": args.secondary_steps, "warmup": args.secondary_warmup,
                                  "uninstrumented_steps": args.secondary_steps, **overrides})
        t0 = time.perf_counter()
        try:
            m = measure(a, world, backend)
        except Exception as exc:  # noqa: BLE001  (the headline must still be printed)
            out[key] = {"config": title, "error": f"{exc.__class__.__name__}: {exc}"}
            print(f"bench.py: secondary {key} failed: {exc!r}", file=sys.stderr, flush=True)
            continue
        kernels = m["kernels"]
        roofline, dominant, hbm_dominant = roofline_of(kernels, a.steps)
        if roofline is None:
            roofline = hbm_roofline(kernels, hbm_dominant)
        transitions = m["global_envs"] * m["horizon"]
        top = sorted(kernels, key=lambda k: -kernels[k]["total_ms"])[:6]
        out[key] = {
            "config": title,
            "workload": workload_text(a, m),
            "value": transitions * a.steps / m["elapsed"],
            "unit": "env transitions/sec",
            "ms_per_step": m["elapsed"] / a.steps * 1e3,
            "value_uninstrumented": transitions * a.uninstrumented_steps / m["plain_elapsed"] if m["plain_elapsed"] else None,
            "collect_ms_per_step": m["collect_ms"] / a.steps,
            "update_ms_per_step": m["step_ms"] / a.steps,
            "steps": a.steps, "warmup": a.warmup,
            "dominant_kernel": dominant,
            "frac": roofline["frac"],
            "executed_frac": roofline.get("executed_frac"),
            "roofline": {k: roofline.get(k) for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "executed_frac",
                                                      "plane_products", "avg_launch_ms", "launches", "share_of_step_ms",
                                                      "traffic")},
            "kernels_ms_per_step": {k: {"ms_per_step": round(kernels[k]["total_ms"] / a.steps, 3),
                                        "avg_ms": kernels[k]["avg_ms"], "launches": kernels[k]["launches"],
                                        "frac": kernels[k].get("frac_of_bf16_mfma_peak", kernels[k].get("frac_of_8TBps"))}
                                    for k in top},
            "max_memory_allocated_GB": m["max_memory_allocated_GB"],
            "wall_s": round(time.perf_counter() - t0, 1),
        }
        print(f"bench.py: secondary {key}: {out[key]['value'] / 1e6:.1f} M/s, {out[key]['ms_per_step']:.1f} ms/step,"
              f" {out[key]['wall_s']} s", file=sys.stderr, flush=True)
    return out


if __name__ == "__main__":
    main()
