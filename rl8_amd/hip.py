"""ctypes binding of ``librl8_amd.so`` (the gfx950 kernels behind
``include/rl8_amd.h``) for PyTorch-ROCm tensors.

PyTorch is plumbing here: it owns device memory and the stream. Every wrapper
passes ``tensor.data_ptr()`` and the current HIP stream to the C ABI, checks the
integer status, and returns. Nothing synchronises.

There is NO CPU fallback: if the shared library is missing, or a tensor is not
on a HIP device, the call raises.

"""

from __future__ import annotations

import ctypes as C
import os
from typing import Any, Sequence

import torch

_LIB_NAME = "librl8_amd.so"
_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), _LIB_NAME)
# Kernel-tuning runs point this at an experimental build of the SAME ABI
# (tools/diag_mlp.sh); there is still no fallback if the file is missing.
_LIB_PATH = os.environ.get("RL8_AMD_LIBRARY", _LIB_PATH)
_lib: None | C.CDLL = None

_ERRORS = {
    -1: "required pointer is NULL",
    -2: "size / shape argument out of range",
    -3: "pointer not aligned as required",
    -4: "unsupported combination of options",
}

MAX_CLASSES = 64
MAX_GATHER_FIELDS = 8
LAYOUT_ENV_MAJOR = 0
LAYOUT_TIME_MAJOR = 1


class HipExtensionError(RuntimeError):
    """The HIP extension is missing or a kernel launch failed."""


class CartPoleCfg(C.Structure):
    _fields_ = [
        ("force_mag", C.c_float),
        ("gravity", C.c_float),
        ("length", C.c_float),
        ("pole_mass", C.c_float),
        ("pole_mass_length", C.c_float),
        ("total_mass", C.c_float),
        ("tau", C.c_float),
        ("semi_implicit", C.c_int32),
    ]


class MountainCarCfg(C.Structure):
    """``rl8_mountain_car_cfg`` (include/rl8_amd.h)."""

    _fields_ = [(name, C.c_float) for name in ("force_mag", "goal_position", "goal_velocity", "gravity",
                                               "max_position", "max_speed", "min_position")]


class PendulumCfg(C.Structure):
    """``rl8_pendulum_cfg`` (include/rl8_amd.h)."""

    _fields_ = [(name, C.c_float) for name in ("dt", "gravity_coeff", "torque_coeff", "max_speed", "max_torque")]



class PPOHparams(C.Structure):
    _fields_ = [
        ("clip_param", C.c_float),
        ("dual_clip_param", C.c_float),
        ("entropy_coeff", C.c_float),
        ("vf_clip_param", C.c_float),
        ("vf_coeff", C.c_float),
        ("grad_scale", C.c_float),
    ]


class GatherField(C.Structure):
    _fields_ = [
        ("src", C.c_void_p),
        ("dst", C.c_void_p),
        ("env_stride", C.c_int64),
        ("time_stride", C.c_int64),
        ("row_elems", C.c_int32),
        ("elem_bytes", C.c_int32),
    ]


_vp, _i64, _u64, _i32, _f32 = C.c_void_p, C.c_int64, C.c_uint64, C.c_int, C.c_float

# name -> argtypes, mirroring include/rl8_amd.h one to one.
SIGNATURES: dict[str, list[Any]] = {
    "rl8_abi_version": [C.c_char_p, _i32],
    "rl8_pw_max_breaks": [],
    "rl8_pw_workspace_bytes": [_i32, _i32],
    "rl8_pw_tower_forward_f32": [_vp, _i64, _vp, _i32, _i32, _vp, _vp],
    "rl8_pw_segment_sums_f32": [_vp, _vp, _i64, _i32, _vp, _i32, _vp, _vp, _vp],
    "rl8_scratch_bytes": [],
    "rl8_dummy_env_step_f32": [_vp, _vp, _i32, _vp, _i64, _vp],
    "rl8_dummy_env_reset_f32": [_vp, _i64, _f32, _u64, _u64, _i64, _vp],
    "rl8_cartpole_step_f32": [_vp, _vp, C.POINTER(CartPoleCfg), _vp, _i64, _vp, _i64, _vp],
    "rl8_cartpole_reset_f32": [_vp, _i64, _f32, _u64, _u64, _i64, _vp, _i64, _vp],
    "rl8_categorical_sample_logp_f32": [_vp, _vp, _vp, _vp, _i64, _i32, _i32, _u64, _u64, _i64, _i32, _vp],
    "rl8_normal_sample_logp_f32": [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _u64, _u64, _i64, _i32, _vp],
    "rl8_rollout_scatter_f32": [_vp, _i64, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f32, _i64, _vp],
    "rl8_rollout_step_dummy_f32": [_i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f32, _i64, _u64, _u64, _i64, _i32, _vp],
    "rl8_rollout_step_dummy_heads_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f32, _i64, _u64,
                                         _u64, _i64, _i32, _vp],
    "rl8_mountain_car_step_f32": [_vp, _vp, C.POINTER(MountainCarCfg), _vp, _i64, _vp, _i64, _vp],
    "rl8_mountain_car_reset_f32": [_vp, _i64, _u64, _u64, _i64, _vp, _i64, _vp],
    "rl8_rollout_step_mountain_car_f32": [_vp, _vp, _vp, _vp, C.POINTER(MountainCarCfg), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f32, _i64, _u64, _u64, _i64, _i32, _vp],
    "rl8_pendulum_step_f32": [_vp, _vp, C.POINTER(PendulumCfg), _vp, _i64, _vp, _i64, _vp],
    "rl8_pendulum_reset_f32": [_vp, _i64, _u64, _u64, _i64, _vp, _i64, _vp],
    "rl8_rollout_step_pendulum_f32": [_i32, _vp, _vp, _vp, _vp, _vp, C.POINTER(PendulumCfg), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f32, _i64, _u64, _u64, _i64, _i32, _vp],
    "rl8_rollout_step_cartpole_f32": [_vp, _vp, _vp, _vp, C.POINTER(CartPoleCfg), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f32, _i64, _u64, _u64, _i64, _i32, _vp],
    "rl8_rollout_stats_f32": [_vp, _vp, _i64, _i64, _i64, _i64, _vp, _vp, _vp],
    "rl8_gae_scan_f32": [_vp, _vp, _vp, _vp, _i64, _i64, _i32, _f32, _f32, _f32, _i32, _vp, _vp, _vp],
    "rl8_advantage_normalise_f32": [_vp, _i64, _i64, _i32, _vp, _vp],
    "rl8_ppo_loss_categorical_fwd_bwd_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, C.POINTER(PPOHparams), _vp, _vp, _vp, _vp, _vp],
    "rl8_ppo_loss_normal_fwd_bwd_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, C.POINTER(PPOHparams), _vp, _vp, _vp, _vp, _vp, _vp],
    "rl8_gather_minibatch": [_vp, _i64, _i64, C.POINTER(GatherField), _i32, _vp],
    "rl8_pack_samples": [_vp, _i32, _i64, _i64, _vp, _i32, _vp],
    "rl8_gather_packed": [_vp, _i64, _vp, _i32, _vp, _i32, _vp],
    "rl8_lstm_supports": [_i32],
    "rl8_lstm_pack_floats": [],
    "rl8_lstm_pack_f32": [_vp, _vp, _vp, _vp, _i32, _vp, _vp],
    "rl8_lstm_forward_f32": [_vp, _i64, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "rl8_lstm_split_supports": [_i32],
    "rl8_lstm_split_packed_bytes": [],
    "rl8_lstm_split_wb_floats": [],
    "rl8_lstm_split_state_bytes": [_i64],
    "rl8_lstm_pack_split": [_vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp],
    "rl8_lstm_split_state": [_vp, _i64, _i64, _vp, _vp],
    "rl8_lstm_split_state_bound": [_vp, _i64, _i64, _vp, _vp, _vp],
    "rl8_lstm_step_split_f32": [_vp, _i64, _i32, _vp, _vp, _i64, _vp, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp],
    "rl8_lstm_rows_backward_pack_bytes": [],
    "rl8_lstm_rows_backward_pack": [_vp, _vp, _vp],
    "rl8_lstm_rows_backward_f32": [_i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "rl8_lstm_rows_backward_heads_f32": [_i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "rl8_mlp_wgrad_f16_strided_f32": [_vp, _i64, _vp, _vp, _i64, _vp, _i64, _vp, _vp, _i32, _vp, _i32, _vp, C.POINTER(C.c_int), _vp],
    "rl8_lstm_wgrad_f16_f32": [_vp, _i64, _vp, _vp, _i64, _vp, _i64, _vp, _vp, _i32, _vp, _i32, _vp, C.POINTER(C.c_int), _vp],
    "rl8_lstm_backward_partial_floats": [_i32],
    "rl8_lstm_backward_max_rows": [],
    "rl8_lstm_backward_f32": [_vp, _i64, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.POINTER(C.c_int), _vp],
    "rl8_linear_heads_max_rows": [],
    "rl8_linear_heads_forward_f32": [_vp, _i64, _vp, _vp, _i32, _vp, _vp],
    "rl8_linear_heads_forward_pair_f32": [_vp, _i64, _vp, _vp, _i32, _vp, _vp, _vp, _i32, _vp, _vp],
    "rl8_linear_heads_backward_f32": [_vp, _vp, _i64, _vp, _i32, _vp, _vp, C.POINTER(C.c_int), _vp],
    "rl8_mlp_wgrad_strided_f32": [_vp, _i64, _vp, _i64, _i64, _vp, _vp, _i32, _vp],
    "rl8_mlp_wgrad_split_strided_f32": [_vp, _i64, _vp, _i64, _i64, _vp, _vp, _i32, _vp, _i32, _vp, C.POINTER(C.c_int), _vp],
    "rl8_mlp_pack_w2_f32": [_vp, _vp, _i32, _vp],
    "rl8_mlp_tower_forward_f32": [_vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp],
    "rl8_mlp_backward_partial_floats": [_i32, _i32],
    "rl8_mlp_backward_max_rows": [],
    "rl8_mlp_tower_backward_f32": [_vp, _vp, _vp, _vp, _i64, _i32, _vp, _vp, _i32, _vp, _vp, C.POINTER(C.c_int), _vp],
    "rl8_mlp_backward_f16_supports": [_i32, _i32],
    "rl8_mlp_tower_backward_f16_f32": [_vp, _vp, _vp, _vp, _i64, _i32, _vp, _vp, _i32, _vp, C.POINTER(C.c_int), _vp, _vp],
    "rl8_mlp_dout_pair_check": [_vp, _i64, _vp, _vp],
    "rl8_mlp_wgrad_fused_pair_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp],
    "rl8_mlp_pack_w2_f16_gate": [_vp, _vp, _i32, _vp, _vp],
    "rl8_mlp_tower_backward_gate_f16_f32": [_vp, _vp, _vp, _vp, _i64, _i32, _vp, _i32, _vp, C.POINTER(C.c_int), _vp, _vp],
    "rl8_mlp_wgrad_gate_bits_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _vp, _vp, _vp, _vp],
    "rl8_mlp_f16_packed_bytes": [],
    "rl8_mlp_forward_f16_supports": [_i32, _i32],
    "rl8_mlp_pack_w2_f16": [_vp, _i32, _vp, _vp],
    "rl8_mlp_tower_forward_f16_f32": [_vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp],
    "rl8_mlp_wgrad_split_f32": [_vp, _vp, _vp, _vp, _i64, _i32, _vp, _vp, _i32, _vp],
    "rl8_mlp_wgrad_fused_split_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _vp, _vp, _vp, _vp],
    "rl8_mlp_wgrad_workspace_bytes": [],
    "rl8_mlp_wgrad_f32": [_vp, _vp, _i64, _vp, _vp, _i32, _vp],
}


#: RL8_ABI_VERSION of include/rl8_amd.h this binding is written against (checked against the library in ``load``).
ABI_VERSION = 106


def library_path() -> str:
    return _LIB_PATH


def load() -> C.CDLL:
    """Load the extension (after torch, so that the HIP runtime torch already
    mapped -- soname ``libamdhip64.so.7`` -- is the one the kernels launch on)."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise HipExtensionError(
                f"{_LIB_PATH} is missing: build it with `python -c 'import"
                " __graft_entry__ as g; g.build()'` (or `make -C rl8_amd/csrc`)."
                " rl8_amd has no CPU fallback."
            )
        lib = C.CDLL(_LIB_PATH)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.argtypes = argtypes
            fn.restype = (
                C.c_int64
                if name in ("rl8_scratch_bytes", "rl8_mlp_backward_partial_floats", "rl8_mlp_wgrad_workspace_bytes",
                            "rl8_lstm_pack_floats", "rl8_lstm_backward_partial_floats",
                            "rl8_lstm_split_packed_bytes", "rl8_lstm_split_wb_floats", "rl8_lstm_split_state_bytes",
                            "rl8_mlp_f16_packed_bytes", "rl8_lstm_rows_backward_pack_bytes", "rl8_pw_workspace_bytes")
                else C.c_int
            )
        built = int(lib.rl8_abi_version(None, 0))
        if built != ABI_VERSION:
            raise HipExtensionError(
                f"{_LIB_PATH} was built with ABI version {built}, this binding is written against {ABI_VERSION}"
                " (include/rl8_amd.h: RL8_ABI_VERSION): rebuild with `make -C rl8_amd/csrc`."
            )
        _lib = lib
    return _lib


def _check(status: int, name: str) -> None:
    if status == 0:
        return
    if status < 0:
        raise ValueError(f"{name}: {_ERRORS.get(status, 'argument check failed')} ({status})")
    raise HipExtensionError(f"{name}: HIP launch failed with hipError_t {status}")


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


class KernelTimer:
    """Optional HIP-event timing of launches, per ABI entry point. Events are
    recorded on the stream the kernel is launched on (torch's current stream);
    nothing synchronises until :meth:`summary` is called. Used by ``bench.py``
    for the roofline figures; off by default."""

    def __init__(self) -> None:
        self.enabled = False
        self.records: dict[str, list[tuple[torch.cuda.Event, torch.cuda.Event, float]]] = {}

    def reset(self) -> None:
        self.records = {}

    def summary(self) -> dict[str, dict[str, float]]:
        torch.cuda.synchronize()
        out = {}
        for name, events in self.records.items():
            ms = [a.elapsed_time(b) for a, b, _ in events]
            units = [u for _, _, u in events]
            out[name] = {
                "launches": len(ms),
                "total_ms": sum(ms),
                "avg_ms": sum(ms) / len(ms),
                "units_per_launch": sum(units) / len(units),
            }
        return out


timer = KernelTimer()


class _timed:
    __slots__ = ("name", "units", "start")

    def __init__(self, name: str, units: float) -> None:
        self.name, self.units = name, units

    def __enter__(self) -> None:
        if timer.enabled:
            self.start = torch.cuda.Event(enable_timing=True)
            self.start.record()

    def __exit__(self, *exc: Any) -> None:
        if timer.enabled:
            end = torch.cuda.Event(enable_timing=True)
            end.record()
            timer.records.setdefault(self.name, []).append((self.start, end, self.units))


def _ptr(t: None | torch.Tensor) -> None | int:
    if t is None:
        return None
    if not t.is_cuda:
        raise HipExtensionError(
            "rl8_amd kernels take HIP device tensors only (got a"
            f" {t.device} tensor); there is no CPU fallback."
        )
    return t.data_ptr()


def _dense(t: torch.Tensor, dtype: torch.dtype, what: str) -> torch.Tensor:
    if t.dtype != dtype:
        raise TypeError(f"{what} must be {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{what} must be contiguous")
    return t


_scratch: dict[tuple[int, int], torch.Tensor] = {}


def scratch(device: torch.device) -> torch.Tensor:
    """Per (device, stream) reduction scratch, allocated once."""
    key = (device.index if device.index is not None else torch.cuda.current_device(), _stream())
    buf = _scratch.get(key)
    if buf is None:
        nbytes = int(load().rl8_scratch_bytes())
        buf = torch.zeros(nbytes // 8, dtype=torch.float64, device=device)
        _scratch[key] = buf
    return buf


def abi_version() -> tuple[int, str]:
    buf = C.create_string_buffer(16)
    v = load().rl8_abi_version(buf, 16)
    return int(v), buf.value.decode()


# --------------------------------------------------------------------------- #
# Environments.
# --------------------------------------------------------------------------- #
def dummy_env_step(state: torch.Tensor, action: torch.Tensor, reward_out: torch.Tensor) -> None:
    n = state.numel()
    discrete = action.dtype == torch.int64
    if not discrete and action.dtype != torch.float32:
        raise TypeError(f"dummy env actions must be int64 or float32, got {action.dtype}")
    _dense(state, torch.float32, "state")
    _dense(reward_out, torch.float32, "reward_out")
    if action.numel() != n or reward_out.numel() != n or not action.is_contiguous():
        raise ValueError("state, action and reward_out must be dense with one element per env")
    _check(
        load().rl8_dummy_env_step_f32(_ptr(state), _ptr(action), int(discrete), _ptr(reward_out), n, _stream()),
        "rl8_dummy_env_step_f32",
    )


def dummy_env_reset(state: torch.Tensor, bounds: float, seed: int, reset_count: int, env_offset: int = 0) -> None:
    _dense(state, torch.float32, "state")
    _check(
        load().rl8_dummy_env_reset_f32(_ptr(state), state.numel(), bounds, seed, reset_count, env_offset, _stream()),
        "rl8_dummy_env_reset_f32",
    )


def cartpole_step(state: torch.Tensor, action: torch.Tensor, cfg: CartPoleCfg, obs_out: torch.Tensor, reward_out: torch.Tensor) -> None:
    _dense(state, torch.float32, "state")
    _dense(action, torch.int64, "action")
    _dense(obs_out, torch.float32, "obs_out")
    _dense(reward_out, torch.float32, "reward_out")
    n = state.shape[1]
    if state.shape[0] != 4 or action.numel() != n or obs_out.numel() != 5 * n or reward_out.numel() != n:
        raise ValueError("cartpole_step: state [4,N], action [N,1], obs_out [N,5], reward_out [N,1]")
    _check(
        load().rl8_cartpole_step_f32(_ptr(state), _ptr(action), C.byref(cfg), _ptr(obs_out), 5, _ptr(reward_out), n, _stream()),
        "rl8_cartpole_step_f32",
    )


def cartpole_reset(state: torch.Tensor, std: float, seed: int, reset_count: int, env_offset: int, obs_out: None | torch.Tensor) -> None:
    _dense(state, torch.float32, "state")
    n = state.shape[1]
    if obs_out is not None:
        _dense(obs_out, torch.float32, "obs_out")
        if obs_out.numel() != 5 * n:
            raise ValueError("obs_out must be [N,5]")
    _check(
        load().rl8_cartpole_reset_f32(_ptr(state), n, std, seed, reset_count, env_offset, _ptr(obs_out), 5, _stream()),
        "rl8_cartpole_reset_f32",
    )


def _classic_step(symbol: str, obs_dim: int, action_dtype: torch.dtype, state: torch.Tensor, action: torch.Tensor,
                  cfg: C.Structure, obs_out: torch.Tensor, reward_out: torch.Tensor) -> None:
    _dense(state, torch.float32, "state")
    _dense(action, action_dtype, "action")
    _dense(obs_out, torch.float32, "obs_out")
    _dense(reward_out, torch.float32, "reward_out")
    n = state.shape[1]
    if state.shape[0] != 2 or action.numel() != n or obs_out.numel() != obs_dim * n or reward_out.numel() != n:
        raise ValueError(f"{symbol}: state [2,N], action [N,1], obs_out [N,{obs_dim}], reward_out [N,1]")
    _check(
        getattr(load(), symbol)(_ptr(state), _ptr(action), C.byref(cfg), _ptr(obs_out), obs_dim, _ptr(reward_out), n, _stream()),
        symbol,
    )


def _classic_reset(symbol: str, obs_dim: int, state: torch.Tensor, seed: int, reset_count: int, env_offset: int,
                   obs_out: None | torch.Tensor) -> None:
    _dense(state, torch.float32, "state")
    n = state.shape[1]
    if state.shape[0] != 2:
        raise ValueError(f"{symbol}: state must be [2,N]")
    if obs_out is not None:
        _dense(obs_out, torch.float32, "obs_out")
        if obs_out.numel() != obs_dim * n:
            raise ValueError(f"obs_out must be [N,{obs_dim}]")
    _check(getattr(load(), symbol)(_ptr(state), n, seed, reset_count, env_offset, _ptr(obs_out), obs_dim, _stream()), symbol)


def mountain_car_step(state: torch.Tensor, action: torch.Tensor, cfg: MountainCarCfg, obs_out: torch.Tensor,
                      reward_out: torch.Tensor) -> None:
    """examples/mountain_car/env.py:12-38: state [2,N] in place, obs_out [N,2], reward_out [N,1]."""
    _classic_step("rl8_mountain_car_step_f32", 2, torch.int64, state, action, cfg, obs_out, reward_out)


def mountain_car_reset(state: torch.Tensor, seed: int, reset_count: int, env_offset: int, obs_out: None | torch.Tensor) -> None:
    _classic_reset("rl8_mountain_car_reset_f32", 2, state, seed, reset_count, env_offset, obs_out)


def pendulum_step(state: torch.Tensor, action: torch.Tensor, cfg: PendulumCfg, obs_out: torch.Tensor,
                  reward_out: torch.Tensor) -> None:
    """examples/pendulum/env.py:12-39: state [2,N] in place, obs_out [N,3], reward_out [N,1]."""
    _classic_step("rl8_pendulum_step_f32", 3, torch.float32, state, action, cfg, obs_out, reward_out)


def pendulum_reset(state: torch.Tensor, seed: int, reset_count: int, env_offset: int, obs_out: None | torch.Tensor) -> None:
    _classic_reset("rl8_pendulum_reset_f32", 3, state, seed, reset_count, env_offset, obs_out)


# --------------------------------------------------------------------------- #
# Samplers.
# --------------------------------------------------------------------------- #
def categorical_sample_logp(
    logits: torch.Tensor,
    noise: None | torch.Tensor,
    *,
    seed: int = 0,
    step: int = 0,
    row_offset: int = 0,
    deterministic: bool = False,
) -> tuple[torch.Tensor, torch.Tensor]:
    """logits [M, A, K] -> (actions [M, A] int64, logp [M, 1])."""
    logits = _dense(logits.detach(), torch.float32, "logits")
    m, a, k = logits.shape
    if noise is not None:
        noise = _dense(noise, torch.float32, "noise")
        if noise.shape != logits.shape:
            raise ValueError("noise must have the shape of logits")
    actions = torch.empty(m, a, dtype=torch.int64, device=logits.device)
    logp = torch.empty(m, 1, dtype=torch.float32, device=logits.device)
    _check(
        load().rl8_categorical_sample_logp_f32(
            _ptr(logits), _ptr(noise), _ptr(actions), _ptr(logp), m, a, k, seed, step, row_offset,
            int(deterministic), _stream(),
        ),
        "rl8_categorical_sample_logp_f32",
    )
    return actions, logp


def normal_sample_logp(
    mean: torch.Tensor,
    log_std: torch.Tensor,
    noise: None | torch.Tensor,
    *,
    squashed: bool,
    seed: int = 0,
    step: int = 0,
    row_offset: int = 0,
    deterministic: bool = False,
) -> tuple[torch.Tensor, torch.Tensor]:
    mean = _dense(mean.detach(), torch.float32, "mean")
    log_std = _dense(log_std.detach(), torch.float32, "log_std")
    m, a = mean.shape
    if log_std.shape != mean.shape:
        raise ValueError("mean and log_std must have the same shape")
    if noise is not None:
        noise = _dense(noise, torch.float32, "noise")
        if noise.shape != mean.shape:
            raise ValueError("noise must have the shape of mean")
    actions = torch.empty(m, a, dtype=torch.float32, device=mean.device)
    logp = torch.empty(m, 1, dtype=torch.float32, device=mean.device)
    _check(
        load().rl8_normal_sample_logp_f32(
            _ptr(mean), _ptr(log_std), _ptr(noise), _ptr(actions), _ptr(logp), m, a, int(squashed),
            seed, step, row_offset, int(deterministic), _stream(),
        ),
        "rl8_normal_sample_logp_f32",
    )
    return actions, logp


# --------------------------------------------------------------------------- #
# Rollout bookkeeping.
# --------------------------------------------------------------------------- #
def rollout_scatter(
    action: torch.Tensor, logp: torch.Tensor, value: torch.Tensor, reward: torch.Tensor, obs: torch.Tensor,
    action_col: torch.Tensor, logp_col: torch.Tensor, value_col: torch.Tensor, reward_col: torch.Tensor,
    obs_col_next: torch.Tensor, rdr_t: None | torch.Tensor, rdr_t1: None | torch.Tensor, gamma: float,
) -> None:
    n = logp.shape[0]
    for name, t in (("action", action), ("logp", logp), ("value", value), ("reward", reward), ("obs", obs),
                    ("action_col", action_col), ("logp_col", logp_col), ("value_col", value_col),
                    ("reward_col", reward_col), ("obs_col_next", obs_col_next)):
        if not t.is_contiguous():
            raise ValueError(f"{name} must be contiguous")
    if action.dtype != action_col.dtype or action.numel() != action_col.numel():
        raise ValueError("action and action_col must match")
    if obs.numel() != obs_col_next.numel() or obs.dtype != torch.float32:
        raise ValueError("obs and obs_col_next must match (float32)")
    _check(
        load().rl8_rollout_scatter_f32(
            _ptr(action), action.element_size() * (action.numel() // n), _ptr(logp), _ptr(value), _ptr(reward),
            _ptr(obs), obs.numel() // n, _ptr(action_col), _ptr(logp_col), _ptr(value_col), _ptr(reward_col),
            _ptr(obs_col_next), _ptr(rdr_t), _ptr(rdr_t1), gamma, n, _stream(),
        ),
        "rl8_rollout_scatter_f32",
    )


def rollout_step_dummy(
    *, discrete: bool, squashed: bool, features: torch.Tensor, features2: None | torch.Tensor, value: torch.Tensor,
    noise: None | torch.Tensor, state: torch.Tensor, action_col: torch.Tensor, logp_col: torch.Tensor,
    value_col: torch.Tensor, reward_col: torch.Tensor, obs_col_next: torch.Tensor, rdr_t: None | torch.Tensor,
    rdr_t1: None | torch.Tensor, gamma: float, seed: int, step: int, env_offset: int, deterministic: bool,
) -> None:
    n = state.numel()
    tensors = [features, value, state, action_col, logp_col, value_col, reward_col, obs_col_next]
    tensors += [t for t in (features2, noise, rdr_t, rdr_t1) if t is not None]
    for t in tensors:
        if not t.is_contiguous():
            raise ValueError("rollout_step_dummy: all tensors must be contiguous")
    want_feat = 2 * n if discrete else n
    if features.numel() != want_feat or value.numel() != n or action_col.numel() != n:
        raise ValueError("rollout_step_dummy: shape mismatch")
    if noise is not None and noise.numel() != want_feat:
        raise ValueError("rollout_step_dummy: noise shape mismatch")
    if action_col.dtype != (torch.int64 if discrete else torch.float32):
        raise TypeError("rollout_step_dummy: action column dtype mismatch")
    for t in (logp_col, value_col, reward_col, obs_col_next):
        if t.numel() != n or t.dtype != torch.float32:
            raise ValueError("rollout_step_dummy: column shape/dtype mismatch")
    with _timed("rollout_step_dummy", n):
        _check(
            load().rl8_rollout_step_dummy_f32(
                int(discrete), int(squashed), _ptr(features), _ptr(features2), _ptr(value), _ptr(noise),
                _ptr(state), _ptr(action_col), _ptr(logp_col), _ptr(value_col), _ptr(reward_col),
                _ptr(obs_col_next), _ptr(rdr_t), _ptr(rdr_t1), gamma, n, seed, step, env_offset,
                int(deterministic), _stream(),
            ),
            "rl8_rollout_step_dummy_f32",
        )


def rollout_step_cartpole(
    *, logits: torch.Tensor, value: torch.Tensor, noise: None | torch.Tensor, state: torch.Tensor, cfg: CartPoleCfg,
    action_col: torch.Tensor, logp_col: torch.Tensor, value_col: torch.Tensor, reward_col: torch.Tensor,
    obs_col_next: torch.Tensor, rdr_t: None | torch.Tensor, rdr_t1: None | torch.Tensor, gamma: float, seed: int,
    step: int, env_offset: int, deterministic: bool,
) -> None:
    n = state.shape[1]
    tensors = [logits, value, state, action_col, logp_col, value_col, reward_col, obs_col_next]
    tensors += [t for t in (noise, rdr_t, rdr_t1) if t is not None]
    for t in tensors:
        if not t.is_contiguous():
            raise ValueError("rollout_step_cartpole: all tensors must be contiguous")
    if logits.numel() != 3 * n or value.numel() != n or obs_col_next.numel() != 5 * n or action_col.numel() != n:
        raise ValueError("rollout_step_cartpole: shape mismatch")
    if noise is not None and noise.numel() != 3 * n:
        raise ValueError("rollout_step_cartpole: noise shape mismatch")
    _check(
        load().rl8_rollout_step_cartpole_f32(
            _ptr(logits), _ptr(value), _ptr(noise), _ptr(state), C.byref(cfg), _ptr(action_col), _ptr(logp_col),
            _ptr(value_col), _ptr(reward_col), _ptr(obs_col_next), _ptr(rdr_t), _ptr(rdr_t1), gamma, n, seed,
            step, env_offset, int(deterministic), _stream(),
        ),
        "rl8_rollout_step_cartpole_f32",
    )


def _check_step_tensors(name: str, tensors: list[None | torch.Tensor]) -> None:
    for t in tensors:
        if t is not None and not t.is_contiguous():
            raise ValueError(f"{name}: all tensors must be contiguous")


def rollout_step_mountain_car(
    *, logits: torch.Tensor, value: torch.Tensor, noise: None | torch.Tensor, state: torch.Tensor, cfg: MountainCarCfg,
    action_col: torch.Tensor, logp_col: torch.Tensor, value_col: torch.Tensor, reward_col: torch.Tensor,
    obs_col_next: torch.Tensor, rdr_t: None | torch.Tensor, rdr_t1: None | torch.Tensor, gamma: float, seed: int,
    step: int, env_offset: int, deterministic: bool,
) -> None:
    n = state.shape[1]
    _check_step_tensors("rollout_step_mountain_car", [logits, value, state, action_col, logp_col, value_col,
                                                      reward_col, obs_col_next, noise, rdr_t, rdr_t1])
    if logits.numel() != 3 * n or value.numel() != n or obs_col_next.numel() != 2 * n or action_col.numel() != n:
        raise ValueError("rollout_step_mountain_car: shape mismatch")
    if noise is not None and noise.numel() != 3 * n:
        raise ValueError("rollout_step_mountain_car: noise shape mismatch")
    _check(
        load().rl8_rollout_step_mountain_car_f32(
            _ptr(logits), _ptr(value), _ptr(noise), _ptr(state), C.byref(cfg), _ptr(action_col), _ptr(logp_col),
            _ptr(value_col), _ptr(reward_col), _ptr(obs_col_next), _ptr(rdr_t), _ptr(rdr_t1), gamma, n, seed,
            step, env_offset, int(deterministic), _stream(),
        ),
        "rl8_rollout_step_mountain_car_f32",
    )


def rollout_step_pendulum(
    *, squashed: bool, mean: torch.Tensor, log_std: torch.Tensor, value: torch.Tensor, noise: None | torch.Tensor,
    state: torch.Tensor, cfg: PendulumCfg, action_col: torch.Tensor, logp_col: torch.Tensor, value_col: torch.Tensor,
    reward_col: torch.Tensor, obs_col_next: torch.Tensor, rdr_t: None | torch.Tensor, rdr_t1: None | torch.Tensor,
    gamma: float, seed: int, step: int, env_offset: int, deterministic: bool,
) -> None:
    n = state.shape[1]
    _check_step_tensors("rollout_step_pendulum", [mean, log_std, value, state, action_col, logp_col, value_col,
                                                  reward_col, obs_col_next, noise, rdr_t, rdr_t1])
    if mean.numel() != n or log_std.numel() != n or value.numel() != n or obs_col_next.numel() != 3 * n or action_col.numel() != n:
        raise ValueError("rollout_step_pendulum: shape mismatch")
    if action_col.dtype != torch.float32:
        raise ValueError("rollout_step_pendulum: the action column must be float32")
    if noise is not None and noise.numel() != n:
        raise ValueError("rollout_step_pendulum: noise shape mismatch")
    _check(
        load().rl8_rollout_step_pendulum_f32(
            int(squashed), _ptr(mean), _ptr(log_std), _ptr(value), _ptr(noise), _ptr(state), C.byref(cfg),
            _ptr(action_col), _ptr(logp_col), _ptr(value_col), _ptr(reward_col), _ptr(obs_col_next), _ptr(rdr_t),
            _ptr(rdr_t1), gamma, n, seed, step, env_offset, int(deterministic), _stream(),
        ),
        "rl8_rollout_step_pendulum_f32",
    )


def buffer_layout(leaf: torch.Tensor) -> tuple[int, int, int]:
    """(layout, env_stride, time_stride) of a [N, T, 1] (or [N, T]) buffer leaf."""
    n, t = leaf.shape[0], leaf.shape[1]
    es, ts = leaf.stride(0), leaf.stride(1)
    if leaf.ndim == 3 and leaf.shape[2] != 1:
        raise ValueError("expected a [N, T, 1] leaf")
    if (ts == 1 or t == 1) and (es == t or n == 1):
        return LAYOUT_ENV_MAJOR, es, ts
    if (es == 1 or n == 1) and (ts == n or t == 1):
        return LAYOUT_TIME_MAJOR, es, ts
    return -1, es, ts


def rollout_stats(rewards: torch.Tensor, rdr: None | torch.Tensor) -> torch.Tensor:
    """rewards / rdr [N, H+1, 1] in either layout -> 12 raw moments (fp64, device)."""
    n, h1 = rewards.shape[0], rewards.shape[1]
    _, es, ts = buffer_layout(rewards)
    if rewards.dtype != torch.float32:
        raise TypeError("rewards must be float32")
    if rdr is not None and (rdr.stride() != rewards.stride() or rdr.shape != rewards.shape):
        raise ValueError("rdr must share the layout of rewards")
    out = torch.empty(12, dtype=torch.float64, device=rewards.device)
    with _timed("rollout_stats", n * (h1 - 1)):
        _check(
            load().rl8_rollout_stats_f32(_ptr(rewards), _ptr(rdr), n, h1 - 1, es, ts, _ptr(out),
                                         _ptr(scratch(rewards.device)), _stream()),
            "rl8_rollout_stats_f32",
        )
    return out


# --------------------------------------------------------------------------- #
# GAE.
# --------------------------------------------------------------------------- #
def gae_scan(
    rewards: torch.Tensor, values: torch.Tensor, adv: torch.Tensor, ret: torch.Tensor, *, layout: int,
    n: int, h: int, gamma: float, gamma_lambda: float, reward_denominator: float, write_scaled_rewards: bool,
) -> torch.Tensor:
    """Launches the scan; returns the device moments tensor (count, sum, sumsq)."""
    moments = torch.empty(3, dtype=torch.float64, device=rewards.device)
    with _timed("gae_scan", n * h):
        _check(
            load().rl8_gae_scan_f32(
                _ptr(rewards), _ptr(values), _ptr(adv), _ptr(ret), n, h, layout, gamma, gamma_lambda,
                reward_denominator, int(write_scaled_rewards), _ptr(moments), _ptr(scratch(rewards.device)),
                _stream(),
            ),
            "rl8_gae_scan_f32",
        )
    return moments


def advantage_normalise(adv: torch.Tensor, *, layout: int, n: int, h: int, moments: torch.Tensor) -> None:
    if moments.dtype != torch.float64 or moments.numel() < 3:
        raise TypeError("moments must be 3 float64 values")
    with _timed("advantage_normalise", n * h):
        _check(
            load().rl8_advantage_normalise_f32(_ptr(adv), n, h, layout, _ptr(moments), _stream()),
            "rl8_advantage_normalise_f32",
        )


# --------------------------------------------------------------------------- #
# PPO loss.
# --------------------------------------------------------------------------- #
def ppo_hparams(
    *, clip_param: float, dual_clip_param: None | float, entropy_coeff: float, vf_clip_param: float,
    vf_coeff: float, grad_scale: float,
) -> PPOHparams:
    return PPOHparams(clip_param, dual_clip_param if dual_clip_param else 0.0, entropy_coeff, vf_clip_param,
                      vf_coeff, grad_scale)


def _loss_inputs(m: int, *named: tuple[str, torch.Tensor, torch.dtype, int]) -> None:
    for name, t, dtype, numel in named:
        _dense(t, dtype, name)
        if t.numel() != numel:
            raise ValueError(f"{name} has {t.numel()} elements, expected {numel} for M={m}")


def ppo_loss_categorical(
    logits: torch.Tensor, value: torch.Tensor, action: torch.Tensor, logp_old: torch.Tensor, adv: torch.Tensor,
    ret: torch.Tensor, hp: PPOHparams, *, with_grad: bool = True,
) -> tuple[torch.Tensor, None | torch.Tensor, None | torch.Tensor]:
    """Returns (loss_sums[5] fp64 device, grad_logits, grad_value)."""
    m, a, k = logits.shape
    _loss_inputs(m, ("logits", logits, torch.float32, m * a * k), ("value", value, torch.float32, m),
                 ("action", action, torch.int64, m * a), ("logp_old", logp_old, torch.float32, m),
                 ("adv", adv, torch.float32, m), ("ret", ret, torch.float32, m))
    sums = torch.empty(5, dtype=torch.float64, device=logits.device)
    g_logits = torch.empty_like(logits) if with_grad else None
    g_value = torch.empty_like(value) if with_grad else None
    with _timed("ppo_loss_categorical", m):
        _check(
            load().rl8_ppo_loss_categorical_fwd_bwd_f32(
                _ptr(logits), _ptr(value), _ptr(action), _ptr(logp_old), _ptr(adv), _ptr(ret), m, a, k,
                C.byref(hp), _ptr(g_logits), _ptr(g_value), _ptr(sums), _ptr(scratch(logits.device)), _stream(),
            ),
            "rl8_ppo_loss_categorical_fwd_bwd_f32",
        )
    return sums, g_logits, g_value


def ppo_loss_normal(
    mean: torch.Tensor, log_std: torch.Tensor, value: torch.Tensor, action: torch.Tensor, logp_old: torch.Tensor,
    adv: torch.Tensor, ret: torch.Tensor, hp: PPOHparams, *, squashed: bool, with_grad: bool = True,
):
    m, a = mean.shape
    _loss_inputs(m, ("mean", mean, torch.float32, m * a), ("log_std", log_std, torch.float32, m * a),
                 ("value", value, torch.float32, m), ("action", action, torch.float32, m * a),
                 ("logp_old", logp_old, torch.float32, m), ("adv", adv, torch.float32, m),
                 ("ret", ret, torch.float32, m))
    sums = torch.empty(5, dtype=torch.float64, device=mean.device)
    g_mean = torch.empty_like(mean) if with_grad else None
    g_ls = torch.empty_like(log_std) if with_grad else None
    g_value = torch.empty_like(value) if with_grad else None
    with _timed("ppo_loss_normal", m):
        _check(
            load().rl8_ppo_loss_normal_fwd_bwd_f32(
                _ptr(mean), _ptr(log_std), _ptr(value), _ptr(action), _ptr(logp_old), _ptr(adv), _ptr(ret), m, a,
                int(squashed), C.byref(hp), _ptr(g_mean), _ptr(g_ls), _ptr(g_value), _ptr(sums),
                _ptr(scratch(mean.device)), _stream(),
            ),
            "rl8_ppo_loss_normal_fwd_bwd_f32",
        )
    return sums, g_mean, g_ls, g_value


# --------------------------------------------------------------------------- #
# Minibatch gather.
# --------------------------------------------------------------------------- #
def gather_minibatch(index: None | torch.Tensor, h: int, leaves: Sequence[torch.Tensor]) -> list[torch.Tensor]:
    """index [M] int64 of reference sample ids (env*H + t); leaves are [N, T, d]
    buffer leaves (any stride over env/time, dense over d). Returns dense [M, d]
    tensors. ``index=None``: every sample in order (``M = N * h``): a tiled
    transposition for leaves of up to 128 bytes per sample (in groups that fit a
    tile row), the general kernels with the implicit index for wider ones."""
    if index is None:
        m = leaves[0].shape[0] * h
    else:
        _dense(index, torch.int64, "index")
        m = index.numel()
    if len(leaves) > MAX_GATHER_FIELDS:
        raise ValueError(f"at most {MAX_GATHER_FIELDS} leaves per gather")
    fields = (GatherField * len(leaves))()
    outs = []
    for i, leaf in enumerate(leaves):
        if leaf.ndim < 2:
            raise ValueError("leaves must be [N, T, ...]")
        trailing = leaf.shape[2:]
        row = 1
        for d in trailing:
            row *= d
        # trailing dims must be dense
        expect = 1
        for d, st in zip(reversed(trailing), reversed(leaf.stride()[2:])):
            if d != 1 and st != expect:
                raise ValueError("leaf trailing dims must be dense")
            expect *= d
        if leaf.element_size() not in (4, 8):
            raise TypeError("leaf elements must be 4 or 8 bytes wide")
        dst = torch.empty((m, *trailing), dtype=leaf.dtype, device=leaf.device)
        fields[i] = GatherField(_ptr(leaf), _ptr(dst), leaf.stride(0), leaf.stride(1), row, leaf.element_size())
        outs.append(dst)
    with _timed("gather_minibatch", m):
        _check(load().rl8_gather_minibatch(_ptr(index) if index is not None else None, m, h, fields, len(leaves), _stream()),
               "rl8_gather_minibatch")
    return outs


class PackedSamples:
    """The training fields of every sample of a rollout buffer, side by side in
    the reference's sample order (``rl8_pack_samples``); ``gather(index)`` returns
    the dense minibatch tensors, one per field."""

    def __init__(self, h: int, leaves: Sequence[torch.Tensor]) -> None:
        if not leaves or len(leaves) > MAX_GATHER_FIELDS:
            raise ValueError(f"between 1 and {MAX_GATHER_FIELDS} leaves per pack")
        n = leaves[0].shape[0]
        self.fields = (GatherField * len(leaves))()
        self.meta: list[tuple[tuple[int, ...], torch.dtype]] = []
        words = 0
        for i, leaf in enumerate(leaves):
            if leaf.ndim < 2 or leaf.shape[0] != n or leaf.shape[1] < h:
                raise ValueError("leaves must be [N, >= H, ...] views of one buffer")
            if leaf.element_size() not in (4, 8):
                raise TypeError("leaf elements must be 4 or 8 bytes wide")
            trailing = tuple(leaf.shape[2:])
            row = 1
            for d in trailing:
                row *= d
            if leaf[0, 0].numel() and not leaf[0, 0].is_contiguous():
                raise ValueError("leaf trailing dims must be dense")
            self.fields[i] = GatherField(_ptr(leaf), None, leaf.stride(0), leaf.stride(1), row, leaf.element_size())
            self.meta.append((trailing, leaf.dtype))
            words += row * (leaf.element_size() // 4)
        self.row_words = (words + 3) // 4 * 4
        self.samples = n * h
        self.device = leaves[0].device
        self.packed = torch.empty(self.samples * self.row_words, dtype=torch.int32, device=self.device)
        with _timed("pack_samples", self.samples):
            _check(load().rl8_pack_samples(self.fields, len(leaves), n, h, _ptr(self.packed), self.row_words, _stream()),
                   "rl8_pack_samples")

    def gather(self, index: torch.Tensor) -> list[torch.Tensor]:
        _dense(index, torch.int64, "index")
        m = index.numel()
        outs = []
        for i, (trailing, dtype) in enumerate(self.meta):
            dst = torch.empty((m, *trailing), dtype=dtype, device=self.device)
            self.fields[i].dst = _ptr(dst)
            outs.append(dst)
        with _timed("gather_packed", m):
            _check(load().rl8_gather_packed(_ptr(index), m, _ptr(self.packed), self.row_words, self.fields,
                                            len(self.meta), _stream()), "rl8_gather_packed")
        return outs


# --------------------------------------------------------------------------- #
# Fused MLP tower (N1).
# --------------------------------------------------------------------------- #
MLP_HIDDEN = 256
MLP_MAX_IN = 16
MLP_MAX_OUT = 8


def mlp_pack_w2(w2: torch.Tensor, *, transposed: bool = False) -> torch.Tensor:
    """[256, 256] nn.Linear weight -> MFMA fragment order (65536 floats)."""
    w2 = _dense(w2.detach(), torch.float32, "w2")
    if tuple(w2.shape) != (MLP_HIDDEN, MLP_HIDDEN):
        raise ValueError("w2 must be [256, 256]")
    packed = torch.empty(MLP_HIDDEN * MLP_HIDDEN, dtype=torch.float32, device=w2.device)
    _check(load().rl8_mlp_pack_w2_f32(_ptr(w2), _ptr(packed), int(transposed), _stream()), "rl8_mlp_pack_w2_f32")
    return packed


def mlp_tower_forward(
    x: torch.Tensor, w1: torch.Tensor, b1: torch.Tensor, w2_packed: torch.Tensor, b2: torch.Tensor,
    w3: torch.Tensor, b3: torch.Tensor, *, save: bool = False,
) -> tuple[torch.Tensor, None | torch.Tensor, None | torch.Tensor]:
    """x [M, d_in] -> out [M, n_out]; with ``save`` also the post-ReLU activations
    h1, h2 ([M, 256]) for the backward pass."""
    x = _dense(x.detach(), torch.float32, "x")
    m, d_in = x.shape
    n_out = w3.shape[0]
    for name, t, shape in (("w1", w1, (MLP_HIDDEN, d_in)), ("b1", b1, (MLP_HIDDEN,)), ("b2", b2, (MLP_HIDDEN,)),
                           ("w3", w3, (n_out, MLP_HIDDEN)), ("b3", b3, (n_out,))):
        _dense(t.detach(), torch.float32, name)
        if tuple(t.shape) != shape:
            raise ValueError(f"{name} must have shape {shape}, got {tuple(t.shape)}")
    if w2_packed.numel() != MLP_HIDDEN * MLP_HIDDEN:
        raise ValueError("w2_packed must come from mlp_pack_w2")
    out = torch.empty(m, n_out, dtype=torch.float32, device=x.device)
    h1 = torch.empty(m, MLP_HIDDEN, dtype=torch.float32, device=x.device) if save else None
    h2 = torch.empty(m, MLP_HIDDEN, dtype=torch.float32, device=x.device) if save else None
    with _timed("mlp_tower_forward_save" if save else "mlp_tower_forward", m):
        _check(
            load().rl8_mlp_tower_forward_f32(
                _ptr(x), m, d_in, _ptr(w1.detach()), _ptr(b1.detach()), _ptr(w2_packed), _ptr(b2.detach()),
                _ptr(w3.detach()), _ptr(b3.detach()), n_out, _ptr(out), _ptr(h1), _ptr(h2), _stream(),
            ),
            "rl8_mlp_tower_forward_f32",
        )
    return out, h1, h2


def mlp_forward_f16_supports(d_in: int, n_out: int) -> bool:
    return bool(load().rl8_mlp_forward_f16_supports(int(d_in), int(n_out)))


def mlp_backward_f16_supports(d_in: int, n_out: int) -> bool:
    return bool(load().rl8_mlp_backward_f16_supports(int(d_in), int(n_out)))


def mlp_pack_w2_f16(w2: torch.Tensor, *, transposed: bool = False) -> torch.Tensor:
    """[256, 256] nn.Linear weight -> two fp16 planes of 2^e * w (hi = fp16(v),
    lo = fp16(v - hi)) in fragment order, followed by {2^e, 2^-e} (uint8)."""
    w2 = _dense(w2.detach(), torch.float32, "w2")
    if tuple(w2.shape) != (MLP_HIDDEN, MLP_HIDDEN):
        raise ValueError("w2 must be [256, 256]")
    lib = load()
    packed = torch.empty(int(lib.rl8_mlp_f16_packed_bytes()), dtype=torch.uint8, device=w2.device)
    _check(lib.rl8_mlp_pack_w2_f16(_ptr(w2), int(transposed), _ptr(packed), _stream()), "rl8_mlp_pack_w2_f16")
    return packed


def mlp_pack_w2_f16_gate(w2: torch.Tensor, w3: torch.Tensor) -> torch.Tensor:
    """B operand of the data-gradient kernel's gate mode: fp16 planes of ``w2[k][i] * w3e[k]`` with
    ``w3e = w3[0]`` (one output) or ``w3[0] - w3[1]`` (two outputs with opposite gradients)."""
    w2 = _dense(w2.detach(), torch.float32, "w2")
    w3 = _dense(w3.detach(), torch.float32, "w3")
    if tuple(w2.shape) != (MLP_HIDDEN, MLP_HIDDEN) or w3.ndim != 2 or w3.shape[1] != MLP_HIDDEN or w3.shape[0] not in (1, 2):
        raise ValueError("w2 must be [256, 256] and w3 [1 or 2, 256]")
    lib = load()
    packed = torch.empty(int(lib.rl8_mlp_f16_packed_bytes()), dtype=torch.uint8, device=w2.device)
    _check(lib.rl8_mlp_pack_w2_f16_gate(_ptr(w2), _ptr(w3), int(w3.shape[0]), _ptr(packed), _stream()),
           "rl8_mlp_pack_w2_f16_gate")
    return packed


def mlp_tower_forward_split(
    x: torch.Tensor, w1: torch.Tensor, b1: torch.Tensor, w2_split: torch.Tensor, b2: torch.Tensor,
    w3: torch.Tensor, b3: torch.Tensor, *, save: bool = False, save_h1: bool = True, save_gate: bool = False,
    save_h2: bool = True, out: None | torch.Tensor = None, h2_out: None | torch.Tensor = None,
    gate_out: None | torch.Tensor = None, timer_name: None | str = None,
) -> tuple[torch.Tensor, ...]:
    """``mlp_tower_forward`` with the 256x256 product as three fp16-plane MFMAs per
    16 k (fp32 accuracy, fp32 in / out / accumulate; ``w2_split`` from
    ``mlp_pack_w2_f16``). ``save_h1=False`` keeps only h2 (the plane backward kernels
    recompute h1). ``save_gate=True`` (with
    ``save``) appends a fourth result, the ReLU gate of h2 as bits ([M, 8] int32:
    bit j of row s = h2[s, j] > 0), which the data-gradient kernel reads instead of h2.
    ``save_h2=False`` (fp16-plane pack, with ``save_gate``): the gate bits ALONE are kept -- all the
    backward pass of a rank-one head needs (``mlp_tower_backward(..., h2=None, w2=..., b2=...)``).
    ``out`` / ``h2_out`` / ``gate_out``: caller-owned dense destinations ([M, n_out] fp32, [M, 256] fp32,
    [M, 8] int32) instead of fresh tensors -- the rollout writes each timestep's rows straight into the slabs the
    first SGD pass reads back (``fused_mlp.RolloutRecord``)."""
    x = _dense(x.detach(), torch.float32, "x")
    m, d_in = x.shape
    n_out = w3.shape[0]
    for name, t, shape in (("w1", w1, (MLP_HIDDEN, d_in)), ("b1", b1, (MLP_HIDDEN,)), ("b2", b2, (MLP_HIDDEN,)),
                           ("w3", w3, (n_out, MLP_HIDDEN)), ("b3", b3, (n_out,))):
        _dense(t.detach(), torch.float32, name)
        if tuple(t.shape) != shape:
            raise ValueError(f"{name} must have shape {shape}, got {tuple(t.shape)}")
    lib = load()
    if w2_split.dtype == torch.uint8 and w2_split.numel() == int(lib.rl8_mlp_f16_packed_bytes()):
        fn, fn_name = lib.rl8_mlp_tower_forward_f16_f32, "rl8_mlp_tower_forward_f16_f32"  # fp16 two-plane pack
    else:
        raise ValueError("w2_split must come from mlp_pack_w2_f16")
    for name, t, dtype, shape in (("out", out, torch.float32, (m, n_out)), ("h2_out", h2_out, torch.float32, (m, MLP_HIDDEN)),
                                  ("gate_out", gate_out, torch.int32, (m, 8))):
        if t is not None:
            _dense(t, dtype, name)
            if tuple(t.shape) != shape:
                raise ValueError(f"{name} must have shape {shape}, got {tuple(t.shape)}")
    if out is None:
        out = torch.empty(m, n_out, dtype=torch.float32, device=x.device)
    if save and not save_h2:
        if not save_gate:
            raise ValueError("save_h2=False needs save_gate=True")
        save_h1 = False
    h1 = torch.empty(m, MLP_HIDDEN, dtype=torch.float32, device=x.device) if save and save_h1 else None
    h2 = (h2_out if h2_out is not None else torch.empty(m, MLP_HIDDEN, dtype=torch.float32, device=x.device)) \
        if save and save_h2 else None
    gate = (gate_out if gate_out is not None else torch.empty(m, 8, dtype=torch.int32, device=x.device)) \
        if save and save_gate else None
    with _timed(timer_name or ("mlp_tower_forward_save" if save else "mlp_tower_forward"), m):
        _check(
            fn(
                _ptr(x), m, d_in, _ptr(w1.detach()), _ptr(b1.detach()), _ptr(w2_split), _ptr(b2.detach()),
                _ptr(w3.detach()), _ptr(b3.detach()), n_out, _ptr(out), _ptr(h1), _ptr(h2), _ptr(gate), _stream(),
            ),
            fn_name,
        )
    return (out, h1, h2, gate) if save_gate else (out, h1, h2)


def _column_sums(t: torch.Tensor) -> torch.Tensor:
    """``t.sum(0)`` of a tall ``[m, n]`` matrix with a few columns.  torch's reduction over
    the long dimension of a [2^23, 3] tensor runs at 24 GB/s (4.1 ms per call, 14 % of the
    CartPole bench's kernel time); folded to [m / 1024, 1024 n] the same sum is a column
    reduction with a contiguous inner dimension."""
    m, n = t.shape
    fold = 1024
    if n == 1 or m < 64 * fold or not t.is_contiguous():
        return t.sum(0)
    groups = m // fold
    out = t[: groups * fold].view(groups, fold * n).sum(0).view(fold, n).sum(0)
    if groups * fold < m:
        out = out + t[groups * fold :].sum(0)
    return out


def mlp_tower_backward(
    x: torch.Tensor, h1: None | torch.Tensor, h2: None | torch.Tensor, dout: torch.Tensor,
    w2t_packed: torch.Tensor, w3: torch.Tensor,
    w1: None | torch.Tensor = None, b1: None | torch.Tensor = None, *, wgrad_split: bool = False,
    gate2: None | torch.Tensor = None, gate_pack=None, w2: None | torch.Tensor = None, b2: None | torch.Tensor = None,
    h2_fn=None, info: None | dict = None, assume_general: bool = False, assume_pair: bool = False,
) -> dict[str, torch.Tensor]:
    """Gradients of one tower's parameters given ``dout`` [M, n_out] and the
    activations saved by the forward pass. Returns ``w1, b1, w2, b2, w3, b3``.

    ``w2t_packed`` from ``mlp_pack_w2(..., transposed=True)`` selects the fp32 MFMA
    kernel; from ``mlp_pack_w2_f16(..., transposed=True)`` (uint8) the fp16-plane
    kernels (fused mode), which also need layer 1 (``w1``, ``b1``) -- they recompute
    the ReLU gate of h1 instead of reading h1 back -- and ``gate2`` (``save_gate`` of
    the forward): the gate bits of h2. ``wgrad_split`` forms dW2 with the plane
    weight-gradient kernel (any width; needs ``w1``, ``b1``) even when the
    data-gradient half runs on the fp32 kernel. ``gate_pack``: a
    callable returning ``mlp_pack_w2_f16_gate(w2, w3)`` -- with it, single-output heads and two-output
    heads whose gradients are exact negatives (checked on ``dout``) run the data gradient in gate mode.
    ``h2=None`` (a forward with ``save_h2=False``: gate bits only) with the layer's own ``w2`` [256, 256] and
    ``b2``: such rank-one heads then take dW3 from the weight-gradient sums (``rl8_mlp_wgrad_gate_bits_f32``) and
    h2 is never touched; if the head turns out not to be rank-one, ``h2_fn()`` must supply h2 (a forward re-run).
    ``info`` (a dict) receives ``rank_one``: whether the gate kernels ran.  ``assume_general`` (with ``h2``): a
    two-output head already known not to be a pair skips the check of ``dout`` (a pass over it and a host read per
    call) and runs the general kernels, which are right for any ``dout``.  ``assume_pair``: ``dout`` [M, 2] is an exact
    pair by construction (the caller vouches: the two-class loss kernel's own output) -- no check either."""
    m, d_in = x.shape
    n_out = w3.shape[0]
    split = w2t_packed.dtype == torch.uint8
    f16 = split
    if split and w2t_packed.numel() != int(load().rl8_mlp_f16_packed_bytes()):
        raise ValueError("w2t_packed: a uint8 pack must come from mlp_pack_w2_f16(..., transposed=True)")
    if f16 and gate2 is None:
        raise ValueError("the fp16-plane backward needs gate2 (mlp_tower_forward_split(save_gate=True))")
    if h1 is None and not split:
        raise ValueError("h1 may be omitted only on the fp16-plane path")
    for name, t, numel in (("x", x, m * d_in), ("h1", h1, m * MLP_HIDDEN), ("h2", h2, m * MLP_HIDDEN),
                           ("dout", dout, m * n_out)):
        if t is None:
            continue
        _dense(t, torch.float32, name)
        if t.numel() != numel:
            raise ValueError(f"{name} has the wrong number of elements")
    if gate2 is not None and (gate2.dtype != torch.int32 or tuple(gate2.shape) != (m, 8) or not gate2.is_contiguous()):
        raise ValueError("gate2 must be the [M, 8] int32 gate of mlp_tower_forward_split(save_gate=True)")
    lib = load()
    width = int(lib.rl8_mlp_backward_partial_floats(d_in, n_out))
    max_rows = int(lib.rl8_mlp_backward_max_rows())
    partials = torch.empty(max_rows, width, dtype=torch.float32, device=x.device)
    rows = C.c_int(0)
    dw2 = None
    if split:
        # fused: the data-gradient kernel stores no dZ2; the weight-gradient kernel
        # re-forms it (and h1) and accumulates the head gradients
        if w1 is None or b1 is None:
            raise ValueError("the fp16-plane backward needs w1 and b1")
        w1p, b1p = _ptr(_dense(w1.detach(), torch.float32, "w1")), _ptr(_dense(b1.detach(), torch.float32, "b1"))
        # two outputs with exactly opposite gradients (a two-way categorical head): the weight
        # gradient can take the gate-plane kernel; checked on the data, the answer read back
        # behind the data-gradient launch so that the GPU has work while the host waits
        gates_on = not int(os.environ.get("RL8_WGRAD_GATE_OFF", "0") or 0)
        pair_flag, pair = None, False
        if n_out == 2 and gates_on and assume_pair:
            pair = True
        elif n_out == 2 and gates_on and not (assume_general and h2 is not None):
            pair_flag = _pair_flag(x.device)
            _check(lib.rl8_mlp_dout_pair_check(_ptr(dout), m, _ptr(pair_flag), _stream()), "rl8_mlp_dout_pair_check")
            if f16 and (gate_pack is not None or h2 is None):  # needed before the data gradient: read it now (a short bubble)
                pair = int(pair_flag[0].item()) == 0
                pair_flag = None
        if h2 is None and not (f16 and gates_on and (n_out == 1 or pair) and w2 is not None and b2 is not None):
            if h2_fn is None:
                raise ValueError("h2=None needs a rank-one head on the fp16-plane path (with w2, b2) or h2_fn")
            h2 = h2_fn()  # not a rank-one head after all: the forward is re-run for h2
        gate_dgrad = f16 and gate_pack is not None and gates_on and (n_out == 1 or pair)
        with _timed("mlp_tower_backward_gate" if gate_dgrad else "mlp_tower_backward", m):
            if gate_dgrad:
                _check(
                    lib.rl8_mlp_tower_backward_gate_f16_f32(
                        _ptr(x), w1p, b1p, _ptr(dout), m, d_in, _ptr(gate_pack()), n_out,
                        _ptr(partials), C.byref(rows), _ptr(gate2), _stream()),
                    "rl8_mlp_tower_backward_gate_f16_f32",
                )
            else:
                _check(
                    lib.rl8_mlp_tower_backward_f16_f32(
                        _ptr(x), w1p, b1p, _ptr(dout), m, d_in, _ptr(w2t_packed), _ptr(w3.detach()), n_out,
                        _ptr(partials), C.byref(rows), _ptr(gate2), _stream()),
                    "rl8_mlp_tower_backward_f16_f32",
                )
        dw2 = torch.empty(MLP_HIDDEN, MLP_HIDDEN, dtype=torch.float32, device=x.device)
        # (bf16 planes for both generations: see rl8_mlp_tower_backward_f16_f32; single-output towers
        # run the gate-plane kernel -- three plane products instead of six -- timed under its own name)
        gate_kernel = n_out == 1 and gates_on
        pair = pair or (pair_flag is not None and int(pair_flag[0].item()) == 0)
        if info is not None:
            info["rank_one"] = bool(gate_kernel or pair)
        with _timed("mlp_wgrad_gate" if gate_kernel or pair else "mlp_wgrad", m):
            if h2 is None:  # rank-one head, gate bits only: dW3 from the weight-gradient sums
                _check(
                    lib.rl8_mlp_wgrad_gate_bits_f32(
                        _ptr(gate2), _ptr(dout), _ptr(x), w1p, b1p, _ptr(_dense(w2.detach(), torch.float32, "w2")),
                        _ptr(_dense(b2.detach(), torch.float32, "b2")), _ptr(w3.detach()), m, d_in, n_out,
                        _ptr(_wgrad_workspace(x.device)), _ptr(dw2), _ptr(partials), _stream()),
                    "rl8_mlp_wgrad_gate_bits_f32",
                )
            elif pair:
                _check(
                    lib.rl8_mlp_wgrad_fused_pair_f32(
                        _ptr(h2), _ptr(dout), _ptr(x), w1p, b1p, _ptr(w3.detach()), m, d_in,
                        _ptr(_wgrad_workspace(x.device)), _ptr(dw2), _ptr(partials), _stream()),
                    "rl8_mlp_wgrad_fused_pair_f32",
                )
            else:
                _check(
                    lib.rl8_mlp_wgrad_fused_split_f32(
                        _ptr(h2), _ptr(dout), _ptr(x), w1p, b1p, _ptr(w3.detach()), m, d_in, n_out,
                        _ptr(_wgrad_workspace(x.device)), _ptr(dw2), _ptr(partials), _stream()),
                    "rl8_mlp_wgrad_fused_split_f32",
                )
    else:
        dz2 = torch.empty(m, MLP_HIDDEN, dtype=torch.float32, device=x.device)
        with _timed("mlp_tower_backward", m):
            _check(
                lib.rl8_mlp_tower_backward_f32(
                    _ptr(x), _ptr(h1), _ptr(h2), _ptr(dout), m, d_in, _ptr(w2t_packed), _ptr(w3.detach()), n_out,
                    _ptr(dz2), _ptr(partials), C.byref(rows), _stream()),
                "rl8_mlp_tower_backward_f32",
            )
    small = partials[: rows.value].sum(0)
    o1 = MLP_HIDDEN * d_in
    grads = {
        "w1": small[:o1].view(MLP_HIDDEN, d_in),
        "b1": small[o1 : o1 + MLP_HIDDEN],
        "b2": small[o1 + MLP_HIDDEN : o1 + 2 * MLP_HIDDEN],
        "w3": small[o1 + 2 * MLP_HIDDEN : o1 + 2 * MLP_HIDDEN + n_out * MLP_HIDDEN].view(n_out, MLP_HIDDEN),
        # (the fused plane backward leaves db3 -- a column sum of dout -- to the caller)
        "b3": _column_sums(dout) if split else small[o1 + 2 * MLP_HIDDEN + n_out * MLP_HIDDEN :],
        "w2": dw2 if split else (mlp_wgrad_split(dz2, x, w1, b1) if wgrad_split else mlp_wgrad(dz2, h1)),
    }
    return grads


_pair_flags: dict[tuple[int, int], torch.Tensor] = {}


def _pair_flag(device: torch.device) -> torch.Tensor:
    key = (device.index if device.index is not None else torch.cuda.current_device(), _stream())
    flag = _pair_flags.get(key)
    if flag is None:
        flag = _pair_flags[key] = torch.zeros(4, dtype=torch.int32, device=device)
    return flag


_wgrad_ws: dict[tuple[int, int], torch.Tensor] = {}


def _wgrad_workspace(device: torch.device) -> torch.Tensor:
    key = (device.index if device.index is not None else torch.cuda.current_device(), _stream())
    ws = _wgrad_ws.get(key)
    if ws is None:
        ws = torch.empty(int(load().rl8_mlp_wgrad_workspace_bytes()) // 4, dtype=torch.float32, device=device)
        ws[-64:].zero_()  # (the guard's lifetime counters live there: include/rl8_amd.h, rl8_mlp_wgrad_workspace_bytes)
        _wgrad_ws[key] = ws
    return ws


def wgrad_guard_counts() -> tuple[int, int]:
    """(weight-gradient calls that consulted the guard of the fp16 planes, calls it sent to the exact bf16 planes)
    over the lifetime of this process's workspaces; one host sync."""
    calls = fires = 0
    for ws in _wgrad_ws.values():
        words = ws[-64:].view(torch.int32)[32:34].tolist()
        calls, fires = calls + words[0], fires + words[1]
    return calls, fires


def mlp_wgrad_split(dz2: torch.Tensor, x: torch.Tensor, w1: torch.Tensor, b1: torch.Tensor) -> torch.Tensor:
    """dW2 [256, 256] = dz2^T @ relu(x @ w1^T + b1) over the rows: bf16-plane MFMAs at
    fp32 accuracy, h1 recomputed from the observations (deterministic)."""
    _dense(dz2, torch.float32, "dz2")
    _dense(x, torch.float32, "x")
    m, d_in = x.shape
    if tuple(dz2.shape) != (m, MLP_HIDDEN):
        raise ValueError("dz2 must be [M, 256] with M = x.shape[0]")
    out = torch.empty(MLP_HIDDEN, MLP_HIDDEN, dtype=torch.float32, device=dz2.device)
    with _timed("mlp_wgrad", m):
        _check(load().rl8_mlp_wgrad_split_f32(_ptr(dz2), _ptr(x), _ptr(_dense(w1.detach(), torch.float32, "w1")),
                                              _ptr(_dense(b1.detach(), torch.float32, "b1")), m, d_in,
                                              _ptr(_wgrad_workspace(dz2.device)), _ptr(out), 0, _stream()),
               "rl8_mlp_wgrad_split_f32")
    return out


def mlp_wgrad(dz2: torch.Tensor, h1: torch.Tensor) -> torch.Tensor:
    """dW2 [256, 256] = dz2^T @ h1 over the rows (fp32 MFMA, deterministic)."""
    _dense(dz2, torch.float32, "dz2")
    _dense(h1, torch.float32, "h1")
    if dz2.shape != h1.shape or dz2.shape[1] != MLP_HIDDEN:
        raise ValueError("dz2 and h1 must both be [M, 256]")
    lib = load()
    ws = _wgrad_workspace(dz2.device)
    out = torch.empty(MLP_HIDDEN, MLP_HIDDEN, dtype=torch.float32, device=dz2.device)
    with _timed("mlp_wgrad", dz2.shape[0]):
        _check(lib.rl8_mlp_wgrad_f32(_ptr(dz2), _ptr(h1), dz2.shape[0], _ptr(ws), _ptr(out), 0, _stream()),
               "rl8_mlp_wgrad_f32")
    return out


# --------------------------------------------------------------------------- #
# OPT-IN: towers of a scalar observation as piecewise-linear tables (nn/piecewise_mlp.py).
# --------------------------------------------------------------------------- #
def pw_max_breaks() -> int:
    return int(load().rl8_pw_max_breaks())


def pw_tower_forward(x: torch.Tensor, table: torch.Tensor, p: int, n_out: int) -> torch.Tensor:
    """x [M, 1] -> out [M, n_out] from ``table`` = [breaks p | anchor p + 1 | value (p + 1) n | slope (p + 1) n] (fp32)."""
    x = _dense(x.detach(), torch.float32, "x")
    _dense(table, torch.float32, "table")
    m = x.shape[0]
    if x.numel() != m or table.numel() != p + (p + 1) * (1 + 2 * n_out):
        raise ValueError("pw_tower_forward: x must be [M, 1] and the table p + (p + 1) (1 + 2 n_out) floats")
    out = torch.empty(m, n_out, dtype=torch.float32, device=x.device)
    with _timed("pw_tower_forward", m):
        _check(load().rl8_pw_tower_forward_f32(_ptr(x), m, _ptr(table), p, n_out, _ptr(out), _stream()),
               "rl8_pw_tower_forward_f32")
    return out


_pw_ws: dict[tuple, torch.Tensor] = {}


def pw_segment_sums(x: torch.Tensor, dout: torch.Tensor, breaks: torch.Tensor, p: int) -> torch.Tensor:
    """[(p + 1), 2, n_out] fp64: per interval the sums of dout and of dout * x over its rows (exact, order-independent)."""
    x = _dense(x.detach(), torch.float32, "x")
    dout = _dense(dout.detach(), torch.float32, "dout")
    m, n_out = dout.shape
    if x.numel() != m:
        raise ValueError("pw_segment_sums: x must be [M, 1] with M = dout.shape[0]")
    lib = load()
    key = (x.device.index if x.device.index is not None else torch.cuda.current_device(), _stream(), p, n_out)
    ws = _pw_ws.get(key)
    if ws is None:
        _pw_ws.clear()  # (one table shape at a time per process is the common case)
        ws = _pw_ws[key] = torch.empty(int(lib.rl8_pw_workspace_bytes(p, n_out)) // 8 + 2, dtype=torch.int64, device=x.device)
    sums = torch.empty(p + 1, 2, n_out, dtype=torch.float64, device=x.device)
    with _timed("pw_segment_sums", m):
        for lo in range(0, m, 1 << 25):  # (2^25 rows per call: the integer accumulators' headroom)
            hi = min(m, lo + (1 << 25))
            part = sums if lo == 0 else torch.empty_like(sums)
            _check(lib.rl8_pw_segment_sums_f32(_ptr(x[lo:hi]), _ptr(dout[lo:hi]), hi - lo, n_out, _ptr(breaks), p, _ptr(ws),
                                               _ptr(part), _stream()), "rl8_pw_segment_sums_f32")
            if lo:
                sums += part
    return sums


# --------------------------------------------------------------------------- #
# Fused LSTM (a-9): one layer, hidden 256, batch_first.
# --------------------------------------------------------------------------- #
LSTM_HIDDEN = 256


def lstm_supports(d_in: int) -> bool:
    return bool(load().rl8_lstm_supports(int(d_in)))


def lstm_pack(w_ih: torch.Tensor, w_hh: torch.Tensor, b_ih: torch.Tensor, b_hh: torch.Tensor) -> torch.Tensor:
    """torch.nn.LSTM parameters (``weight_ih_l0`` [1024, d], ``weight_hh_l0``
    [1024, 256], ``bias_ih_l0``, ``bias_hh_l0`` [1024]) -> forward weights in MFMA
    fragment order with the input projection and biases folded in."""
    d_in = w_ih.shape[1]
    for name, t, shape in (("w_ih", w_ih, (4 * LSTM_HIDDEN, d_in)), ("w_hh", w_hh, (4 * LSTM_HIDDEN, LSTM_HIDDEN)),
                           ("b_ih", b_ih, (4 * LSTM_HIDDEN,)), ("b_hh", b_hh, (4 * LSTM_HIDDEN,))):
        _dense(t.detach(), torch.float32, name)
        if tuple(t.shape) != shape:
            raise ValueError(f"{name} must have shape {shape}, got {tuple(t.shape)}")
    lib = load()
    packed = torch.empty(int(lib.rl8_lstm_pack_floats()), dtype=torch.float32, device=w_hh.device)
    _check(lib.rl8_lstm_pack_f32(_ptr(w_ih.detach()), _ptr(w_hh.detach()), _ptr(b_ih.detach()), _ptr(b_hh.detach()),
                                 d_in, _ptr(packed), _stream()), "rl8_lstm_pack_f32")
    return packed


def lstm_forward(x: torch.Tensor, h0: torch.Tensor, c0: torch.Tensor, w_packed: torch.Tensor, *, save: bool = False):
    """x [B, L, d], h0 / c0 [B, 256] -> (hs [B, L, 256], hn, cn [B, 256], gates, cs);
    ``gates`` [B, L, 4, 256] and ``cs`` [B, L, 256] only with ``save``."""
    x = _dense(x.detach(), torch.float32, "x")
    b, l, d_in = x.shape
    for name, t in (("h0", h0), ("c0", c0)):
        _dense(t, torch.float32, name)
        if tuple(t.shape) != (b, LSTM_HIDDEN):
            raise ValueError(f"{name} must be [{b}, {LSTM_HIDDEN}], got {tuple(t.shape)}")
    dev = x.device
    hs = torch.empty(b, l, LSTM_HIDDEN, dtype=torch.float32, device=dev)
    hn = torch.empty(b, LSTM_HIDDEN, dtype=torch.float32, device=dev)
    cn = torch.empty(b, LSTM_HIDDEN, dtype=torch.float32, device=dev)
    gates = torch.empty(b, l, 4, LSTM_HIDDEN, dtype=torch.float32, device=dev) if save else None
    cs = torch.empty(b, l, LSTM_HIDDEN, dtype=torch.float32, device=dev) if save else None
    with _timed("lstm_forward_save" if save else "lstm_forward", b * l):
        _check(load().rl8_lstm_forward_f32(_ptr(x), b, l, d_in, _ptr(h0), _ptr(c0), _ptr(w_packed), _ptr(hs), _ptr(hn),
                                           _ptr(cn), _ptr(gates), _ptr(cs), _stream()), "rl8_lstm_forward_f32")
    return hs, hn, cn, gates, cs


def lstm_split_supports(d_in: int) -> bool:
    return bool(load().rl8_lstm_split_supports(int(d_in)))


def lstm_pack_split(w_ih: torch.Tensor, w_hh: torch.Tensor, b_ih: torch.Tensor, b_hh: torch.Tensor):
    """torch.nn.LSTM parameters -> (W_hh as bf16 planes in the step kernel's fragment order,
    ``wb`` [1024, 8] = [w_ih | 0.. | b_ih + b_hh])."""
    d_in = w_ih.shape[1]
    for name, t, shape in (("w_ih", w_ih, (4 * LSTM_HIDDEN, d_in)), ("w_hh", w_hh, (4 * LSTM_HIDDEN, LSTM_HIDDEN)),
                           ("b_ih", b_ih, (4 * LSTM_HIDDEN,)), ("b_hh", b_hh, (4 * LSTM_HIDDEN,))):
        _dense(t.detach(), torch.float32, name)
        if tuple(t.shape) != shape:
            raise ValueError(f"{name} must have shape {shape}, got {tuple(t.shape)}")
    lib = load()
    packed = torch.empty(int(lib.rl8_lstm_split_packed_bytes()), dtype=torch.uint8, device=w_hh.device)
    wb = torch.empty(int(lib.rl8_lstm_split_wb_floats()), dtype=torch.float32, device=w_hh.device)
    _check(lib.rl8_lstm_pack_split(_ptr(w_ih.detach()), _ptr(w_hh.detach()), _ptr(b_ih.detach()), _ptr(b_hh.detach()),
                                   d_in, _ptr(packed), _ptr(wb), _stream()), "rl8_lstm_pack_split")
    return packed, wb


def lstm_state_planes(rows: int, device: torch.device | str, copies: int = 1) -> torch.Tensor:
    """``copies`` buffers (whole 128-row tiles each) for the bf16 planes of ``rows`` rows of
    hidden state; a multi-step forward ping-pongs between two."""
    return torch.empty(copies * int(load().rl8_lstm_split_state_bytes(int(rows))), dtype=torch.uint8, device=device)


def lstm_split_state(h: torch.Tensor, *, out: None | torch.Tensor = None,
                     bound_out: None | torch.Tensor = None) -> torch.Tensor:
    """``h`` [B, 256] -> its fp16 planes in the step kernel's operand order (``out`` or a fresh buffer);
    ``bound_out`` (one float32 element) receives max |h|."""
    h = _dense(h, torch.float32, "h")
    b, dev, lib = h.shape[0], h.device, load()
    if h.ndim != 2 or h.shape[1] != LSTM_HIDDEN:
        raise ValueError("lstm_split_state: h must be [B, 256]")
    if out is None:
        out = lstm_state_planes(b, dev)
    elif out.dtype != torch.uint8 or out.device != dev or out.numel() < int(lib.rl8_lstm_split_state_bytes(b)):
        raise ValueError("lstm_split_state: `out` must hold the planes of b rows")
    if bound_out is not None:
        if bound_out.dtype != torch.float32 or bound_out.numel() != 1 or bound_out.device != dev:
            raise ValueError("bound_out must be one float32 element on h's device")
        _check(lib.rl8_lstm_split_state_bound(_ptr(h), LSTM_HIDDEN, b, _ptr(out), _ptr(bound_out), _stream()),
               "rl8_lstm_split_state_bound")
    else:
        _check(lib.rl8_lstm_split_state(_ptr(h), LSTM_HIDDEN, b, _ptr(out), _stream()), "rl8_lstm_split_state")
    return out


def lstm_forward_split(x: torch.Tensor, h0: torch.Tensor, c0: torch.Tensor, packed: torch.Tensor, wb: torch.Tensor,
                       *, save: bool = False, planes: None | torch.Tensor = None,
                       h0_bound_out: None | torch.Tensor = None, h0_planes: None | torch.Tensor = None):
    """As :func:`lstm_forward` on the fp16-plane step kernel: one state split + one step
    launch per timestep. Same outputs and saved layouts (``gates`` [B, L, 4, 256], ``cs``).
    ``h0_bound_out`` (one float32 element): receives max |h0|, which the state split sees
    anyway (:func:`lstm_backward`'s ``h0_bound``). ``h0_planes``: the planes of ``h0`` from an
    earlier :func:`lstm_split_state` (read only; no split is made here)."""
    x = _dense(x.detach(), torch.float32, "x")
    b, l, d_in = x.shape
    for name, t in (("h0", h0), ("c0", c0)):
        _dense(t, torch.float32, name)
        if tuple(t.shape) != (b, LSTM_HIDDEN):
            raise ValueError(f"{name} must be [{b}, {LSTM_HIDDEN}], got {tuple(t.shape)}")
    dev = x.device
    lib = load()
    hs = torch.empty(b, l, LSTM_HIDDEN, dtype=torch.float32, device=dev)
    cs = torch.empty(b, l, LSTM_HIDDEN, dtype=torch.float32, device=dev)
    gates = torch.empty(b, l, 4, LSTM_HIDDEN, dtype=torch.float32, device=dev) if save else None
    if planes is None:
        planes = lstm_state_planes(b, dev, copies=2 if l > 1 else 1)
    # two plane buffers: a step reads h_{t-1}'s planes from one and leaves h_t's in the other
    half = planes.numel() // 2 if l > 1 else 0
    if l > 1 and half < int(lib.rl8_lstm_split_state_bytes(b)):
        raise ValueError("lstm_forward_split: `planes` must hold two state-plane buffers for L > 1")
    H, stream = LSTM_HIDDEN, _stream()
    xp, hsp, csp, gp, pp = _ptr(x), _ptr(hs), _ptr(cs), _ptr(gates), _ptr(planes)
    if h0_planes is not None:
        if h0_bound_out is not None:
            raise ValueError("lstm_forward_split: h0_bound_out comes from the split that made h0_planes")
        if h0_planes.dtype != torch.uint8 or h0_planes.device != dev or h0_planes.numel() < int(lib.rl8_lstm_split_state_bytes(b)):
            raise ValueError("lstm_forward_split: h0_planes must hold the planes of b rows")
    else:
        lstm_split_state(h0, out=planes, bound_out=h0_bound_out)
    for t in range(l):
        c_prev, c_pitch = (_ptr(c0), H) if t == 0 else (csp + (t - 1) * H * 4, l * H)
        p_in, p_out = pp + (t & 1) * half, (pp + ((t + 1) & 1) * half) if t + 1 < l else None
        if t == 0 and h0_planes is not None:
            p_in = _ptr(h0_planes)
        with _timed("lstm_step_save" if save else "lstm_step", b):
            _check(lib.rl8_lstm_step_split_f32(
                xp + t * d_in * 4, l * d_in, d_in, p_in, c_prev, c_pitch, _ptr(packed), _ptr(wb), b,
                hsp + t * H * 4, l * H, csp + t * H * 4, l * H, (gp + t * 4 * H * 4) if save else None, l * 4 * H,
                p_out, stream), "rl8_lstm_step_split_f32")
    hn, cn = hs[:, l - 1], cs[:, l - 1]
    return hs, hn, cn, gates, (cs if save else None)


def lstm_pack_transposed(w_hh: torch.Tensor) -> torch.Tensor:
    """``weight_hh_l0`` [1024, 256] -> the four gate blocks packed for the backward
    data-gradient product (``rl8_mlp_pack_w2_f32(..., transposed=1)`` per gate)."""
    w_hh = _dense(w_hh.detach(), torch.float32, "w_hh")
    if tuple(w_hh.shape) != (4 * LSTM_HIDDEN, LSTM_HIDDEN):
        raise ValueError("w_hh must be [1024, 256]")
    return torch.cat([mlp_pack_w2(w_hh[LSTM_HIDDEN * q : LSTM_HIDDEN * (q + 1)], transposed=True) for q in range(4)])


def lstm_rows_backward_pack(w_hh: torch.Tensor) -> torch.Tensor:
    """``w_hh`` [1024, 256] -> the bf16 planes of ``W_hh^T`` in the order
    ``rl8_lstm_rows_backward_f32`` streams them."""
    w_hh = _dense(w_hh.detach(), torch.float32, "w_hh")
    if tuple(w_hh.shape) != (4 * LSTM_HIDDEN, LSTM_HIDDEN):
        raise ValueError("w_hh must be [1024, 256]")
    lib = load()
    packed = torch.empty(int(lib.rl8_lstm_rows_backward_pack_bytes()), dtype=torch.uint8, device=w_hh.device)
    _check(lib.rl8_lstm_rows_backward_pack(_ptr(w_hh), _ptr(packed), _stream()), "rl8_lstm_rows_backward_pack")
    return packed


ROWS_BACKWARD_HEADS = 4  # head outputs rl8_lstm_rows_backward_heads_f32 takes


def lstm_rows_backward(c0: torch.Tensor, gates: torch.Tensor, cs: torch.Tensor, dhs: None | torch.Tensor,
                       packed: torch.Tensor, *, with_bound: bool = False,
                       heads: None | tuple[torch.Tensor, torch.Tensor] = None):
    """dgates [B, L, 4, 256] from what the forward saved and dhs [B, L, 256]: the
    backward through time with the recurrent product on bf16 planes. ``with_bound``:
    also a one-element device tensor holding max |dgates| (what the fp16-plane weight
    gradient scales its first operand by). ``heads`` = (dout [B * L, n], w [n, 256]),
    n <= 4, instead of ``dhs`` (None): dL/dh_t = dout x w is formed inside the kernel."""
    b, l = gates.shape[0], gates.shape[1]
    checks = [("c0", c0, (b, LSTM_HIDDEN)), ("gates", gates, (b, l, 4, LSTM_HIDDEN)), ("cs", cs, (b, l, LSTM_HIDDEN))]
    if heads is None:
        checks.append(("dhs", dhs, (b, l, LSTM_HIDDEN)))
    elif dhs is not None:
        raise ValueError("lstm_rows_backward: either dhs or heads")
    for name, t, shape in checks:
        _dense(t, torch.float32, name)
        if tuple(t.shape) != shape:
            raise ValueError(f"{name} must have shape {shape}, got {tuple(t.shape)}")
    dev = gates.device
    dgates = torch.empty(b, l, 4, LSTM_HIDDEN, dtype=torch.float32, device=dev)
    dc = torch.empty(b, LSTM_HIDDEN, dtype=torch.float32, device=dev)
    bound = torch.empty(1, dtype=torch.float32, device=dev) if with_bound else None
    if heads is not None:
        dout, w = heads
        n = w.shape[0]
        if n > ROWS_BACKWARD_HEADS or tuple(w.shape) != (n, LSTM_HIDDEN) or dout.numel() != b * l * n:
            raise ValueError(f"heads: dout [B * L, n], w [n, 256], n <= {ROWS_BACKWARD_HEADS}")
        # four floats per row-step, four weight rows: one 16-byte piece per sequence and head-row segment
        dout4 = torch.zeros(b * l, ROWS_BACKWARD_HEADS, dtype=torch.float32, device=dev)
        dout4[:, :n] = dout.reshape(b * l, n)
        w4 = torch.zeros(ROWS_BACKWARD_HEADS, LSTM_HIDDEN, dtype=torch.float32, device=dev)
        w4[:n] = w.detach()
        with _timed("lstm_rows_backward", b * l):
            _check(load().rl8_lstm_rows_backward_heads_f32(b, l, _ptr(c0), _ptr(gates), _ptr(cs), _ptr(dout4), _ptr(w4),
                                                           _ptr(packed), _ptr(dgates), _ptr(dc), _ptr(bound), _stream()),
                   "rl8_lstm_rows_backward_heads_f32")
        return (dgates, bound) if with_bound else dgates
    with _timed("lstm_rows_backward", b * l):
        _check(load().rl8_lstm_rows_backward_f32(b, l, _ptr(c0), _ptr(gates), _ptr(cs), _ptr(dhs), _ptr(packed),
                                                 _ptr(dgates), _ptr(dc), _ptr(bound), _stream()), "rl8_lstm_rows_backward_f32")
    return (dgates, bound) if with_bound else dgates


#: column-sum partial rows of the fused LSTM weight gradient, per (device, stream, L, d_in)
_lstm_colsum_ws: dict[tuple, torch.Tensor] = {}


def lstm_backward(
    x: torch.Tensor, h0: torch.Tensor, c0: torch.Tensor, hs: torch.Tensor, gates: torch.Tensor, cs: torch.Tensor,
    dhs: None | torch.Tensor, whht_packed: None | torch.Tensor, *, split: None | bool = None,
    rows_packed: None | torch.Tensor = None, h0_bound: None | torch.Tensor = None,
    heads: None | tuple[torch.Tensor, torch.Tensor] = None, hs_bound: None | float = None,
) -> dict[str, torch.Tensor]:
    """Parameter gradients of the LSTM given ``dhs`` [B, L, 256] (gradient of every
    ``h_t``) and what ``lstm_forward(..., save=True)`` returned. Returns ``w_ih``,
    ``w_hh``, ``b`` (the gradient of each of the two bias vectors). ``split``: the
    weight-gradient GEMMs on bf16 planes (True) or the fp32 MFMA (False); the caller
    passes the mode its forward ran in (``fused_lstm.use_split``), None reads
    ``RL8_AMD_LSTM_GEMM`` as the forward's module switch does at import.
    ``rows_packed`` (:func:`lstm_rows_backward_pack`): the backward through time runs
    on bf16 planes too (``rl8_lstm_rows_backward_f32``) instead of the fp32-MFMA kernel
    that reads ``whht_packed``; only with ``split`` (its dW_ih / bias sums come from the
    weight-gradient kernel). ``h0_bound``: one float32 element >= max |h0| when the caller
    has it (:func:`lstm_forward_split`'s ``h0_bound_out``); computed here otherwise.
    ``heads`` (with ``rows_packed``, ``dhs`` None): see :func:`lstm_rows_backward`.
    ``hs_bound``: a number >= max |hs| the caller vouches for -- 1.0 when ``hs`` is this LSTM's own output
    (|o * tanh(c)| < 1), which is what ``nn.fused_lstm`` passes; None: taken from ``hs`` here (one reduction over it).
    The fp16 planes of the weight gradient are scaled by it: a value above the bound would overflow them."""
    x = _dense(x.detach(), torch.float32, "x")
    b, l, d_in = x.shape
    if heads is not None and (rows_packed is None or dhs is not None):
        raise ValueError("lstm_backward: heads needs rows_packed and no dhs")
    for name, t, shape in (("h0", h0, (b, LSTM_HIDDEN)), ("c0", c0, (b, LSTM_HIDDEN)), ("hs", hs, (b, l, LSTM_HIDDEN)),
                           ("gates", gates, (b, l, 4, LSTM_HIDDEN)), ("cs", cs, (b, l, LSTM_HIDDEN)),
                           *([("dhs", dhs, (b, l, LSTM_HIDDEN))] if heads is None else [])):
        _dense(t, torch.float32, name)
        if tuple(t.shape) != shape:
            raise ValueError(f"{name} must have shape {shape}, got {tuple(t.shape)}")
    lib = load()
    dev = x.device
    H = LSTM_HIDDEN
    if split is None:
        split = os.environ.get("RL8_AMD_LSTM_GEMM", "split") == "split"
    # With the bf16-plane weight-gradient kernel and a compiled input width, dW_ih and the
    # bias gradient come out of that kernel as column sums of the dG it reads anyway; else
    # the backward call makes one more pass over dG for them.
    fused_colsums = split and lstm_split_supports(d_in)
    rows = C.c_int(0)
    partials = None
    dg_bound = None
    if rows_packed is not None:
        if not fused_colsums:
            raise ValueError("rows_packed needs the bf16-plane weight gradient (split) and a compiled input width")
        dgates, dg_bound = lstm_rows_backward(c0, gates, cs, dhs, rows_packed, with_bound=True, heads=heads)
    else:
        dgates = torch.empty(b, l, 4, H, dtype=torch.float32, device=dev)
        if not fused_colsums:
            width = int(lib.rl8_lstm_backward_partial_floats(d_in))
            partials = torch.empty(int(lib.rl8_lstm_backward_max_rows()), width, dtype=torch.float32, device=dev)
        with _timed("lstm_backward", b * l):
            _check(lib.rl8_lstm_backward_f32(_ptr(x), b, l, d_in, _ptr(c0), _ptr(gates), _ptr(cs), _ptr(dhs),
                                             _ptr(whht_packed), _ptr(dgates), _ptr(partials), C.byref(rows), _stream()),
                   "rl8_lstm_backward_f32")
    m = b * l
    ws = _wgrad_workspace(dev)
    dw_hh = torch.empty(4 * H, H, dtype=torch.float32, device=dev)
    # per gate and timestep dW_hh[q] += dG_q^T h_{t-1} over the B rows of that step: h_{t-1} is
    # h0 for t = 0 and hs[:, t-1] (row pitch L*256) after, so no shifted copy of hs is made
    dgp, hsp, h0p, wsp, dwp, stream = _ptr(dgates), _ptr(hs), _ptr(h0), _ptr(ws), _ptr(dw_hh), _stream()
    if fused_colsums:
        xt = [x[:, t].contiguous() for t in range(l)]            # dense [B][d] per step (the kernel's scalar loads)
        ckey = (dev.index or 0, _stream() or 0, l, d_in)
        cols = _lstm_colsum_ws.get(ckey)                          # [gate][step][workgroup][...], kept between calls
        if cols is None:
            cols = _lstm_colsum_ws[ckey] = torch.empty(4, l, 256, H * (d_in + 1), dtype=torch.float32, device=dev)
        crow = C.c_int(0)
        # with the backward kernel's bound on |dG| the products run on fp16 planes (three instead of six): dG scaled by
        # one power of two for the tensor, h_{t-1} by one from max |h0| (t = 0) or 1 (an LSTM's own outputs)
        f16 = dg_bound is not None and os.environ.get("RL8_AMD_LSTM_WGRAD_PLANES", "f16") != "bf16"
        if f16:
            if h0_bound is None:
                h0_bound = torch.linalg.vector_norm(h0, ord=float("inf")).reshape(1)
            elif h0_bound.dtype != torch.float32 or h0_bound.numel() != 1 or h0_bound.device != dev:
                raise ValueError("h0_bound must be one float32 element on x's device")
            if hs_bound is None:  # (one pass over hs: callers that know their hs say so)
                one = torch.linalg.vector_norm(hs, ord=float("inf")).reshape(1) if l > 1 else torch.ones(1, dtype=torch.float32, device=dev)
            else:
                one = torch.full((1,), float(hs_bound), dtype=torch.float32, device=dev)
        if f16 and b >= 128 and os.environ.get("RL8_AMD_LSTM_WGRAD_GATES", "fused") != "separate":
            # the four gates of a timestep in one launch: h_{t-1} comes out of HBM once instead of four times
            gkey = (dev.index or 0, _stream() or 0, l, d_in, "gates")
            gcols = _lstm_colsum_ws.get(gkey)                     # [step][gate][workgroup of the gate][...]
            if gcols is None:
                gcols = _lstm_colsum_ws[gkey] = torch.empty(l, 4, 64, H * (d_in + 1), dtype=torch.float32, device=dev)
            with _timed("lstm_wgrad", m):
                for t in range(l):
                    h_prev, h_pitch = (h0p, H) if t == 0 else (hsp + (t - 1) * H * 4, l * H)
                    _check(lib.rl8_lstm_wgrad_f16_f32(
                        dgp + t * 4 * H * 4, l * 4 * H, _ptr(dg_bound), h_prev, h_pitch, _ptr(h0_bound if t == 0 else one), b,
                        wsp, dwp, int(t > 0), _ptr(xt[t]), d_in, _ptr(gcols[t]), C.byref(crow), stream), "rl8_lstm_wgrad_f16_f32")
            # (a step's rows sit gate-major, crow.value per gate -- 64 at the sizes that fill the chip)
            width = H * (d_in + 1)
            sums = gcols.view(l, -1)[:, : 4 * crow.value * width].reshape(l, 4, crow.value, width).sum(dim=(0, 2))
            return {"w_ih": sums[:, : H * d_in].reshape(4 * H, d_in), "w_hh": dw_hh, "b": sums[:, H * d_in :].reshape(4 * H)}
        with _timed("lstm_wgrad", m):
            for q in range(4):
                for t in range(l):
                    h_prev, h_pitch = (h0p, H) if t == 0 else (hsp + (t - 1) * H * 4, l * H)
                    if f16:
                        _check(lib.rl8_mlp_wgrad_f16_strided_f32(
                            dgp + (t * 4 * H + q * H) * 4, l * 4 * H, _ptr(dg_bound), h_prev, h_pitch,
                            _ptr(h0_bound if t == 0 else one), b, wsp, dwp + 4 * H * H * q, int(t > 0),
                            _ptr(xt[t]), d_in, _ptr(cols[q, t]), C.byref(crow), stream), "rl8_mlp_wgrad_f16_strided_f32")
                        continue
                    _check(lib.rl8_mlp_wgrad_split_strided_f32(
                        dgp + (t * 4 * H + q * H) * 4, l * 4 * H, h_prev, h_pitch, b, wsp, dwp + 4 * H * H * q, int(t > 0),
                        _ptr(xt[t]), d_in, _ptr(cols[q, t]), C.byref(crow), stream), "rl8_mlp_wgrad_split_strided_f32")
        sums = cols[:, :, : crow.value].sum(dim=(1, 2))          # [4][256*(d+1)], steps then workgroups in order
        dw_ih = sums[:, : H * d_in].reshape(4 * H, d_in)
        db = sums[:, H * d_in :].reshape(4 * H)
        return {"w_ih": dw_ih, "w_hh": dw_hh, "b": db}
    small = partials[: rows.value].sum(0)
    fn, name = ((lib.rl8_mlp_wgrad_split_strided_f32, "rl8_mlp_wgrad_split_strided_f32") if split
                else (lib.rl8_mlp_wgrad_strided_f32, "rl8_mlp_wgrad_strided_f32"))
    extra = (None, 0, None, None) if split else ()
    with _timed("lstm_wgrad", m):
        for q in range(4):
            for t in range(l):
                h_prev, h_pitch = (h0p, H) if t == 0 else (hsp + (t - 1) * H * 4, l * H)
                _check(fn(dgp + (t * 4 * H + q * H) * 4, l * 4 * H, h_prev, h_pitch, b, wsp, dwp + 4 * H * H * q,
                          int(t > 0), *extra, stream), name)
    return {"w_ih": small[: 4 * LSTM_HIDDEN * d_in].view(4 * LSTM_HIDDEN, d_in), "w_hh": dw_hh,
            "b": small[4 * LSTM_HIDDEN * d_in :]}


HEADS_MAX_OUT = 8  # outputs of one linear-heads launch


def linear_heads_forward_pair(h: torch.Tensor, w_a: torch.Tensor, b_a: torch.Tensor, w_b: torch.Tensor,
                              b_b: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
    """Two layers on the same rows in one pass: (h x w_a^T + b_a [M, n_a], h x w_b^T + b_b [M, n_b])."""
    h = _dense(h.detach(), torch.float32, "h")
    w_a, b_a, w_b, b_b = (_dense(t.detach(), torch.float32, n) for t, n in ((w_a, "w_a"), (b_a, "b_a"), (w_b, "w_b"), (b_b, "b_b")))
    m, n_a, n_b = h.shape[0], w_a.shape[0], w_b.shape[0]
    if h.shape[1] != LSTM_HIDDEN or w_a.shape[1] != LSTM_HIDDEN or w_b.shape[1] != LSTM_HIDDEN or b_a.numel() != n_a \
            or b_b.numel() != n_b:
        raise ValueError("linear_heads_forward_pair: h [M, 256], w [n, 256], b [n]")
    out_a = torch.empty(m, n_a, dtype=torch.float32, device=h.device)
    out_b = torch.empty(m, n_b, dtype=torch.float32, device=h.device)
    with _timed("linear_heads_forward", m):
        _check(load().rl8_linear_heads_forward_pair_f32(_ptr(h), m, _ptr(w_a), _ptr(b_a), n_a, _ptr(out_a), _ptr(w_b),
                                                        _ptr(b_b), n_b, _ptr(out_b), _stream()),
               "rl8_linear_heads_forward_pair_f32")
    return out_a, out_b


def linear_heads_forward(h: torch.Tensor, w: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """h [M, 256] x w [n, 256]^T + b [n] -> [M, n] (n <= 8)."""
    h = _dense(h.detach(), torch.float32, "h")
    w = _dense(w.detach(), torch.float32, "w")
    b = _dense(b.detach(), torch.float32, "b")
    m, n = h.shape[0], w.shape[0]
    if h.shape[1] != LSTM_HIDDEN or w.shape[1] != LSTM_HIDDEN or b.numel() != n:
        raise ValueError("linear_heads_forward: h [M, 256], w [n, 256], b [n]")
    out = torch.empty(m, n, dtype=torch.float32, device=h.device)
    with _timed("linear_heads_forward", m):
        _check(load().rl8_linear_heads_forward_f32(_ptr(h), m, _ptr(w), _ptr(b), n, _ptr(out), _stream()),
               "rl8_linear_heads_forward_f32")
    return out


def linear_heads_backward(h: torch.Tensor, dout: torch.Tensor, w: torch.Tensor, *,
                          need_dh: bool = True) -> tuple[None | torch.Tensor, torch.Tensor, torch.Tensor]:
    """-> (dh [M, 256], dw [n, 256], db [n]); ``need_dh=False``: dh is None and never written."""
    h = _dense(h.detach(), torch.float32, "h")
    dout = _dense(dout, torch.float32, "dout")
    w = _dense(w.detach(), torch.float32, "w")
    m, n = h.shape[0], w.shape[0]
    if tuple(dout.shape) != (m, n):
        raise ValueError("linear_heads_backward: dout must be [M, n]")
    lib = load()
    dh = torch.empty(m, LSTM_HIDDEN, dtype=torch.float32, device=h.device) if need_dh else None
    partials = torch.empty(int(lib.rl8_linear_heads_max_rows()), n * LSTM_HIDDEN + n, dtype=torch.float32, device=h.device)
    rows = C.c_int(0)
    with _timed("linear_heads_backward", m):
        _check(lib.rl8_linear_heads_backward_f32(_ptr(h), _ptr(dout), m, _ptr(w), n, _ptr(dh), _ptr(partials),
                                                 C.byref(rows), _stream()), "rl8_linear_heads_backward_f32")
    small = partials[: rows.value].sum(0)
    return dh, small[: n * LSTM_HIDDEN].view(n, LSTM_HIDDEN), small[n * LSTM_HIDDEN :]
