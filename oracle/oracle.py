"""ctypes front-end of the CPU oracle (``rl8_oracle.c``). TEST INFRASTRUCTURE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module, and only as the checker / reported baseline. The
product package ``rl8_amd`` never imports it.

All functions take and return numpy arrays laid out as the reference lays out
its tensors (env-major ``[N, H+1, 1]`` buffers, ``[M, A, K]`` logits, ...).

"""

from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# RL8_ORACLE_LIB: another build of the same source (tests/test_oracle_sanitizers.py points it at the ASan + UBSan one)
_LIB_PATH = os.environ.get("RL8_ORACLE_LIB") or os.path.join(_HERE, "_build", "librl8_oracle.so")
_lib: None | C.CDLL = None


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (a few seconds); returns the library path."""
    src = os.path.join(_HERE, "rl8_oracle.c")
    hdr = os.path.join(_HERE, "..", "include", "rl8_philox.h")
    if os.environ.get("RL8_ORACLE_LIB"):
        return _LIB_PATH  # (built by whoever set it)
    stale = not os.path.exists(_LIB_PATH) or any(
        os.path.exists(p) and os.path.getmtime(p) > os.path.getmtime(_LIB_PATH)
        for p in (src, hdr)
    )
    if force or stale:
        subprocess.run(["make", "-C", _HERE, "-s", "-B"], check=True)
    return _LIB_PATH


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


class PPOHparams(C.Structure):
    _fields_ = [
        ("clip_param", C.c_float),
        ("dual_clip_param", C.c_float),
        ("entropy_coeff", C.c_float),
        ("vf_clip_param", C.c_float),
        ("vf_coeff", C.c_float),
        ("loss_scale", C.c_float),
    ]


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.oracle_exponential.restype = C.c_float
        _lib.oracle_exponential.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32]
        _lib.oracle_normal.restype = C.c_float
        _lib.oracle_normal.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32]
    return _lib


def _p(a: None | np.ndarray):
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"], "oracle arrays must be C-contiguous"
    return a.ctypes.data_as(C.c_void_p)


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a), dtype=np.float32)


def _i64(a) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a), dtype=np.int64)


def ppo_hparams(
    *,
    clip_param: float = 0.2,
    dual_clip_param: None | float = None,
    entropy_coeff: float = 0.0,
    vf_clip_param: float = 5.0,
    vf_coeff: float = 1.0,
    grad_accumulation_steps: int = 1,
) -> PPOHparams:
    return PPOHparams(
        clip_param,
        dual_clip_param if dual_clip_param else 0.0,
        entropy_coeff,
        vf_clip_param,
        vf_coeff,
        1.0 / grad_accumulation_steps,
    )


# --------------------------------------------------------------------------- #
# Environments.
# --------------------------------------------------------------------------- #
def dummy_env_step(state: np.ndarray, action: np.ndarray) -> tuple[np.ndarray, np.ndarray]:
    """Returns (new_state, reward); ``action`` int64 => discrete, float32 => continuous."""
    state = _f32(state).copy()
    reward = np.empty_like(state)
    n = state.size
    if np.issubdtype(np.asarray(action).dtype, np.integer):
        lib().oracle_dummy_env_step_discrete(_p(state), _p(_i64(action)), _p(reward), C.c_int64(n))
    else:
        lib().oracle_dummy_env_step_continuous(_p(state), _p(_f32(action)), _p(reward), C.c_int64(n))
    return state, reward


def dummy_env_reset(n: int, bounds: float, seed: int, reset_count: int, env_offset: int = 0) -> np.ndarray:
    state = np.empty((n, 1), np.float32)
    lib().oracle_dummy_env_reset(
        _p(state), C.c_int64(n), C.c_float(bounds), C.c_uint64(seed), C.c_uint64(reset_count), C.c_int64(env_offset)
    )
    return state


def cartpole_cfg(
    *,
    force_mag: float = 5.0,
    gravity: float = 9.8,
    length: float = 0.5,
    pole_mass: float = 0.1,
    pole_mass_length: float = 0.05,
    total_mass: float = 1.1,
    tau: float = 0.02,
    kinematics_integrator: str = "euler",
    **_,
) -> CartPoleCfg:
    return CartPoleCfg(
        force_mag, gravity, length, pole_mass, pole_mass_length, total_mass, tau,
        0 if kinematics_integrator == "euler" else 1,
    )


def cartpole_step(state: np.ndarray, action: np.ndarray, cfg: CartPoleCfg):
    """state [4, n] -> (state', obs [n, 5], reward [n, 1])."""
    state = _f32(state).copy()
    n = state.shape[1]
    obs = np.empty((n, 5), np.float32)
    reward = np.empty((n, 1), np.float32)
    lib().oracle_cartpole_step(_p(state), _p(_i64(action)), C.byref(cfg), _p(obs), _p(reward), C.c_int64(n))
    return state, obs, reward


def cartpole_reset(n: int, std: float, seed: int, reset_count: int, env_offset: int = 0) -> np.ndarray:
    state = np.empty((4, n), np.float32)
    lib().oracle_cartpole_reset(
        _p(state), C.c_int64(n), C.c_float(std), C.c_uint64(seed), C.c_uint64(reset_count), C.c_int64(env_offset)
    )
    return state


class MountainCarCfg(C.Structure):
    _fields_ = [(k, C.c_float) for k in ("force_mag", "goal_position", "goal_velocity", "gravity", "max_position",
                                         "max_speed", "min_position")]


def mountain_car_cfg(*, force_mag=0.001, goal_position=0.5, goal_velocity=0.0, gravity=0.0025, max_position=0.6,
                     max_speed=0.07, min_position=-1.2) -> MountainCarCfg:
    return MountainCarCfg(force_mag, goal_position, goal_velocity, gravity, max_position, max_speed, min_position)


def mountain_car_step(state: np.ndarray, action: np.ndarray, cfg: MountainCarCfg):
    """state [2, n] -> (state', obs [n, 2], reward [n, 1])  (examples/mountain_car/env.py:12-38)."""
    state = _f32(state).copy()
    n = state.shape[1]
    obs = np.empty((n, 2), np.float32)
    reward = np.empty((n, 1), np.float32)
    lib().oracle_mountain_car_step(_p(state), _p(_i64(action)), C.byref(cfg), _p(obs), _p(reward), C.c_int64(n))
    return state, obs, reward


def mountain_car_reset(n: int, seed: int, reset_count: int, env_offset: int = 0) -> np.ndarray:
    state = np.empty((2, n), np.float32)
    lib().oracle_mountain_car_reset(_p(state), C.c_int64(n), C.c_uint64(seed), C.c_uint64(reset_count),
                                    C.c_int64(env_offset))
    return state


class PendulumCfg(C.Structure):
    _fields_ = [(k, C.c_float) for k in ("dt", "gravity_coeff", "torque_coeff", "max_speed", "max_torque")]


def pendulum_cfg(*, dt=0.05, g=10.0, l=1.0, m=1.0, max_speed=8.0, max_torque=2.0) -> PendulumCfg:  # noqa: E741
    return PendulumCfg(dt, 3 * g / (2 * l), 3.0 / (m * l**2), max_speed, max_torque)


def pendulum_step(state: np.ndarray, action: np.ndarray, cfg: PendulumCfg):
    """state [2, n], action [n, 1] -> (state', obs [n, 3], reward [n, 1])  (examples/pendulum/env.py:12-39)."""
    state = _f32(state).copy()
    n = state.shape[1]
    obs = np.empty((n, 3), np.float32)
    reward = np.empty((n, 1), np.float32)
    lib().oracle_pendulum_step(_p(state), _p(_f32(action)), C.byref(cfg), _p(obs), _p(reward), C.c_int64(n))
    return state, obs, reward


def pendulum_reset(n: int, seed: int, reset_count: int, env_offset: int = 0) -> tuple[np.ndarray, np.ndarray]:
    state = np.empty((2, n), np.float32)
    obs = np.empty((n, 3), np.float32)
    lib().oracle_pendulum_reset(_p(state), _p(obs), C.c_int64(n), C.c_uint64(seed), C.c_uint64(reset_count),
                                C.c_int64(env_offset))
    return state, obs


# --------------------------------------------------------------------------- #
# Rollout bookkeeping and stats.
# --------------------------------------------------------------------------- #
def rdr_step(rdr_t: np.ndarray, reward: np.ndarray, gamma: float) -> np.ndarray:
    rdr_t = _f32(rdr_t)
    out = np.empty_like(rdr_t)
    lib().oracle_rdr_step(_p(rdr_t), _p(_f32(reward)), _p(out), C.c_double(gamma), C.c_int64(rdr_t.size))
    return out


STAT_KEYS = (
    "returns/min", "returns/max", "returns/mean", "returns/std",
    "rewards/min", "rewards/max", "rewards/mean", "rewards/std", "reward_scale",
)


def rollout_stats(rewards: np.ndarray, rdr: None | np.ndarray) -> dict[str, float]:
    """rewards, rdr: env-major [n, h+1, 1]."""
    rewards = _f32(rewards)
    n, h1 = rewards.shape[0], rewards.shape[1]
    out = np.empty(9, np.float64)
    lib().oracle_rollout_stats(
        _p(rewards), _p(_f32(rdr)) if rdr is not None else None, C.c_int64(n), C.c_int64(h1 - 1), _p(out)
    )
    return dict(zip(STAT_KEYS, out.tolist()))


# --------------------------------------------------------------------------- #
# GAE.
# --------------------------------------------------------------------------- #
def gae(
    rewards: np.ndarray,
    values: np.ndarray,
    *,
    gamma: float = 0.95,
    gae_lambda: float = 0.95,
    reward_scale: float = 1.0,
    normalize_advantages: bool = True,
):
    """env-major [n, h+1, 1] in; returns dict(scaled_rewards, advantages, returns, mean, std)."""
    rewards = _f32(rewards).copy()
    values = _f32(values)
    n, h1 = rewards.shape[0], rewards.shape[1]
    adv = np.empty_like(rewards)
    ret = np.empty_like(rewards)
    moments = np.empty(2, np.float32)
    lib().oracle_gae(
        _p(rewards), _p(values), _p(adv), _p(ret), C.c_int64(n), C.c_int64(h1 - 1),
        C.c_double(gamma), C.c_double(gae_lambda), C.c_double(reward_scale),
        C.c_int(int(normalize_advantages)), _p(moments),
    )
    return {
        "scaled_rewards": rewards, "advantages": adv, "returns": ret,
        "mean": float(moments[0]), "std": float(moments[1]),
    }


# --------------------------------------------------------------------------- #
# Samplers.
# --------------------------------------------------------------------------- #
def categorical_sample(
    logits: np.ndarray,
    q: None | np.ndarray = None,
    *,
    seed: int = 0,
    step: int = 0,
    row_offset: int = 0,
    deterministic: bool = False,
):
    """logits [m, a, k] (+ optional injected Exp(1) noise of the same shape)
    -> (actions [m, a] int64, logp [m, 1])."""
    logits = _f32(logits)
    m, a, k = logits.shape
    actions = np.empty((m, a), np.int64)
    logp = np.empty((m, 1), np.float32)
    lib().oracle_categorical_sample(
        _p(logits), _p(_f32(q)) if q is not None else None, _p(actions), _p(logp),
        C.c_int64(m), C.c_int(a), C.c_int(k), C.c_uint64(seed), C.c_uint64(step),
        C.c_int64(row_offset), C.c_int(int(deterministic)),
    )
    return actions, logp


def normal_sample(
    mean: np.ndarray,
    log_std: np.ndarray,
    eps: None | np.ndarray = None,
    *,
    squashed: bool = False,
    seed: int = 0,
    step: int = 0,
    row_offset: int = 0,
    deterministic: bool = False,
):
    mean = _f32(mean)
    m, a = mean.shape
    actions = np.empty((m, a), np.float32)
    logp = np.empty((m, 1), np.float32)
    lib().oracle_normal_sample(
        _p(mean), _p(_f32(log_std)), _p(_f32(eps)) if eps is not None else None,
        _p(actions), _p(logp), C.c_int64(m), C.c_int(a), C.c_int(int(squashed)),
        C.c_uint64(seed), C.c_uint64(step), C.c_int64(row_offset), C.c_int(int(deterministic)),
    )
    return actions, logp


# --------------------------------------------------------------------------- #
# PPO losses (forward + backward).
# --------------------------------------------------------------------------- #
LOSS_KEYS = ("entropy", "policy", "vf", "total", "kl")


def ppo_loss_categorical(logits, values, actions, logp_old, adv, returns, hp: PPOHparams, *, grads: bool = True):
    logits = _f32(logits)
    m, a, k = logits.shape
    g_logits = np.empty_like(logits) if grads else None
    g_values = np.empty((m, 1), np.float32) if grads else None
    losses = np.empty(5, np.float64)
    lib().oracle_ppo_loss_categorical(
        _p(logits), _p(_f32(values)), _p(_i64(actions)), _p(_f32(logp_old)), _p(_f32(adv)),
        _p(_f32(returns)), C.c_int64(m), C.c_int(a), C.c_int(k), C.byref(hp),
        _p(g_logits), _p(g_values), _p(losses),
    )
    return dict(zip(LOSS_KEYS, losses.tolist())), g_logits, g_values


def ppo_loss_normal(mean, log_std, values, actions, logp_old, adv, returns, hp: PPOHparams, *, squashed: bool, grads: bool = True):
    mean = _f32(mean)
    m, a = mean.shape
    g_mean = np.empty_like(mean) if grads else None
    g_ls = np.empty_like(mean) if grads else None
    g_values = np.empty((m, 1), np.float32) if grads else None
    losses = np.empty(5, np.float64)
    lib().oracle_ppo_loss_normal(
        _p(mean), _p(_f32(log_std)), _p(_f32(values)), _p(_f32(actions)), _p(_f32(logp_old)),
        _p(_f32(adv)), _p(_f32(returns)), C.c_int64(m), C.c_int(a), C.c_int(int(squashed)),
        C.byref(hp), _p(g_mean), _p(g_ls), _p(g_values), _p(losses),
    )
    return dict(zip(LOSS_KEYS, losses.tolist())), g_mean, g_ls, g_values


# --------------------------------------------------------------------------- #
# Batcher.
# --------------------------------------------------------------------------- #
def gather_rows(index: np.ndarray, src: np.ndarray) -> np.ndarray:
    src = np.ascontiguousarray(src)
    index = _i64(index)
    row_bytes = src.strides[0]
    dst = np.empty((index.size, *src.shape[1:]), src.dtype)
    lib().oracle_gather_rows(_p(index), _p(src), _p(dst), C.c_int64(index.size), C.c_int64(row_bytes))
    return dst


def permutation(m: int, seed: int, iteration: int) -> np.ndarray:
    out = np.empty(m, np.int64)
    lib().oracle_permutation(_p(out), C.c_int64(m), C.c_uint64(seed), C.c_uint64(iteration))
    return out


def philox_words(seed: int, row: int, step: int, stream_block: int) -> np.ndarray:
    out = np.empty(4, np.uint32)
    lib().oracle_philox_words(C.c_uint64(seed), C.c_uint64(row), C.c_uint64(step), C.c_uint32(stream_block), _p(out))
    return out


def exponential(seed: int, row: int, step: int, w: int) -> float:
    return float(lib().oracle_exponential(seed, row, step, w))


def normal(seed: int, row: int, step: int, w: int) -> float:
    return float(lib().oracle_normal(seed, row, step, w))
