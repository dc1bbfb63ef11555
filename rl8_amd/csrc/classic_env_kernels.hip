// N2: the reference's other two example environments -- MountainCar
// (examples/mountain_car/env.py:12-38) and Pendulum (examples/pendulum/env.py:12-39)
// -- on the template of the CartPole kernels: struct-of-arrays state [2][N], one
// lane per env, a standalone step / reset pair behind Env.step / Env.reset and a
// fused per-timestep kernel (sampler + physics + rollout-buffer bookkeeping) for
// Algorithm.collect().  Pendulum is the continuous-action case: the fused kernel
// draws from Normal / SquashedNormal (src/rl8/distributions.py:240-330).
//
// Traffic per env and timestep, fused kernels: MountainCar reads logits 12 B,
// value 4, state 8, rdr 4 and writes action 8, logp 4, value 4, reward 4,
// obs 8, state 8, rdr 4 = 68 B; Pendulum 64 B.  HBM-bound, a few microseconds
// per launch at N = 2^20.
#include "common.hip.h"
#include "device_math.hip.h"

namespace rl8 {

__global__ __launch_bounds__(kBlock) void mountain_car_step_kernel(
    float *__restrict__ state, const int64_t *__restrict__ action, rl8_mountain_car_cfg cfg,
    float *__restrict__ obs, int64_t obs_stride, float *__restrict__ reward, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    float p = state[i], v = state[n + i];
    const float r = mountain_car_advance(p, v, action[i], cfg);
    state[i] = p;
    state[n + i] = v;
    obs[i * obs_stride] = p;
    obs[i * obs_stride + 1] = v;
    reward[i] = r;
  }
}

__global__ __launch_bounds__(kBlock) void mountain_car_reset_kernel(
    float *__restrict__ state, int64_t n, uint64_t seed, uint64_t reset_count, int64_t env_offset,
    float *__restrict__ obs, int64_t obs_stride) {
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    uint32_t r[4];
    float z0, z1;
    rl8_philox4x32_10(seed, (uint64_t)(i + env_offset), reset_count,
                      rl8_stream_block(RL8_STREAM_RESET, 0), r);
    rl8_box_muller(r[0], r[1], &z0, &z1);
    const float p = z0 * 0.05f + -0.5f, v = z1 * 0.05f + 0.0f;
    state[i] = p;
    state[n + i] = v;
    if (obs) {
      obs[i * obs_stride] = p;
      obs[i * obs_stride + 1] = v;
    }
  }
}

__global__ __launch_bounds__(kBlock) void rollout_step_mountain_car_kernel(
    const float *__restrict__ logits, const float *__restrict__ value,
    const float *__restrict__ noise, float *__restrict__ state, rl8_mountain_car_cfg cfg,
    int64_t *__restrict__ action_col, float *__restrict__ logp_col, float *__restrict__ value_col,
    float *__restrict__ reward_col, float *__restrict__ obs_col_next,
    const float *__restrict__ rdr_t, float *__restrict__ rdr_t1, float gamma, int64_t n,
    uint64_t seed, uint64_t step, int64_t env_offset, int deterministic) {
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    const float x[3] = {logits[3 * i], logits[3 * i + 1], logits[3 * i + 2]};
    float q[3];
    if (noise) {
      q[0] = noise[3 * i]; q[1] = noise[3 * i + 1]; q[2] = noise[3 * i + 2];
    }
    float lp;
    const int act = categorical_draw<3>(x, noise ? q : nullptr, seed, (uint64_t)(i + env_offset),
                                        step, 0u, deterministic != 0, &lp);
    float p = state[i], v = state[n + i];
    const float r = mountain_car_advance(p, v, act, cfg);
    state[i] = p;
    state[n + i] = v;
    *reinterpret_cast<float2 *>(obs_col_next + 2 * i) = make_float2(p, v);
    action_col[i] = act;
    reward_col[i] = r;
    logp_col[i] = lp;
    value_col[i] = value[i];
    if (rdr_t1) rdr_t1[i] = gamma * rdr_t[i] + r;
  }
}

__global__ __launch_bounds__(kBlock) void pendulum_step_kernel(
    float *__restrict__ state, const float *__restrict__ action, rl8_pendulum_cfg cfg,
    float *__restrict__ obs, int64_t obs_stride, float *__restrict__ reward, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    float th = state[i], td = state[n + i];
    const float r = pendulum_advance(th, td, action[i], cfg);
    state[i] = th;
    state[n + i] = td;
    float *ob = obs + i * obs_stride;
    ob[0] = cosf(th); ob[1] = sinf(th); ob[2] = td;
    reward[i] = r;
  }
}

__global__ __launch_bounds__(kBlock) void pendulum_reset_kernel(
    float *__restrict__ state, int64_t n, uint64_t seed, uint64_t reset_count, int64_t env_offset,
    float *__restrict__ obs, int64_t obs_stride) {
  const float pi = (float)3.141592653589793;
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    uint32_t r[4];
    rl8_philox4x32_10(seed, (uint64_t)(i + env_offset), reset_count,
                      rl8_stream_block(RL8_STREAM_RESET, 0), r);
    const float th = rl8_u01_24(r[0]) * (pi - (-pi)) + (-pi);
    const float td = rl8_u01_24(r[1]) * (1.0f - (-1.0f)) + (-1.0f);
    state[i] = th;
    state[n + i] = td;
    if (obs) {
      float *ob = obs + i * obs_stride;
      ob[0] = cosf(th); ob[1] = sinf(th); ob[2] = td;
    }
  }
}

__global__ __launch_bounds__(kBlock) void rollout_step_pendulum_kernel(
    int squashed, const float *__restrict__ mean, const float *__restrict__ log_std,
    const float *__restrict__ value, const float *__restrict__ noise, float *__restrict__ state,
    rl8_pendulum_cfg cfg, float *__restrict__ action_col, float *__restrict__ logp_col,
    float *__restrict__ value_col, float *__restrict__ reward_col,
    float *__restrict__ obs_col_next, const float *__restrict__ rdr_t, float *__restrict__ rdr_t1,
    float gamma, int64_t n, uint64_t seed, uint64_t step, int64_t env_offset, int deterministic) {
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    float e = 0.0f;
    if (!deterministic) e = noise ? noise[i] : rl8_normal(seed, (uint64_t)(i + env_offset), step, 0u);
    float l, c;
    const float act = normal_draw(mean[i], log_std[i], e, squashed != 0, deterministic != 0, &l, &c);
    const float lp = squashed ? l - c : l;
    float th = state[i], td = state[n + i];
    const float r = pendulum_advance(th, td, act, cfg);
    state[i] = th;
    state[n + i] = td;
    float *ob = obs_col_next + 3 * i;
    ob[0] = cosf(th); ob[1] = sinf(th); ob[2] = td;
    action_col[i] = act;
    reward_col[i] = r;
    logp_col[i] = lp;
    value_col[i] = value[i];
    if (rdr_t1) rdr_t1[i] = gamma * rdr_t[i] + r;
  }
}

}  // namespace rl8

using namespace rl8;

RL8_API int rl8_mountain_car_step_f32(float *state, const int64_t *action,
                                      const rl8_mountain_car_cfg *cfg, float *obs_out,
                                      int64_t obs_stride, float *reward_out, int64_t n,
                                      void *stream) {
  if (!state || !action || !cfg || !obs_out || !reward_out) return RL8_ENULL;
  if (n <= 0 || obs_stride < 2) return RL8_ESIZE;
  mountain_car_step_kernel<<<grid_for(n, kBlock), kBlock, 0, (hipStream_t)stream>>>(
      state, action, *cfg, obs_out, obs_stride, reward_out, n);
  return launch_status();
}

RL8_API int rl8_mountain_car_reset_f32(float *state, int64_t n, uint64_t seed,
                                       uint64_t reset_count, int64_t env_offset, float *obs_out,
                                       int64_t obs_stride, void *stream) {
  if (!state) return RL8_ENULL;
  if (n <= 0 || (obs_out && obs_stride < 2)) return RL8_ESIZE;
  mountain_car_reset_kernel<<<grid_for(n, kBlock), kBlock, 0, (hipStream_t)stream>>>(
      state, n, seed, reset_count, env_offset, obs_out, obs_stride);
  return launch_status();
}

RL8_API int rl8_rollout_step_mountain_car_f32(
    const float *logits, const float *value, const float *noise, float *state,
    const rl8_mountain_car_cfg *cfg, int64_t *action_col, float *logp_col, float *value_col,
    float *reward_col, float *obs_col_next, const float *rdr_t, float *rdr_t1, float gamma,
    int64_t n, uint64_t seed, uint64_t step, int64_t env_offset, int deterministic, void *stream) {
  if (!logits || !value || !state || !cfg || !action_col || !logp_col || !value_col ||
      !reward_col || !obs_col_next)
    return RL8_ENULL;
  if ((rdr_t == nullptr) != (rdr_t1 == nullptr)) return RL8_ENULL;
  if (n <= 0) return RL8_ESIZE;
  if (reinterpret_cast<uintptr_t>(obs_col_next) & 7u) return RL8_EALIGN;
  rollout_step_mountain_car_kernel<<<grid_for(n, kBlock), kBlock, 0, (hipStream_t)stream>>>(
      logits, value, noise, state, *cfg, action_col, logp_col, value_col, reward_col, obs_col_next,
      rdr_t, rdr_t1, gamma, n, seed, step, env_offset, deterministic);
  return launch_status();
}

RL8_API int rl8_pendulum_step_f32(float *state, const float *action, const rl8_pendulum_cfg *cfg,
                                  float *obs_out, int64_t obs_stride, float *reward_out, int64_t n,
                                  void *stream) {
  if (!state || !action || !cfg || !obs_out || !reward_out) return RL8_ENULL;
  if (n <= 0 || obs_stride < 3) return RL8_ESIZE;
  pendulum_step_kernel<<<grid_for(n, kBlock), kBlock, 0, (hipStream_t)stream>>>(
      state, action, *cfg, obs_out, obs_stride, reward_out, n);
  return launch_status();
}

RL8_API int rl8_pendulum_reset_f32(float *state, int64_t n, uint64_t seed, uint64_t reset_count,
                                   int64_t env_offset, float *obs_out, int64_t obs_stride,
                                   void *stream) {
  if (!state) return RL8_ENULL;
  if (n <= 0 || (obs_out && obs_stride < 3)) return RL8_ESIZE;
  pendulum_reset_kernel<<<grid_for(n, kBlock), kBlock, 0, (hipStream_t)stream>>>(
      state, n, seed, reset_count, env_offset, obs_out, obs_stride);
  return launch_status();
}

RL8_API int rl8_rollout_step_pendulum_f32(int squashed, const float *mean, const float *log_std,
                                          const float *value, const float *noise, float *state,
                                          const rl8_pendulum_cfg *cfg, float *action_col,
                                          float *logp_col, float *value_col, float *reward_col,
                                          float *obs_col_next, const float *rdr_t, float *rdr_t1,
                                          float gamma, int64_t n, uint64_t seed, uint64_t step,
                                          int64_t env_offset, int deterministic, void *stream) {
  if (!mean || !log_std || !value || !state || !cfg || !action_col || !logp_col || !value_col ||
      !reward_col || !obs_col_next)
    return RL8_ENULL;
  if ((rdr_t == nullptr) != (rdr_t1 == nullptr)) return RL8_ENULL;
  if (n <= 0) return RL8_ESIZE;
  rollout_step_pendulum_kernel<<<grid_for(n, kBlock), kBlock, 0, (hipStream_t)stream>>>(
      squashed, mean, log_std, value, noise, state, *cfg, action_col, logp_col, value_col,
      reward_col, obs_col_next, rdr_t, rdr_t1, gamma, n, seed, step, env_offset, deterministic);
  return launch_status();
}
