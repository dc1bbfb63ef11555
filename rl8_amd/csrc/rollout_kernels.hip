// K1 / K1b / K2 / K7: everything that happens once per timestep of
// Algorithm.collect(): action sampling, the batched env.step, the reversed-
// discounted-return recurrence and the rollout-buffer column writes.
//
// Reference: src/rl8/env.py:197-259 (dummy envs), examples/cartpole/env.py:12-64
// and :128-150 (CartPole), src/rl8/distributions.py:113-170 (sampling / logp),
// src/rl8/algorithms/_feedforward.py:373-393 (bookkeeping).
//
// Per timestep these move ~32-64 B per env (tens of MB at N = 2^20), i.e. a
// few microseconds of HBM time -- the same order as a launch.  The reference
// issues ~20-30 launches per timestep; here the policy network's outputs go
// through ONE fused launch per timestep for the built-in envs (sampler + step +
// bookkeeping), with every buffer column a contiguous time-major slab so all
// accesses are coalesced.  Standalone kernels back the public Env.step /
// Distribution.sample API and custom environments.
#include "common.hip.h"
#include "device_math.hip.h"

namespace rl8 {

// ---- standalone env kernels -------------------------------------------------
__global__ __launch_bounds__(kBlock) void dummy_env_step_kernel(float *__restrict__ state,
                                                                const void *__restrict__ action,
                                                                int is_discrete,
                                                                float *__restrict__ reward,
                                                                int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    const float s = is_discrete
                        ? dummy_step_discrete(state[i], static_cast<const int64_t *>(action)[i])
                        : dummy_step_continuous(state[i], static_cast<const float *>(action)[i]);
    state[i] = s;
    reward[i] = -fabsf(s);
  }
}

__global__ __launch_bounds__(kBlock) void dummy_env_reset_kernel(float *__restrict__ state,
                                                                 int64_t n, float bounds,
                                                                 uint64_t seed,
                                                                 uint64_t reset_count,
                                                                 int64_t env_offset) {
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    uint32_t r[4];
    rl8_philox4x32_10(seed, (uint64_t)(i + env_offset), reset_count,
                      rl8_stream_block(RL8_STREAM_RESET, 0), r);
    state[i] = rl8_u01_24(r[0]) * (bounds - (-bounds)) + (-bounds);
  }
}

__global__ __launch_bounds__(kBlock) void cartpole_step_kernel(
    float *__restrict__ state, const int64_t *__restrict__ action, rl8_cartpole_cfg cfg,
    float *__restrict__ obs, int64_t obs_stride, float *__restrict__ reward, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    CartPoleState s = {state[i], state[n + i], state[2 * n + i], state[3 * n + i]};
    const CartPoleOut o = cartpole_advance(s, action[i], cfg);
    state[i] = s.x; state[n + i] = s.x_dot; state[2 * n + i] = s.theta; state[3 * n + i] = s.theta_dot;
    float *ob = obs + i * obs_stride;
    ob[0] = s.x; ob[1] = s.x_dot; ob[2] = o.cos_theta; ob[3] = o.sin_theta; ob[4] = s.theta_dot;
    reward[i] = o.reward;
  }
}

__global__ __launch_bounds__(kBlock) void cartpole_reset_kernel(float *__restrict__ state,
                                                                int64_t n, float std,
                                                                uint64_t seed,
                                                                uint64_t reset_count,
                                                                int64_t env_offset,
                                                                float *__restrict__ obs,
                                                                int64_t obs_stride) {
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    uint32_t r[4];
    float z[4];
    rl8_philox4x32_10(seed, (uint64_t)(i + env_offset), reset_count,
                      rl8_stream_block(RL8_STREAM_RESET, 0), r);
    rl8_box_muller(r[0], r[1], &z[0], &z[1]);
    rl8_box_muller(r[2], r[3], &z[2], &z[3]);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      z[k] = z[k] * std + 0.0f;
      state[k * n + i] = z[k];
    }
    if (obs) {
      float *ob = obs + i * obs_stride;
      ob[0] = z[0]; ob[1] = z[1]; ob[2] = cosf(z[2]); ob[3] = sinf(z[2]); ob[4] = z[3];
    }
  }
}

// ---- standalone samplers ----------------------------------------------------
__global__ __launch_bounds__(kBlock) void categorical_sample_kernel(
    const float *__restrict__ logits, const float *__restrict__ noise,
    int64_t *__restrict__ action, float *__restrict__ logp, int64_t m, int a, int k, uint64_t seed,
    uint64_t step, int64_t row_offset, int deterministic) {
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  float nl[RL8_MAX_CLASSES], p[RL8_MAX_CLASSES];
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < m; i += stride) {
    float lp = 0.0f;
    for (int d = 0; d < a; ++d) {
      categorical_normalise_dyn<true>(logits + (i * a + d) * k, k, nl, p);
      int best = 0;
      float best_v = -INFINITY;
      for (int j = 0; j < k; ++j) {
        float v;
        if (deterministic) {
          v = p[j];
        } else {
          const float q = noise ? noise[(i * a + d) * k + j]
                                : rl8_exponential(seed, (uint64_t)(i + row_offset), step,
                                                  (uint32_t)(d * k + j));
          v = p[j] / q;
        }
        if (v > best_v) {
          best_v = v;
          best = j;
        }
      }
      action[i * a + d] = best;
      lp = d == 0 ? nl[best] : lp + nl[best];
    }
    logp[i] = lp;
  }
}

__global__ __launch_bounds__(kBlock) void normal_sample_kernel(
    const float *__restrict__ mean, const float *__restrict__ log_std,
    const float *__restrict__ noise, float *__restrict__ action, float *__restrict__ logp,
    int64_t m, int a, int squashed, uint64_t seed, uint64_t step, int64_t row_offset,
    int deterministic) {
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < m; i += stride) {
    float lp = 0.0f, corr = 0.0f;
    for (int d = 0; d < a; ++d) {
      float e = 0.0f;
      if (!deterministic)
        e = noise ? noise[i * a + d]
                  : rl8_normal(seed, (uint64_t)(i + row_offset), step, (uint32_t)d);
      float l, c;
      action[i * a + d] =
          normal_draw(mean[i * a + d], log_std[i * a + d], e, squashed != 0, deterministic != 0,
                      &l, &c);
      lp = d == 0 ? l : lp + l;
      corr = d == 0 ? c : corr + c;
    }
    logp[i] = squashed ? lp - corr : lp;
  }
}

// ---- bookkeeping for a generic Env (K1b) -----------------------------------
__global__ __launch_bounds__(kBlock) void rollout_scatter_kernel(
    const char *__restrict__ action, int64_t action_row_bytes, const float *__restrict__ logp,
    const float *__restrict__ value, const float *__restrict__ reward,
    const float *__restrict__ obs, int64_t obs_dim, char *__restrict__ action_col,
    float *__restrict__ logp_col, float *__restrict__ value_col, float *__restrict__ reward_col,
    float *__restrict__ obs_col_next, const float *__restrict__ rdr_t,
    float *__restrict__ rdr_t1, float gamma, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  const int64_t tid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  for (int64_t i = tid; i < n; i += stride) {
    const float r = reward[i];
    logp_col[i] = logp[i];
    value_col[i] = value[i];
    reward_col[i] = r;
    if (rdr_t1) rdr_t1[i] = gamma * rdr_t[i] + r;
  }
  // Rows wider than 4 B are copied as dwords, still lane-contiguous.
  const int64_t a_words = n * action_row_bytes / 4;
  const uint32_t *asrc = reinterpret_cast<const uint32_t *>(action);
  uint32_t *adst = reinterpret_cast<uint32_t *>(action_col);
  for (int64_t i = tid; i < a_words; i += stride) adst[i] = asrc[i];
  const int64_t o_words = n * obs_dim;
  for (int64_t i = tid; i < o_words; i += stride) obs_col_next[i] = obs[i];
}

// ---- fused per-timestep kernels --------------------------------------------
// Dummy envs: one lane per env.  Reads features (8 B), value (4), state (4),
// rdr[t] (4) [+ injected noise]; writes action (8 / 4), logp, value, reward,
// obs[t+1], state, rdr[t+1] (4 each).
template <bool DISCRETE>
__global__ __launch_bounds__(kBlock) void rollout_step_dummy_kernel(
    int squashed, const float *__restrict__ features, const float *__restrict__ features2,
    const float *__restrict__ value, const float *__restrict__ noise, float *__restrict__ state,
    void *__restrict__ action_col, float *__restrict__ logp_col, float *__restrict__ value_col,
    float *__restrict__ reward_col, float *__restrict__ obs_col_next,
    const float *__restrict__ rdr_t, float *__restrict__ rdr_t1, float gamma, int64_t n,
    uint64_t seed, uint64_t step, int64_t env_offset, int deterministic) {
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    float s = state[i], lp;
    if (DISCRETE) {
      const float2 xl = *reinterpret_cast<const float2 *>(features + 2 * i);
      const float x[2] = {xl.x, xl.y};
      float q[2];
      if (noise) {
        const float2 qn = *reinterpret_cast<const float2 *>(noise + 2 * i);
        q[0] = qn.x; q[1] = qn.y;
      }
      const int act = categorical_draw<2>(x, noise ? q : nullptr, seed, (uint64_t)(i + env_offset),
                                          step, 0u, deterministic != 0, &lp);
      static_cast<int64_t *>(action_col)[i] = act;
      s = dummy_step_discrete(s, act);
    } else {
      float e = 0.0f;
      if (!deterministic)
        e = noise ? noise[i] : rl8_normal(seed, (uint64_t)(i + env_offset), step, 0u);
      float l, c;
      const float act = normal_draw(features[i], features2[i], e, squashed != 0,
                                    deterministic != 0, &l, &c);
      lp = squashed ? l - c : l;
      static_cast<float *>(action_col)[i] = act;
      s = dummy_step_continuous(s, act);
    }
    const float r = -fabsf(s);
    state[i] = s;
    obs_col_next[i] = s;
    reward_col[i] = r;
    logp_col[i] = lp;
    value_col[i] = value[i];
    if (rdr_t1) rdr_t1[i] = gamma * rdr_t[i] + r;
  }
}

// The same for the recurrent discrete model (two-way categorical) with the two output heads evaluated in the kernel:
// logits = h x w_pol^T + b_pol, value = h x w_vf^T + b_vf from the LSTM's h_t [N][256] -- four lanes per env, each a
// quarter of the 256 inputs, the arithmetic and its order those of linear_heads_forward_kernel (lstm_kernels.hip), so
// the rollout is bit for bit the one of the two launches it replaces -- then lane 0 of the four draws the action and
// steps the env.  At the recurrent bench's 8 192 envs per GPU both launches were a few microseconds of work behind a
// launch each.
__global__ __launch_bounds__(kBlock) void rollout_step_dummy_heads_kernel(
    const float4 *__restrict__ h, const float *__restrict__ w_pol, const float *__restrict__ b_pol,
    const float *__restrict__ w_vf, const float *__restrict__ b_vf, const float *__restrict__ noise,
    float *__restrict__ state, int64_t *__restrict__ action_col, float *__restrict__ logp_col,
    float *__restrict__ value_col, float *__restrict__ reward_col, float *__restrict__ obs_col_next,
    const float *__restrict__ rdr_t, float *__restrict__ rdr_t1, float gamma, int64_t n, uint64_t seed, uint64_t step,
    int64_t env_offset, int deterministic) {
  constexpr int kIn = 256, kOut = 3;
  __shared__ float4 ws[kOut * kIn / 4];
  for (int i = threadIdx.x; i < kOut * kIn / 4; i += kBlock)
    ws[i] = i < 2 * (kIn / 4) ? reinterpret_cast<const float4 *>(w_pol)[i] : reinterpret_cast<const float4 *>(w_vf)[i - 2 * (kIn / 4)];
  __syncthreads();
  const int q4 = threadIdx.x & 3;
  const int64_t stride = (int64_t)gridDim.x * (kBlock / 4);
  for (int64_t i = (int64_t)blockIdx.x * (kBlock / 4) + (threadIdx.x >> 2); i < n; i += stride) {
    float o[kOut] = {0.0f, 0.0f, 0.0f};
    const float4 *hr = h + i * (kIn / 4) + q4 * 16;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const float4 v = hr[k];
#pragma unroll
      for (int q = 0; q < kOut; ++q) {
        const float4 wv = ws[q * (kIn / 4) + q4 * 16 + k];
        o[q] = __builtin_fmaf(v.x, wv.x, o[q]);
        o[q] = __builtin_fmaf(v.y, wv.y, o[q]);
        o[q] = __builtin_fmaf(v.z, wv.z, o[q]);
        o[q] = __builtin_fmaf(v.w, wv.w, o[q]);
      }
    }
#pragma unroll
    for (int q = 0; q < kOut; ++q) {
      o[q] += __shfl_xor(o[q], 1, kWave);
      o[q] += __shfl_xor(o[q], 2, kWave);
    }
    if (q4 != 0) continue;
    const float x[2] = {o[0] + b_pol[0], o[1] + b_pol[1]};
    float q[2], lp;
    if (noise) {
      const float2 qn = *reinterpret_cast<const float2 *>(noise + 2 * i);
      q[0] = qn.x; q[1] = qn.y;
    }
    const int act = categorical_draw<2>(x, noise ? q : nullptr, seed, (uint64_t)(i + env_offset), step, 0u,
                                        deterministic != 0, &lp);
    action_col[i] = act;
    const float s = dummy_step_discrete(state[i], act);
    const float r = -fabsf(s);
    state[i] = s;
    obs_col_next[i] = s;
    reward_col[i] = r;
    logp_col[i] = lp;
    value_col[i] = o[2] + b_vf[0];
    if (rdr_t1) rdr_t1[i] = gamma * rdr_t[i] + r;
  }
}

// CartPole (K = 3): one lane per env; obs[t+1] rows are 20 B so the five
// components of 64 consecutive envs fill 1280 contiguous bytes per wave.
__global__ __launch_bounds__(kBlock) void rollout_step_cartpole_kernel(
    const float *__restrict__ logits, const float *__restrict__ value,
    const float *__restrict__ noise, float *__restrict__ state, rl8_cartpole_cfg cfg,
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
    CartPoleState s = {state[i], state[n + i], state[2 * n + i], state[3 * n + i]};
    const CartPoleOut o = cartpole_advance(s, act, cfg);
    state[i] = s.x; state[n + i] = s.x_dot; state[2 * n + i] = s.theta; state[3 * n + i] = s.theta_dot;
    float *ob = obs_col_next + 5 * i;
    ob[0] = s.x; ob[1] = s.x_dot; ob[2] = o.cos_theta; ob[3] = o.sin_theta; ob[4] = s.theta_dot;
    action_col[i] = act;
    reward_col[i] = o.reward;
    logp_col[i] = lp;
    value_col[i] = value[i];
    if (rdr_t1) rdr_t1[i] = gamma * rdr_t[i] + o.reward;
  }
}

}  // namespace rl8

using namespace rl8;

RL8_API int rl8_dummy_env_step_f32(float *state, const void *action, int is_discrete,
                                   float *reward_out, int64_t n, void *stream) {
  if (!state || !action || !reward_out) return RL8_ENULL;
  if (n <= 0) return RL8_ESIZE;
  dummy_env_step_kernel<<<grid_for(n, kBlock), kBlock, 0, (hipStream_t)stream>>>(
      state, action, is_discrete, reward_out, n);
  return launch_status();
}

RL8_API int rl8_dummy_env_reset_f32(float *state, int64_t n, float bounds, uint64_t seed,
                                    uint64_t reset_count, int64_t env_offset, void *stream) {
  if (!state) return RL8_ENULL;
  if (n <= 0) return RL8_ESIZE;
  dummy_env_reset_kernel<<<grid_for(n, kBlock), kBlock, 0, (hipStream_t)stream>>>(
      state, n, bounds, seed, reset_count, env_offset);
  return launch_status();
}

RL8_API int rl8_cartpole_step_f32(float *state, const int64_t *action, const rl8_cartpole_cfg *cfg,
                                  float *obs_out, int64_t obs_stride, float *reward_out, int64_t n,
                                  void *stream) {
  if (!state || !action || !cfg || !obs_out || !reward_out) return RL8_ENULL;
  if (n <= 0 || obs_stride < 5) return RL8_ESIZE;
  cartpole_step_kernel<<<grid_for(n, kBlock), kBlock, 0, (hipStream_t)stream>>>(
      state, action, *cfg, obs_out, obs_stride, reward_out, n);
  return launch_status();
}

RL8_API int rl8_cartpole_reset_f32(float *state, int64_t n, float std, uint64_t seed,
                                   uint64_t reset_count, int64_t env_offset, float *obs_out,
                                   int64_t obs_stride, void *stream) {
  if (!state) return RL8_ENULL;
  if (n <= 0 || (obs_out && obs_stride < 5)) return RL8_ESIZE;
  cartpole_reset_kernel<<<grid_for(n, kBlock), kBlock, 0, (hipStream_t)stream>>>(
      state, n, std, seed, reset_count, env_offset, obs_out, obs_stride);
  return launch_status();
}

RL8_API int rl8_categorical_sample_logp_f32(const float *logits, const float *noise,
                                            int64_t *action_out, float *logp_out, int64_t m, int a,
                                            int k, uint64_t seed, uint64_t step,
                                            int64_t row_offset, int deterministic, void *stream) {
  if (!logits || !action_out || !logp_out) return RL8_ENULL;
  if (m <= 0 || a <= 0 || k <= 1 || k > RL8_MAX_CLASSES) return RL8_ESIZE;
  categorical_sample_kernel<<<grid_for(m, kBlock), kBlock, 0, (hipStream_t)stream>>>(
      logits, noise, action_out, logp_out, m, a, k, seed, step, row_offset, deterministic);
  return launch_status();
}

RL8_API int rl8_normal_sample_logp_f32(const float *mean, const float *log_std, const float *noise,
                                       float *action_out, float *logp_out, int64_t m, int a,
                                       int squashed, uint64_t seed, uint64_t step,
                                       int64_t row_offset, int deterministic, void *stream) {
  if (!mean || !log_std || !action_out || !logp_out) return RL8_ENULL;
  if (m <= 0 || a <= 0) return RL8_ESIZE;
  normal_sample_kernel<<<grid_for(m, kBlock), kBlock, 0, (hipStream_t)stream>>>(
      mean, log_std, noise, action_out, logp_out, m, a, squashed, seed, step, row_offset,
      deterministic);
  return launch_status();
}

RL8_API int rl8_rollout_scatter_f32(const void *action, int64_t action_row_bytes,
                                    const float *logp, const float *value, const float *reward,
                                    const float *obs, int64_t obs_dim, void *action_col,
                                    float *logp_col, float *value_col, float *reward_col,
                                    float *obs_col_next, const float *rdr_t, float *rdr_t1,
                                    float gamma, int64_t n, void *stream) {
  if (!action || !logp || !value || !reward || !obs || !action_col || !logp_col || !value_col ||
      !reward_col || !obs_col_next)
    return RL8_ENULL;
  if ((rdr_t == nullptr) != (rdr_t1 == nullptr)) return RL8_ENULL;
  if (n <= 0 || obs_dim <= 0 || action_row_bytes <= 0 || action_row_bytes % 4) return RL8_ESIZE;
  int64_t widest = n * (obs_dim > action_row_bytes / 4 ? obs_dim : action_row_bytes / 4);
  rollout_scatter_kernel<<<grid_for(widest, kBlock), kBlock, 0, (hipStream_t)stream>>>(
      static_cast<const char *>(action), action_row_bytes, logp, value, reward, obs, obs_dim,
      static_cast<char *>(action_col), logp_col, value_col, reward_col, obs_col_next, rdr_t, rdr_t1,
      gamma, n);
  return launch_status();
}

RL8_API int rl8_rollout_step_dummy_f32(int is_discrete, int squashed, const float *features,
                                       const float *features2, const float *value,
                                       const float *noise, float *state, void *action_col,
                                       float *logp_col, float *value_col, float *reward_col,
                                       float *obs_col_next, const float *rdr_t, float *rdr_t1,
                                       float gamma, int64_t n, uint64_t seed, uint64_t step,
                                       int64_t env_offset, int deterministic, void *stream) {
  if (!features || !value || !state || !action_col || !logp_col || !value_col || !reward_col ||
      !obs_col_next)
    return RL8_ENULL;
  if (!is_discrete && !features2) return RL8_ENULL;
  if ((rdr_t == nullptr) != (rdr_t1 == nullptr)) return RL8_ENULL;
  if (n <= 0) return RL8_ESIZE;
  if (is_discrete && ((reinterpret_cast<uintptr_t>(features) & 7u) ||
                      (noise && (reinterpret_cast<uintptr_t>(noise) & 7u))))
    return RL8_EALIGN;
  hipStream_t s = (hipStream_t)stream;
  static const int cap = env_int("RL8_STEP_GRID_CAP");
  const int grid = grid_for(n, kBlock, cap > 0 ? cap : kMaxGrid);
  if (is_discrete)
    rollout_step_dummy_kernel<true><<<grid, kBlock, 0, s>>>(
        squashed, features, features2, value, noise, state, action_col, logp_col, value_col,
        reward_col, obs_col_next, rdr_t, rdr_t1, gamma, n, seed, step, env_offset, deterministic);
  else
    rollout_step_dummy_kernel<false><<<grid, kBlock, 0, s>>>(
        squashed, features, features2, value, noise, state, action_col, logp_col, value_col,
        reward_col, obs_col_next, rdr_t, rdr_t1, gamma, n, seed, step, env_offset, deterministic);
  return launch_status();
}

RL8_API int rl8_rollout_step_dummy_heads_f32(const float *h, const float *w_pol, const float *b_pol, const float *w_vf,
                                             const float *b_vf, const float *noise, float *state, int64_t *action_col,
                                             float *logp_col, float *value_col, float *reward_col, float *obs_col_next,
                                             const float *rdr_t, float *rdr_t1, float gamma, int64_t n, uint64_t seed,
                                             uint64_t step, int64_t env_offset, int deterministic, void *stream) {
  if (!h || !w_pol || !b_pol || !w_vf || !b_vf || !state || !action_col || !logp_col || !value_col || !reward_col ||
      !obs_col_next)
    return RL8_ENULL;
  if ((rdr_t == nullptr) != (rdr_t1 == nullptr)) return RL8_ENULL;
  if (n <= 0) return RL8_ESIZE;
  if (!aligned16(h) || !aligned16(w_pol) || !aligned16(w_vf) || (noise && (reinterpret_cast<uintptr_t>(noise) & 7u)))
    return RL8_EALIGN;
  rollout_step_dummy_heads_kernel<<<grid_for(n, kBlock / 4), kBlock, 0, (hipStream_t)stream>>>(
      reinterpret_cast<const float4 *>(h), w_pol, b_pol, w_vf, b_vf, noise, state, action_col, logp_col, value_col,
      reward_col, obs_col_next, rdr_t, rdr_t1, gamma, n, seed, step, env_offset, deterministic);
  return launch_status();
}

RL8_API int rl8_rollout_step_cartpole_f32(const float *logits, const float *value,
                                          const float *noise, float *state,
                                          const rl8_cartpole_cfg *cfg, int64_t *action_col,
                                          float *logp_col, float *value_col, float *reward_col,
                                          float *obs_col_next, const float *rdr_t, float *rdr_t1,
                                          float gamma, int64_t n, uint64_t seed, uint64_t step,
                                          int64_t env_offset, int deterministic, void *stream) {
  if (!logits || !value || !state || !cfg || !action_col || !logp_col || !value_col ||
      !reward_col || !obs_col_next)
    return RL8_ENULL;
  if ((rdr_t == nullptr) != (rdr_t1 == nullptr)) return RL8_ENULL;
  if (n <= 0) return RL8_ESIZE;
  rollout_step_cartpole_kernel<<<grid_for(n, kBlock), kBlock, 0, (hipStream_t)stream>>>(
      logits, value, noise, state, *cfg, action_col, logp_col, value_col, reward_col, obs_col_next,
      rdr_t, rdr_t1, gamma, n, seed, step, env_offset, deterministic);
  return launch_status();
}
