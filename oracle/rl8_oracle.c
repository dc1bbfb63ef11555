/*
 * rl8_oracle.c -- CPU restatement of the rl8 PPO hot path.  TEST INFRASTRUCTURE.
 *
 * This file is the parity oracle for the HIP kernels in rl8_amd/csrc. It is NOT
 * part of the product: only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it, and only as the checker / reported baseline.
 *
 * Every function restates, in plain scalar C (fp32 arithmetic, one rounding per
 * reference op, compiled with -ffp-contract=off so nothing is fused), the
 * operation sequence of the reference function it cites. Citations are relative
 * to /root/reference (theOGognf/rl8).
 *
 * Pinning: tests/test_oracle_golden.py checks every function here against the
 * golden vectors in tests/golden/ (.npz files), which were produced by importing the
 * unmodified reference in the build container (tests/golden/generate_fixtures.py).
 *
 * Third-party arithmetic on this path lives in torch (torch.distributions
 * Categorical / Normal log_prob / entropy / sample, F.smooth_l1_loss,
 * torch.std_mean, torch.multinomial), torch 2.10.0 in the build container; its
 * published definitions are restated inline and pinned by the same fixtures.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "rl8_philox.h"

#define ORACLE_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------ */
/* Correctly-rounded-in-practice exp/log for the sampler: evaluate in fp64,  */
/* round once to fp32.  The HIP sampler does the same so that action indices */
/* agree bit-for-bit between host and device at any size.                    */
/* ------------------------------------------------------------------------ */
static inline float cr_expf(float x) { return (float)exp((double)x); }
static inline float cr_logf(float x) { return (float)log((double)x); }

/* ------------------------------------------------------------------------ */
/* a-1  DummyEnv.step          src/rl8/env.py:224-230 (continuous), :253-259  */
/* ------------------------------------------------------------------------ */
/* Discrete: state += 2*a - 1 (int64 arithmetic, then promoted to f32 by the
 * in-place add); continuous: state += a.  obs aliases state; reward = -|state|. */
ORACLE_API void oracle_dummy_env_step_discrete(float *state, const int64_t *action,
                                               float *reward, int64_t n) {
  for (int64_t i = 0; i < n; ++i) {
    int64_t d = 2 * action[i] - 1;
    state[i] = state[i] + (float)d;
    reward[i] = -fabsf(state[i]);
  }
}

ORACLE_API void oracle_dummy_env_step_continuous(float *state, const float *action,
                                                 float *reward, int64_t n) {
  for (int64_t i = 0; i < n; ++i) {
    state[i] = state[i] + action[i];
    reward[i] = -fabsf(state[i]);
  }
}

/* DummyEnv.reset  src/rl8/env.py:197-203: state ~ U(-bounds, bounds).  torch's
 * generator stream cannot be reproduced on a GPU, so the build owns the noise:
 * Philox4x32-10 keyed by (seed), counter (env, reset_count, STREAM_RESET, 0);
 * u in [0,1) with 24 bits; state = u * (2*bounds) - bounds, i.e. torch's
 * uniform_ transform `u * (to - from) + from`. */
ORACLE_API void oracle_dummy_env_reset(float *state, int64_t n, float bounds,
                                       uint64_t seed, uint64_t reset_count,
                                       int64_t env_offset) {
  for (int64_t i = 0; i < n; ++i) {
    uint32_t r[4];
    rl8_philox4x32_10(seed, (uint64_t)(i + env_offset), reset_count,
                      rl8_stream_block(RL8_STREAM_RESET, 0), r);
    float u = rl8_u01_24(r[0]);
    state[i] = u * (bounds - (-bounds)) + (-bounds);
  }
}

/* ------------------------------------------------------------------------ */
/* a-2  CartPole step          examples/cartpole/env.py:12-64                 */
/* ------------------------------------------------------------------------ */
typedef struct {
  float force_mag, gravity, length, pole_mass, pole_mass_length, total_mass, tau;
  int32_t semi_implicit; /* 0 = "euler", 1 = anything else */
} oracle_cartpole_cfg;

/* state: SoA [4][n] rows x, x_dot, theta, theta_dot (updated in place);
 * obs_out: [n][5] row-major (the reference returns the transpose view of a
 * [5][n] block; values are identical); reward_out: [n]. */
ORACLE_API void oracle_cartpole_step(float *state, const int64_t *action,
                                     const oracle_cartpole_cfg *cfg, float *obs_out,
                                     float *reward_out, int64_t n) {
  float *xs = state, *xds = state + n, *ths = state + 2 * n, *thds = state + 3 * n;
  const float four_thirds = (float)(4.0 / 3.0);
  for (int64_t i = 0; i < n; ++i) {
    float x = xs[i], x_dot = xds[i], theta = ths[i], theta_dot = thds[i];
    float force = (float)(action[i] - 1) * cfg->force_mag; /* :31 */
    float costheta = cosf(theta), sintheta = sinf(theta);   /* :32-33 */
    /* :37  tmp = (force + pml * theta_dot**2 * sintheta) / total_mass */
    float tmp = (force + (cfg->pole_mass_length * (theta_dot * theta_dot)) * sintheta) /
                cfg->total_mass;
    /* :38-40 */
    float num = cfg->gravity * sintheta - costheta * tmp;
    float den = cfg->length *
                (four_thirds - (cfg->pole_mass * (costheta * costheta)) / cfg->total_mass);
    float theta_acc = num / den;
    /* :41 */
    float x_acc = tmp - ((cfg->pole_mass_length * theta_acc) * costheta) / cfg->total_mass;
    if (!cfg->semi_implicit) { /* :43-47 */
      x = x + cfg->tau * x_dot;
      x_dot = x_dot + cfg->tau * x_acc;
      theta = theta + cfg->tau * theta_dot;
      theta_dot = theta_dot + cfg->tau * theta_acc;
    } else { /* :48-52 */
      x_dot = x_dot + cfg->tau * x_acc;
      x = x + cfg->tau * x_dot;
      theta_dot = theta_dot + cfg->tau * theta_acc;
      theta = theta + cfg->tau * theta_dot;
    }
    xs[i] = x; xds[i] = x_dot; ths[i] = theta; thds[i] = theta_dot;
    float c = cosf(theta), s = sinf(theta); /* :55 */
    obs_out[i * 5 + 0] = x;
    obs_out[i * 5 + 1] = x_dot;
    obs_out[i * 5 + 2] = c;
    obs_out[i * 5 + 3] = s;
    obs_out[i * 5 + 4] = theta_dot;
    /* :56-63 */
    float theta_error = fabsf(c - 1.0f) + fabsf(s - 0.0f);
    float other = (fabsf(x) + fabsf(x_dot)) + fabsf(theta_dot);
    reward_out[i] = -(theta_error + other);
  }
}

/* CartPole.reset  examples/cartpole/env.py:128-136: state ~ N(0, 0.01) [4,n];
 * build-owned noise (see oracle_dummy_env_reset): Box-Muller in fp64 on two
 * 24-bit uniforms per pair, counter (env, reset_count, STREAM_RESET, 0) gives
 * 4 words -> 2 pairs -> 4 normals = the 4 state rows of one env. */
ORACLE_API void oracle_cartpole_reset(float *state, int64_t n, float std, uint64_t seed,
                                      uint64_t reset_count, int64_t env_offset) {
  for (int64_t i = 0; i < n; ++i) {
    uint32_t r[4];
    float z[4];
    rl8_philox4x32_10(seed, (uint64_t)(i + env_offset), reset_count,
                      rl8_stream_block(RL8_STREAM_RESET, 0), r);
    rl8_box_muller(r[0], r[1], &z[0], &z[1]);
    rl8_box_muller(r[2], r[3], &z[2], &z[3]);
    for (int k = 0; k < 4; ++k) state[k * n + i] = z[k] * std + 0.0f;
  }
}

/* ------------------------------------------------------------------------ */
/* N2  MountainCar step        examples/mountain_car/env.py:12-38             */
/* ------------------------------------------------------------------------ */
typedef struct {
  float force_mag, goal_position, goal_velocity, gravity, max_position, max_speed, min_position;
} oracle_mountain_car_cfg;

/* state: SoA [2][n] rows position, velocity (updated in place); obs_out [n][2]
 * (the reference returns state.T); reward_out [n]. */
ORACLE_API void oracle_mountain_car_step(float *state, const int64_t *action,
                                         const oracle_mountain_car_cfg *cfg, float *obs_out,
                                         float *reward_out, int64_t n) {
  float *ps = state, *vs = state + n;
  for (int64_t i = 0; i < n; ++i) {
    float p = ps[i], v = vs[i];
    /* :29  velocity += (action - 1) * force_mag - gravity * cos(3 * position) */
    const float push = (float)(action[i] - 1) * cfg->force_mag;
    const float hill = cfg->gravity * cosf(3.0f * p);
    v = v + (push - hill);
    /* :30 */
    v = v < -cfg->max_speed ? -cfg->max_speed : (v > cfg->max_speed ? cfg->max_speed : v);
    /* :31-32 */
    p = p + v;
    p = p < cfg->min_position ? cfg->min_position : (p > cfg->max_position ? cfg->max_position : p);
    /* :33 */
    if (p == cfg->min_position && v < 0.0f) v = 0.0f;
    /* :35-37 */
    float r = fabsf(p - cfg->goal_position);
    r = r * -1.0f;
    if (p >= cfg->goal_position && v >= cfg->goal_velocity) r = 1.0f;
    ps[i] = p;
    vs[i] = v;
    obs_out[2 * i + 0] = p;
    obs_out[2 * i + 1] = v;
    reward_out[i] = r;
  }
}

/* MountainCar.reset  examples/mountain_car/env.py:96-104: position ~ N(-0.5, 0.05),
 * velocity ~ N(0, 0.05); build-owned noise: one Box-Muller pair per env from
 * counter (env, reset_count, STREAM_RESET, 0). */
ORACLE_API void oracle_mountain_car_reset(float *state, int64_t n, uint64_t seed,
                                          uint64_t reset_count, int64_t env_offset) {
  for (int64_t i = 0; i < n; ++i) {
    uint32_t r[4];
    float z0, z1;
    rl8_philox4x32_10(seed, (uint64_t)(i + env_offset), reset_count,
                      rl8_stream_block(RL8_STREAM_RESET, 0), r);
    rl8_box_muller(r[0], r[1], &z0, &z1);
    state[i] = z0 * 0.05f + -0.5f;
    state[n + i] = z1 * 0.05f + 0.0f;
  }
}

/* ------------------------------------------------------------------------ */
/* N2  Pendulum step           examples/pendulum/env.py:12-39                 */
/* ------------------------------------------------------------------------ */
/* gravity_coeff = 3g/(2l), torque_coeff = 3/(m l^2): the reference forms both in
 * double on the host and multiplies the fp32 tensors by them (:32). */
typedef struct {
  float dt, gravity_coeff, torque_coeff, max_speed, max_torque;
} oracle_pendulum_cfg;

/* torch.remainder on floats: fmod, then shifted into the divisor's sign. */
static float oracle_remainder(float a, float b) {
  float mod = fmodf(a, b);
  if (mod != 0.0f && ((b < 0.0f) != (mod < 0.0f))) mod += b;
  return mod;
}

/* state: SoA [2][n] rows th, thdot (updated in place); action [n] float;
 * obs_out [n][3] = cos th', sin th', thdot'; reward_out [n] = -costs(th, thdot, u). */
ORACLE_API void oracle_pendulum_step(float *state, const float *action,
                                     const oracle_pendulum_cfg *cfg, float *obs_out,
                                     float *reward_out, int64_t n) {
  const float pi = (float)3.141592653589793, two_pi = (float)(2 * 3.141592653589793);
  float *ths = state, *tds = state + n;
  for (int64_t i = 0; i < n; ++i) {
    const float th = ths[i], thdot = tds[i];
    /* :26 */
    float u = action[i];
    u = u < -cfg->max_torque ? -cfg->max_torque : (u > cfg->max_torque ? cfg->max_torque : u);
    /* :27-31 */
    const float ang = oracle_remainder(th + pi, two_pi) - pi;
    const float costs = (ang * ang + 0.1f * (thdot * thdot)) + 0.001f * (u * u);
    /* :33-35 */
    float nd = thdot + (cfg->gravity_coeff * sinf(th) + cfg->torque_coeff * u) * cfg->dt;
    nd = nd < -cfg->max_speed ? -cfg->max_speed : (nd > cfg->max_speed ? cfg->max_speed : nd);
    const float nth = th + nd * cfg->dt;
    ths[i] = nth;
    tds[i] = nd;
    obs_out[3 * i + 0] = cosf(nth);
    obs_out[3 * i + 1] = sinf(nth);
    obs_out[3 * i + 2] = nd;
    reward_out[i] = -costs;
  }
}

/* Pendulum.reset  examples/pendulum/env.py:101-113: th ~ U(-pi, pi), thdot ~ U(-1, 1)
 * (torch's uniform_: u * (b - a) + a on a 24-bit u); obs = cos, sin, thdot. */
ORACLE_API void oracle_pendulum_reset(float *state, float *obs_out, int64_t n, uint64_t seed,
                                      uint64_t reset_count, int64_t env_offset) {
  const float pi = (float)3.141592653589793;
  for (int64_t i = 0; i < n; ++i) {
    uint32_t r[4];
    rl8_philox4x32_10(seed, (uint64_t)(i + env_offset), reset_count,
                      rl8_stream_block(RL8_STREAM_RESET, 0), r);
    const float th = rl8_u01_24(r[0]) * (pi - (-pi)) + (-pi);
    const float td = rl8_u01_24(r[1]) * (1.0f - (-1.0f)) + (-1.0f);
    state[i] = th;
    state[n + i] = td;
    if (obs_out) {
      obs_out[3 * i + 0] = cosf(th);
      obs_out[3 * i + 1] = sinf(th);
      obs_out[3 * i + 2] = td;
    }
  }
}

/* ------------------------------------------------------------------------ */
/* a-3  Rollout bookkeeping    src/rl8/algorithms/_feedforward.py:378-393     */
/* ------------------------------------------------------------------------ */
/* rdr[:, t+1] = gamma * rdr[:, t] + r  (mul, then add: two roundings). */
ORACLE_API void oracle_rdr_step(const float *rdr_t, const float *reward, float *rdr_t1,
                                double gamma, int64_t n) {
  const float g = (float)gamma;
  for (int64_t i = 0; i < n; ++i) rdr_t1[i] = g * rdr_t[i] + reward[i];
}

/* ------------------------------------------------------------------------ */
/* a-4  Collect statistics     src/rl8/algorithms/_feedforward.py:411-436     */
/* ------------------------------------------------------------------------ */
/* rewards, rdr: env-major [n][h+1].  out[0..8] = returns min,max,mean,std,
 * rewards min,max,mean,std, std(rdr[:,1:]).  std is unbiased; torch's CPU
 * std_mean accumulates float data in double, as done here. */
ORACLE_API void oracle_rollout_stats(const float *rewards, const float *rdr, int64_t n,
                                     int64_t h, double *out) {
  double ret_min = INFINITY, ret_max = -INFINITY, rew_min = INFINITY, rew_max = -INFINITY;
  double ret_sum = 0, ret_sq = 0, rew_sum = 0, rew_sq = 0, rdr_sum = 0, rdr_sq = 0;
  const int64_t stride = h + 1;
  for (int64_t i = 0; i < n; ++i) {
    float ret = 0.0f; /* torch.sum(rewards[:, :-1], dim=1) in f32 */
    for (int64_t t = 0; t < h; ++t) {
      float r = rewards[i * stride + t];
      ret = ret + r;
      if (r < rew_min) rew_min = r;
      if (r > rew_max) rew_max = r;
      rew_sum += r;
    }
    if (ret < ret_min) ret_min = ret;
    if (ret > ret_max) ret_max = ret;
    ret_sum += ret;
  }
  const double ret_mean = ret_sum / (double)n;
  const double rew_mean = rew_sum / (double)(n * h);
  if (rdr)
    for (int64_t i = 0; i < n; ++i)
      for (int64_t t = 1; t <= h; ++t) rdr_sum += rdr[i * stride + t];
  const double rdr_mean = rdr_sum / (double)(n * h);
  for (int64_t i = 0; i < n; ++i) {
    float ret = 0.0f;
    for (int64_t t = 0; t < h; ++t) {
      float r = rewards[i * stride + t];
      ret = ret + r;
      rew_sq += ((double)r - rew_mean) * ((double)r - rew_mean);
      if (rdr) {
        double d = (double)rdr[i * stride + t + 1] - rdr_mean;
        rdr_sq += d * d;
      }
    }
    ret_sq += ((double)ret - ret_mean) * ((double)ret - ret_mean);
  }
  out[0] = ret_min; out[1] = ret_max; out[2] = ret_mean;
  out[3] = sqrt(ret_sq / (double)(n - 1));
  out[4] = rew_min; out[5] = rew_max; out[6] = rew_mean;
  out[7] = sqrt(rew_sq / (double)(n * h - 1));
  out[8] = rdr ? sqrt(rdr_sq / (double)(n * h - 1)) : 1.0;
}

/* ------------------------------------------------------------------------ */
/* a-5  generalized_advantage_estimate   src/rl8/nn/functional.py:50-123      */
/* ------------------------------------------------------------------------ */
/* rewards (in/out: scaled in place, :106), values, adv, ret: env-major
 * [n][h+1].  moments_out[0] = mean, [1] = std of adv[:, :h] (unbiased) as the
 * f32 values the normalisation used (0,1 when normalize == 0). */
ORACLE_API void oracle_gae(float *rewards, const float *values, float *adv, float *ret,
                           int64_t n, int64_t h, double gamma, double gae_lambda,
                           double reward_scale, int normalize, float *moments_out) {
  const int64_t stride = h + 1;
  const float denom = (float)(reward_scale + 1e-8); /* :106 python double, cast once */
  const float g = (float)gamma;
  const float gl = (float)(gamma * gae_lambda); /* :114 double product, cast once */
  for (int64_t i = 0; i < n * stride; ++i) rewards[i] = rewards[i] / denom;
  for (int64_t i = 0; i < n; ++i) {
    const float *r = rewards + i * stride, *v = values + i * stride;
    float *a = adv + i * stride, *q = ret + i * stride;
    float prev = 0.0f;
    a[h] = 0.0f; /* zeros_like, :105 */
    for (int64_t t = h - 1; t >= 0; --t) {
      float delta = r[t] + (g * v[t + 1] - v[t]); /* :109-112 */
      prev = delta + gl * prev;                    /* :113-115 */
      a[t] = prev;
    }
    for (int64_t t = 0; t <= h; ++t) q[t] = a[t] + v[t]; /* :117, all H+1 columns */
  }
  float mean_f = 0.0f, std_f = 1.0f;
  if (normalize) { /* :118-122 */
    double sum = 0.0, sq = 0.0;
    const double cnt = (double)(n * h);
    for (int64_t i = 0; i < n; ++i)
      for (int64_t t = 0; t < h; ++t) sum += adv[i * stride + t];
    const double mean = sum / cnt;
    for (int64_t i = 0; i < n; ++i)
      for (int64_t t = 0; t < h; ++t) {
        double d = (double)adv[i * stride + t] - mean;
        sq += d * d;
      }
    mean_f = (float)mean;
    std_f = (float)sqrt(sq / (cnt - 1.0));
    const float sd = std_f + 1e-8f;
    for (int64_t i = 0; i < n; ++i)
      for (int64_t t = 0; t < h; ++t)
        adv[i * stride + t] = (adv[i * stride + t] - mean_f) / sd;
  }
  if (moments_out) { moments_out[0] = mean_f; moments_out[1] = std_f; }
}

/* ------------------------------------------------------------------------ */
/* a-6  Distributions          src/rl8/distributions.py:98-170                */
/* ------------------------------------------------------------------------ */
/* torch.distributions.Categorical(logits=x): normalised logits nl = x - lse(x),
 * lse = max + log(sum(exp(x - max))); probs = softmax(nl); log_prob = nl[a];
 * entropy = -sum(p * nl). */
static inline void cat_normalise(const float *x, int k, float *nl, float *p, int cr) {
  float mx = x[0];
  for (int j = 1; j < k; ++j) mx = x[j] > mx ? x[j] : mx;
  float s = 0.0f;
  for (int j = 0; j < k; ++j) s += cr ? cr_expf(x[j] - mx) : expf(x[j] - mx);
  const float lse = mx + (cr ? cr_logf(s) : logf(s));
  float mx2 = -INFINITY;
  for (int j = 0; j < k; ++j) {
    nl[j] = x[j] - lse;
    mx2 = nl[j] > mx2 ? nl[j] : mx2;
  }
  float s2 = 0.0f;
  for (int j = 0; j < k; ++j) {
    p[j] = cr ? cr_expf(nl[j] - mx2) : expf(nl[j] - mx2);
    s2 += p[j];
  }
  for (int j = 0; j < k; ++j) p[j] = p[j] / s2;
}

#define ORACLE_MAX_CLASSES 64

/* Categorical.sample (distributions.py:121-122 -> torch.multinomial, 1 draw):
 * action = argmax_j p_j / q_j with q ~ Exp(1) (first index wins ties);
 * logp = sum over action dims of nl[action].  logits [m][a][k]; noise either
 * injected (q [m][a][k]) or drawn from Philox: counter (row, step, STREAM_ACTION,
 * block) -> word w = (dim*k + j); q = -log(u), u in (0,1]. */
ORACLE_API void oracle_categorical_sample(const float *logits, const float *q_in,
                                          int64_t *action, float *logp, int64_t m, int a,
                                          int k, uint64_t seed, uint64_t step,
                                          int64_t row_offset, int deterministic) {
  float nl[ORACLE_MAX_CLASSES], p[ORACLE_MAX_CLASSES];
  for (int64_t i = 0; i < m; ++i) {
    float lp = 0.0f;
    for (int d = 0; d < a; ++d) {
      const float *x = logits + (i * a + d) * k;
      cat_normalise(x, k, nl, p, 1);
      int best = 0;
      float best_v = -INFINITY;
      for (int j = 0; j < k; ++j) {
        float v;
        if (deterministic) {
          v = p[j]; /* Categorical.mode = argmax(probs) */
        } else {
          float q = q_in ? q_in[(i * a + d) * k + j]
                         : rl8_exponential(seed, (uint64_t)(i + row_offset), step,
                                           (uint32_t)(d * k + j));
          v = p[j] / q;
        }
        if (v > best_v) { best_v = v; best = j; }
      }
      action[i * a + d] = best;
      lp = (d == 0) ? nl[best] : lp + nl[best];
    }
    logp[i] = lp;
  }
}

static const float LOG_SQRT_2PI = 0.91893853320467267f; /* math.log(math.sqrt(2*pi)) */

static inline float normal_log_prob(float value, float loc, float scale) {
  /* torch.distributions.Normal.log_prob */
  float var = scale * scale;
  float log_scale = logf(scale);
  float d = value - loc;
  return ((-(d * d)) / (2.0f * var) - log_scale) - LOG_SQRT_2PI;
}

/* Normal / SquashedNormal sample + logp (distributions.py:135-170).
 * eps_in == NULL -> Philox Box-Muller normals, counter (row, step,
 * STREAM_ACTION, dim/2). */
ORACLE_API void oracle_normal_sample(const float *mean, const float *log_std,
                                     const float *eps_in, float *action, float *logp,
                                     int64_t m, int a, int squashed, uint64_t seed,
                                     uint64_t step, int64_t row_offset,
                                     int deterministic) {
  const float feps = 1.1920928955078125e-07f; /* torch.finfo(float32).eps */
  for (int64_t i = 0; i < m; ++i) {
    float lp = 0.0f, corr = 0.0f;
    for (int d = 0; d < a; ++d) {
      float mu = mean[i * a + d], sc = expf(log_std[i * a + d]);
      float e = 0.0f;
      if (!deterministic)
        e = eps_in ? eps_in[i * a + d]
                   : rl8_normal(seed, (uint64_t)(i + row_offset), step, (uint32_t)d);
      float raw = deterministic ? mu : mu + sc * e;
      float act = squashed ? tanhf(raw) : raw;
      action[i * a + d] = act;
      float l;
      if (squashed) { /* :160-168 */
        float c = fminf(fmaxf(act, -1.0f + feps), 1.0f - feps);
        float u = 0.5f * (log1pf(c) - log1pf(-c));
        l = normal_log_prob(u, mu, sc);
        l = fminf(fmaxf(l, -100.0f), 100.0f);
        float t = logf((1.0f - act * act) + feps);
        corr = (d == 0) ? t : corr + t;
      } else {
        l = normal_log_prob(act, mu, sc);
      }
      lp = (d == 0) ? l : lp + l;
    }
    logp[i] = squashed ? lp - corr : lp;
  }
}

/* ------------------------------------------------------------------------ */
/* a-7  ppo_losses + approximate KL + backward                               */
/*      src/rl8/nn/functional.py:259-363, algorithms/_feedforward.py:545-559 */
/* ------------------------------------------------------------------------ */
typedef struct {
  float clip_param;
  float dual_clip_param; /* <= 0 : disabled (reference: None / falsy) */
  float entropy_coeff;
  float vf_clip_param;
  float vf_coeff;
  float loss_scale; /* 1 / grad_accumulation_steps (:545) */
} oracle_ppo_hparams;

/* Per-sample pieces shared by the three distributions.  Returns the policy
 * term and d(policy term)/d(logp_new). */
static inline void ppo_policy_term(float logp_new, float logp_old, float adv,
                                   const oracle_ppo_hparams *hp, float *term,
                                   float *dterm_dlogp, float *kl_term) {
  const float lr = logp_new - logp_old;
  const float ratio = expf(lr);                                  /* :316-319 */
  const float lo = 1.0f - hp->clip_param, hi = 1.0f + hp->clip_param;
  const float clamped = fminf(fmaxf(ratio, lo), hi);
  const float s1 = adv * ratio;                                  /* :331 */
  const float s2 = adv * clamped;                                /* :332-334 */
  const float inside = (ratio >= lo && ratio <= hi) ? 1.0f : 0.0f; /* clamp bwd */
  /* torch.min(a, b) backward: all to the smaller, split evenly on ties. */
  float w1, w2;
  if (s1 < s2) { w1 = 1.0f; w2 = 0.0f; }
  else if (s1 > s2) { w1 = 0.0f; w2 = 1.0f; }
  else { w1 = 0.5f; w2 = 0.5f; }
  float clip1 = s1 < s2 ? s1 : s2;
  float dclip1_dratio = w1 * adv + w2 * adv * inside;
  float out = clip1, dout = dclip1_dratio;
  if (hp->dual_clip_param > 0.0f) {                              /* :335-343 */
    if (adv < 0.0f) {
      const float floor_ = hp->dual_clip_param * adv;
      if (clip1 > floor_) { out = clip1; dout = dclip1_dratio; }
      else if (clip1 < floor_) { out = floor_; dout = 0.0f; }
      else { out = clip1; dout = 0.5f * dclip1_dratio; }
    }
  }
  *term = out;
  *dterm_dlogp = dout * ratio;
  *kl_term = (ratio - 1.0f) - lr;                                /* _feedforward.py:558 */
}

static inline void ppo_vf_term(float value, float ret, const oracle_ppo_hparams *hp,
                               float *term, float *dterm_dvalue) {
  /* F.smooth_l1_loss(beta=1, reduction="none") then clamp(0, vf_clip) :320-330 */
  const float d = value - ret;
  const float ad = fabsf(d);
  float l, dl;
  if (ad < 1.0f) { l = 0.5f * d * d; dl = d; }
  else { l = ad - 0.5f; dl = d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f); }
  if (l > hp->vf_clip_param) { *term = hp->vf_clip_param; *dterm_dvalue = 0.0f; }
  else if (l < 0.0f) { *term = 0.0f; *dterm_dvalue = 0.0f; }
  else { *term = l; *dterm_dvalue = dl; }
}

/* Output convention for all three:
 *   sums[0..4] = sum over samples of entropy, policy, vf terms, (unused), kl
 *   losses[0..4] = entropy, policy, vf, total, kl   (already * loss_scale,
 *                  except kl which the reference leaves unscaled, :557-559)
 *   grad_* = d(total * loss_scale)/d(input), i.e. what autograd leaves in
 *            features.grad / values.grad after (losses["total"]).backward().  */
static void ppo_finish(const double *sums, int64_t m, const oracle_ppo_hparams *hp,
                       double *losses) {
  const double inv = 1.0 / (double)m;
  const double ent = hp->entropy_coeff != 0.0f ? sums[0] * inv : 0.0;
  const double pol = sums[1] * inv, vf = sums[2] * inv;
  double total = (double)hp->vf_coeff * vf - pol;
  if (hp->entropy_coeff != 0.0f) total -= (double)hp->entropy_coeff * ent;
  losses[0] = ent * hp->loss_scale;
  losses[1] = pol * hp->loss_scale;
  losses[2] = vf * hp->loss_scale;
  losses[3] = total * hp->loss_scale;
  losses[4] = sums[4] * inv;
}

ORACLE_API void oracle_ppo_loss_categorical(
    const float *logits, const float *values, const int64_t *actions,
    const float *logp_old, const float *adv, const float *returns, int64_t m, int a,
    int k, const oracle_ppo_hparams *hp, float *grad_logits, float *grad_values,
    double *losses) {
  double sums[5] = {0, 0, 0, 0, 0};
  const float gscale = hp->loss_scale / (float)m;
  float nl[ORACLE_MAX_CLASSES], p[ORACLE_MAX_CLASSES];
  for (int64_t i = 0; i < m; ++i) {
    float logp = 0.0f, ent = 0.0f;
    for (int d = 0; d < a; ++d) {
      cat_normalise(logits + (i * a + d) * k, k, nl, p, 0);
      float l = nl[actions[i * a + d]];
      logp = d == 0 ? l : logp + l;
      float e = 0.0f;
      for (int j = 0; j < k; ++j) e += nl[j] * p[j]; /* Categorical.entropy */
      ent = d == 0 ? -e : ent + (-e);
    }
    float term, dterm, klt, vterm, dv;
    ppo_policy_term(logp, logp_old[i], adv[i], hp, &term, &dterm, &klt);
    ppo_vf_term(values[i], returns[i], hp, &vterm, &dv);
    sums[0] += ent; sums[1] += term; sums[2] += vterm; sums[4] += klt;
    if (grad_values) grad_values[i] = gscale * hp->vf_coeff * dv;
    if (grad_logits) {
      for (int d = 0; d < a; ++d) {
        cat_normalise(logits + (i * a + d) * k, k, nl, p, 0);
        float h = 0.0f;
        for (int j = 0; j < k; ++j) h += nl[j] * p[j];
        h = -h;
        const int64_t act = actions[i * a + d];
        for (int j = 0; j < k; ++j) {
          float dlogp = (j == act ? 1.0f : 0.0f) - p[j];
          float g = -dterm * dlogp; /* total = ... - policy */
          if (hp->entropy_coeff != 0.0f) {
            float dent = -p[j] * (nl[j] + h); /* dH/dx_j */
            g -= hp->entropy_coeff * dent;
          }
          grad_logits[(i * a + d) * k + j] = gscale * g;
        }
      }
    }
  }
  ppo_finish(sums, m, hp, losses);
}

/* Normal (squashed == 0) and SquashedNormal (squashed == 1). */
ORACLE_API void oracle_ppo_loss_normal(
    const float *mean, const float *log_std, const float *values, const float *actions,
    const float *logp_old, const float *adv, const float *returns, int64_t m, int a,
    int squashed, const oracle_ppo_hparams *hp, float *grad_mean, float *grad_log_std,
    float *grad_values, double *losses) {
  double sums[5] = {0, 0, 0, 0, 0};
  const float gscale = hp->loss_scale / (float)m;
  const float feps = 1.1920928955078125e-07f;
  const float HALF_LOG_2PI_PLUS_HALF = 1.4189385332046727f; /* 0.5 + 0.5*log(2*pi) */
  for (int64_t i = 0; i < m; ++i) {
    float logp = 0.0f, corr = 0.0f, ent = 0.0f;
    for (int d = 0; d < a; ++d) {
      const float mu = mean[i * a + d], ls = log_std[i * a + d];
      const float sc = expf(ls), act = actions[i * a + d];
      float l;
      if (squashed) {
        float c = fminf(fmaxf(act, -1.0f + feps), 1.0f - feps);
        float u = 0.5f * (log1pf(c) - log1pf(-c));
        l = fminf(fmaxf(normal_log_prob(u, mu, sc), -100.0f), 100.0f);
        float t = logf((1.0f - act * act) + feps);
        corr = d == 0 ? t : corr + t;
      } else {
        l = normal_log_prob(act, mu, sc);
        float e = HALF_LOG_2PI_PLUS_HALF + logf(sc); /* Normal.entropy */
        ent = d == 0 ? e : ent + e;
      }
      logp = d == 0 ? l : logp + l;
    }
    if (squashed) logp = logp - corr;
    float term, dterm, klt, vterm, dv;
    ppo_policy_term(logp, logp_old[i], adv[i], hp, &term, &dterm, &klt);
    ppo_vf_term(values[i], returns[i], hp, &vterm, &dv);
    sums[0] += ent; sums[1] += term; sums[2] += vterm; sums[4] += klt;
    if (grad_values) grad_values[i] = gscale * hp->vf_coeff * dv;
    if (grad_mean) {
      for (int d = 0; d < a; ++d) {
        const float mu = mean[i * a + d], ls = log_std[i * a + d];
        const float sc = expf(ls), act = actions[i * a + d];
        float x = act, pass = 1.0f;
        if (squashed) {
          float c = fminf(fmaxf(act, -1.0f + feps), 1.0f - feps);
          x = 0.5f * (log1pf(c) - log1pf(-c));
          float l = normal_log_prob(x, mu, sc);
          pass = (l >= -100.0f && l <= 100.0f) ? 1.0f : 0.0f;
        }
        const float z = (x - mu) / sc;
        const float dlogp_dmu = pass * (z / sc);
        const float dlogp_dls = pass * (z * z - 1.0f);
        float gm = -dterm * dlogp_dmu;
        float gs = -dterm * dlogp_dls;
        if (!squashed && hp->entropy_coeff != 0.0f) gs -= hp->entropy_coeff * 1.0f;
        grad_mean[i * a + d] = gscale * gm;
        grad_log_std[i * a + d] = gscale * gs;
      }
    }
  }
  ppo_finish(sums, m, hp, losses);
}

/* ------------------------------------------------------------------------ */
/* a-8  Batcher gather         src/rl8/_utils.py:211-225                      */
/* ------------------------------------------------------------------------ */
ORACLE_API void oracle_gather_rows(const int64_t *index, const void *src, void *dst,
                                   int64_t m, int64_t row_bytes) {
  for (int64_t i = 0; i < m; ++i)
    memcpy((char *)dst + i * row_bytes, (const char *)src + index[i] * row_bytes,
           (size_t)row_bytes);
}

/* Build-owned permutation for minibatch shuffling when no permutation is
 * injected: Fisher-Yates driven by Philox (counter (i, iteration,
 * STREAM_SHUFFLE, 0)), identical on host and device-side callers because it is
 * generated on the host in both. */
ORACLE_API void oracle_permutation(int64_t *out, int64_t m, uint64_t seed,
                                   uint64_t iteration) {
  for (int64_t i = 0; i < m; ++i) out[i] = i;
  for (int64_t i = m - 1; i > 0; --i) {
    uint32_t r[4];
    rl8_philox4x32_10(seed, (uint64_t)i, iteration,
                      rl8_stream_block(RL8_STREAM_SHUFFLE, 0), r);
    uint64_t x = ((uint64_t)r[0] << 32) | r[1];
    int64_t j = (int64_t)(x % (uint64_t)(i + 1));
    int64_t tmp = out[i]; out[i] = out[j]; out[j] = tmp;
  }
}

/* Raw noise, exposed so tests can check the device generator word-for-word. */
ORACLE_API void oracle_philox_words(uint64_t seed, uint64_t c0, uint64_t c1, uint32_t stream,
                                    uint32_t *out4) {
  rl8_philox4x32_10(seed, c0, c1, stream, out4);
}

ORACLE_API float oracle_exponential(uint64_t seed, uint64_t row, uint64_t step, uint32_t w) {
  return rl8_exponential(seed, row, step, w);
}

ORACLE_API float oracle_normal(uint64_t seed, uint64_t row, uint64_t step, uint32_t w) {
  return rl8_normal(seed, row, step, w);
}
