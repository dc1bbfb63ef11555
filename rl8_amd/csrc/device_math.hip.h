// Per-element device arithmetic of the PPO path, written once and inlined into
// the standalone and the fused kernels alike.  Compiled with -ffp-contract=off:
// every reference op rounds once, nothing is fused into FMA.
#pragma once

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "rl8_amd.h"
#include "rl8_philox.h"

namespace rl8 {

// exp / log evaluated in fp64 and rounded once: host (oracle) and device then
// agree bit-for-bit, which is what makes sampled action INDICES reproducible.
__device__ __forceinline__ float cr_expf(float x) { return (float)exp((double)x); }
__device__ __forceinline__ float cr_logf(float x) { return (float)log((double)x); }

// ---- DummyEnv.step (src/rl8/env.py:224-230, 253-259) ----------------------
__device__ __forceinline__ float dummy_step_discrete(float state, int64_t action) {
  return state + (float)(2 * action - 1);
}
__device__ __forceinline__ float dummy_step_continuous(float state, float action) {
  return state + action;
}

// ---- CartPole physics (examples/cartpole/env.py:30-63) --------------------
// MountainCar (examples/mountain_car/env.py:29-37), one env; returns the reward.
__device__ __forceinline__ float mountain_car_advance(float &p, float &v, int64_t action,
                                                      const rl8_mountain_car_cfg &c) {
  const float push = (float)(action - 1) * c.force_mag;
  const float hill = c.gravity * cosf(3.0f * p);
  v = v + (push - hill);
  v = v < -c.max_speed ? -c.max_speed : (v > c.max_speed ? c.max_speed : v);
  p = p + v;
  p = p < c.min_position ? c.min_position : (p > c.max_position ? c.max_position : p);
  if (p == c.min_position && v < 0.0f) v = 0.0f;
  float r = fabsf(p - c.goal_position) * -1.0f;
  if (p >= c.goal_position && v >= c.goal_velocity) r = 1.0f;
  return r;
}

// torch.remainder on floats: fmod, then shifted into the divisor's sign.
__device__ __forceinline__ float torch_remainder(float a, float b) {
  float mod = fmodf(a, b);
  if (mod != 0.0f && ((b < 0.0f) != (mod < 0.0f))) mod += b;
  return mod;
}

// Pendulum (examples/pendulum/env.py:26-39), one env; returns the reward (minus
// the cost of the state and torque before the step).
__device__ __forceinline__ float pendulum_advance(float &th, float &thdot, float action,
                                                  const rl8_pendulum_cfg &c) {
  const float pi = (float)3.141592653589793, two_pi = (float)(2 * 3.141592653589793);
  const float u = action < -c.max_torque ? -c.max_torque : (action > c.max_torque ? c.max_torque : action);
  const float ang = torch_remainder(th + pi, two_pi) - pi;
  const float costs = (ang * ang + 0.1f * (thdot * thdot)) + 0.001f * (u * u);
  float nd = thdot + (c.gravity_coeff * sinf(th) + c.torque_coeff * u) * c.dt;
  nd = nd < -c.max_speed ? -c.max_speed : (nd > c.max_speed ? c.max_speed : nd);
  th = th + nd * c.dt;
  thdot = nd;
  return -costs;
}

struct CartPoleState {
  float x, x_dot, theta, theta_dot;
};
struct CartPoleOut {
  float cos_theta, sin_theta, reward;
};

__device__ __forceinline__ CartPoleOut cartpole_advance(CartPoleState &s, int64_t action,
                                                        const rl8_cartpole_cfg &c) {
  const float force = (float)(action - 1) * c.force_mag;
  const float costheta = cosf(s.theta), sintheta = sinf(s.theta);
  const float tmp =
      (force + (c.pole_mass_length * (s.theta_dot * s.theta_dot)) * sintheta) / c.total_mass;
  const float four_thirds = (float)(4.0 / 3.0);
  const float theta_acc =
      (c.gravity * sintheta - costheta * tmp) /
      (c.length * (four_thirds - (c.pole_mass * (costheta * costheta)) / c.total_mass));
  const float x_acc = tmp - ((c.pole_mass_length * theta_acc) * costheta) / c.total_mass;
  if (!c.semi_implicit) {
    s.x = s.x + c.tau * s.x_dot;
    s.x_dot = s.x_dot + c.tau * x_acc;
    s.theta = s.theta + c.tau * s.theta_dot;
    s.theta_dot = s.theta_dot + c.tau * theta_acc;
  } else {
    s.x_dot = s.x_dot + c.tau * x_acc;
    s.x = s.x + c.tau * s.x_dot;
    s.theta_dot = s.theta_dot + c.tau * theta_acc;
    s.theta = s.theta + c.tau * s.theta_dot;
  }
  CartPoleOut o;
  o.cos_theta = cosf(s.theta);
  o.sin_theta = sinf(s.theta);
  const float theta_error = fabsf(o.cos_theta - 1.0f) + fabsf(o.sin_theta - 0.0f);
  const float other = (fabsf(s.x) + fabsf(s.x_dot)) + fabsf(s.theta_dot);
  o.reward = -(theta_error + other);
  return o;
}

// ---- Categorical (torch.distributions.Categorical(logits=x)) ---------------
// nl = x - logsumexp(x); p = softmax(nl).  CR = correctly-rounded transcendentals
// (sampler); otherwise the fast fp32 ones (loss kernel, tolerance-checked).
template <int K, bool CR>
__device__ __forceinline__ void categorical_normalise(const float (&x)[K], float (&nl)[K],
                                                      float (&p)[K]) {
  float mx = x[0];
#pragma unroll
  for (int j = 1; j < K; ++j) mx = x[j] > mx ? x[j] : mx;
  float s = 0.0f;
#pragma unroll
  for (int j = 0; j < K; ++j) s += CR ? cr_expf(x[j] - mx) : expf(x[j] - mx);
  const float lse = mx + (CR ? cr_logf(s) : logf(s));
  float mx2 = -INFINITY;
#pragma unroll
  for (int j = 0; j < K; ++j) {
    nl[j] = x[j] - lse;
    mx2 = nl[j] > mx2 ? nl[j] : mx2;
  }
  float s2 = 0.0f;
#pragma unroll
  for (int j = 0; j < K; ++j) {
    p[j] = CR ? cr_expf(nl[j] - mx2) : expf(nl[j] - mx2);
    s2 += p[j];
  }
#pragma unroll
  for (int j = 0; j < K; ++j) p[j] = p[j] / s2;
}

// Loss-kernel variant (tolerance-checked, 1e-5): the hardware exp2 / log2 /
// rcp units, and p_j = e_j / sum(e) from the first pass -- the reference's second
// softmax over the already-normalised logits renormalises by 1 +- 1e-7, which is
// dropped here.  ~4x fewer instructions per sample than the exact-order variant,
// which keeps the fused loss kernel HBM-bound rather than ALU-bound.
template <int K>
__device__ __forceinline__ void categorical_normalise_fast(const float (&x)[K], float (&nl)[K],
                                                           float (&p)[K]) {
  float mx = x[0];
#pragma unroll
  for (int j = 1; j < K; ++j) mx = fmaxf(mx, x[j]);
  float s = 0.0f;
#pragma unroll
  for (int j = 0; j < K; ++j) {
    p[j] = __expf(x[j] - mx);
    s += p[j];
  }
  const float lse = mx + __logf(s);
  const float inv = __builtin_amdgcn_rcpf(s);
#pragma unroll
  for (int j = 0; j < K; ++j) {
    nl[j] = x[j] - lse;
    p[j] = p[j] * inv;
  }
}

// Runtime-K variant (K <= RL8_MAX_CLASSES) for the generic kernels.
template <bool CR>
__device__ __forceinline__ void categorical_normalise_dyn(const float *x, int k, float *nl,
                                                          float *p) {
  float mx = x[0];
  for (int j = 1; j < k; ++j) mx = x[j] > mx ? x[j] : mx;
  float s = 0.0f;
  for (int j = 0; j < k; ++j) s += CR ? cr_expf(x[j] - mx) : expf(x[j] - mx);
  const float lse = mx + (CR ? cr_logf(s) : logf(s));
  float mx2 = -INFINITY;
  for (int j = 0; j < k; ++j) {
    nl[j] = x[j] - lse;
    mx2 = nl[j] > mx2 ? nl[j] : mx2;
  }
  float s2 = 0.0f;
  for (int j = 0; j < k; ++j) {
    p[j] = CR ? cr_expf(nl[j] - mx2) : expf(nl[j] - mx2);
    s2 += p[j];
  }
  for (int j = 0; j < k; ++j) p[j] = p[j] / s2;
}

// One action dim of Categorical.sample (== torch.multinomial, one draw):
// argmax_j p_j / q_j, first index wins ties.  Returns the class; *logp = nl[class].
template <int K>
__device__ __forceinline__ int categorical_draw(const float (&x)[K], const float *q_injected,
                                                uint64_t seed, uint64_t row, uint64_t step,
                                                uint32_t word0, bool deterministic, float *logp) {
  float nl[K], p[K];
  categorical_normalise<K, true>(x, nl, p);
  int best = 0;
  float best_v = -INFINITY;
#pragma unroll
  for (int j = 0; j < K; ++j) {
    float v;
    if (deterministic) {
      v = p[j];
    } else {
      const float q = q_injected ? q_injected[j] : rl8_exponential(seed, row, step, word0 + j);
      v = p[j] / q;
    }
    if (v > best_v) {
      best_v = v;
      best = j;
    }
  }
  float l = nl[0];
#pragma unroll
  for (int j = 1; j < K; ++j) l = (best == j) ? nl[j] : l;
  *logp = l;
  return best;
}

// ---- Normal / SquashedNormal (src/rl8/distributions.py:135-170) ------------
constexpr float kLogSqrt2Pi = 0.91893853320467267f;
constexpr float kNormalEntropyConst = 1.4189385332046727f;  // 0.5 + 0.5*log(2*pi)
constexpr float kF32Eps = 1.1920928955078125e-07f;

__device__ __forceinline__ float normal_log_prob(float value, float loc, float scale) {
  const float var = scale * scale;
  const float d = value - loc;
  return ((-(d * d)) / (2.0f * var) - logf(scale)) - kLogSqrt2Pi;
}

// atanh through the reference's formula: 0.5 * (log1p(c) - log1p(-c)) on the
// clamped sample (:161-162).
__device__ __forceinline__ float squashed_invert(float act) {
  const float c = fminf(fmaxf(act, -1.0f + kF32Eps), 1.0f - kF32Eps);
  return 0.5f * (log1pf(c) - log1pf(-c));
}

// One action dim: returns the action, accumulates the two logp pieces.
__device__ __forceinline__ float normal_draw(float mu, float log_std, float eps, bool squashed,
                                             bool deterministic, float *lp, float *corr) {
  const float sc = expf(log_std);
  const float raw = deterministic ? mu : mu + sc * eps;
  const float act = squashed ? tanhf(raw) : raw;
  if (squashed) {
    float l = normal_log_prob(squashed_invert(act), mu, sc);
    *lp = fminf(fmaxf(l, -100.0f), 100.0f);
    *corr = logf((1.0f - act * act) + kF32Eps);
  } else {
    *lp = normal_log_prob(act, mu, sc);
    *corr = 0.0f;
  }
  return act;
}

// ---- PPO per-sample terms (src/rl8/nn/functional.py:316-349) --------------
struct PolicyTerm {
  float term;         // min(s1, s2) / dual-clipped variant
  float dterm_dlogp;  // derivative wrt logp_new
  float kl;           // (ratio - 1) - log_ratio
};

__device__ __forceinline__ PolicyTerm ppo_policy_term(float logp_new, float logp_old, float adv,
                                                      const rl8_ppo_hparams &hp) {
  const float lr = logp_new - logp_old;
  const float ratio = __expf(lr);
  const float lo = 1.0f - hp.clip_param, hi = 1.0f + hp.clip_param;
  const float clamped = fminf(fmaxf(ratio, lo), hi);
  const float s1 = adv * ratio;
  const float s2 = adv * clamped;
  const float inside = (ratio >= lo && ratio <= hi) ? 1.0f : 0.0f;
  // torch.min backward: everything to the smaller operand, halves on a tie.
  const float w1 = s1 < s2 ? 1.0f : (s1 > s2 ? 0.0f : 0.5f);
  const float w2 = 1.0f - w1;
  const float clip1 = s1 < s2 ? s1 : s2;
  const float dclip1 = w1 * adv + w2 * adv * inside;
  float out = clip1, dout = dclip1;
  if (hp.dual_clip_param > 0.0f && adv < 0.0f) {
    const float floor_ = hp.dual_clip_param * adv;
    if (clip1 < floor_) {
      out = floor_;
      dout = 0.0f;
    } else if (clip1 == floor_) {
      dout = 0.5f * dclip1;
    }
  }
  PolicyTerm r;
  r.term = out;
  r.dterm_dlogp = dout * ratio;
  r.kl = (ratio - 1.0f) - lr;
  return r;
}

// clamp(smooth_l1(value, ret, beta=1), 0, vf_clip) and its derivative.
__device__ __forceinline__ float ppo_vf_term(float value, float ret, const rl8_ppo_hparams &hp,
                                             float *dterm) {
  const float d = value - ret;
  const float ad = fabsf(d);
  float l, dl;
  if (ad < 1.0f) {
    l = 0.5f * d * d;
    dl = d;
  } else {
    l = ad - 0.5f;
    dl = d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f);
  }
  if (l > hp.vf_clip_param) {
    *dterm = 0.0f;
    return hp.vf_clip_param;
  }
  *dterm = dl;
  return l;
}

}  // namespace rl8
