// K4: fused PPO loss, forward and backward in one pass.
// Restates src/rl8/nn/functional.py:316-363 (clipped / dual-clipped surrogate,
// clamped Huber value loss, entropy bonus), the approximate-KL monitor of
// src/rl8/algorithms/_feedforward.py:552-559, the distribution log-prob /
// entropy of src/rl8/distributions.py:113-170, and what autograd would leave in
// features.grad / values.grad after `total.backward()`.
//
// HBM-bound, no reuse: each sample is read once (features, value, action,
// logp_old, advantage, return) and its two gradients written once -- 44 B per
// sample for Categorical(K=2).  The ~50 eager launches + autograd graph of the
// reference collapse into one launch; loss terms are accumulated per thread in
// fp64, reduced with wave shuffles and written as per-block partials that a
// one-block kernel sums in a fixed order (bitwise reproducible, no atomics).
//
// The gradient scale 1/(M_global * grad_accumulation_steps) is known before the
// launch, so gradients do not wait for the reduced loss.
#include <stdlib.h>

#include "common.hip.h"
#include "device_math.hip.h"

namespace rl8 {

constexpr int kLossCols = 4;  // entropy, policy, vf, kl

// Publishes this block's row; the last block to arrive sums rows [0, rows) in
// order and writes out[0..4] = sum entropy, policy, vf, sample count, sum kl.
__device__ __forceinline__ void publish_loss_row(double (&acc)[kLossCols], double *partials,
                                                 int row, int rows, double count, double *out,
                                                 double *smem) {
  if (threadIdx.x == 0) {
#pragma unroll
    for (int c = 0; c < kLossCols; ++c)
      publish_partial(partials + (int64_t)row * kPartialWidth + c, acc[c]);
  }
  if (!last_block_arrives(ticket_word(partials))) return;
  double tot[kLossCols] = {0.0, 0.0, 0.0, 0.0};
  fold_partial_rows<kLossCols>(partials, rows, [&](const double(&v)[kLossCols]) {
#pragma unroll
    for (int c = 0; c < kLossCols; ++c) tot[c] += v[c];
  });
  block_reduce<kLossCols, SumOp>(tot, smem);
  if (threadIdx.x == 0) {
    out[0] = tot[0];
    out[1] = tot[1];
    out[2] = tot[2];
    out[3] = count;
    out[4] = tot[3];
    *ticket_word(partials) = 0u;
  }
}

// ---------------------------------------------------------------------------
// Categorical, A == 1, compile-time K, SPT samples per thread with 16-byte
// loads wherever SPT*K*4 and SPT*4 are multiples of 16.
// ---------------------------------------------------------------------------
template <int K, int SPT, bool HAS_GRAD>
__global__ __launch_bounds__(kBlock) void ppo_loss_categorical_kernel(
    const float *__restrict__ logits, const float *__restrict__ value,
    const int64_t *__restrict__ action, const float *__restrict__ logp_old,
    const float *__restrict__ adv, const float *__restrict__ ret, int64_t m, rl8_ppo_hparams hp,
    float *__restrict__ grad_logits, float *__restrict__ grad_value,
    double *__restrict__ partials, int extra_rows, double *__restrict__ sums_out) {
  __shared__ double smem[kLossCols * kWavesPerBlock];
  double acc[kLossCols] = {0.0, 0.0, 0.0, 0.0};
  const bool with_entropy = hp.entropy_coeff != 0.0f;
  const int64_t groups = m / SPT;  // full groups; the tail went to a generic launch
  const int64_t stride = (int64_t)gridDim.x * kBlock;
#pragma unroll 2
  for (int64_t g = (int64_t)blockIdx.x * kBlock + threadIdx.x; g < groups; g += stride) {
    const int64_t i0 = g * SPT;
    float x[SPT * K], v[SPT], lo[SPT], ad[SPT], rt[SPT];
    int64_t act[SPT];
    if constexpr (SPT == 4) {
      // SPT*K floats = K float4; the others one float4 / two 16-byte loads.
      const float4 *xp = reinterpret_cast<const float4 *>(logits + i0 * K);
#pragma unroll
      for (int q = 0; q < K; ++q) {
        const float4 t = xp[q];
        x[4 * q + 0] = t.x; x[4 * q + 1] = t.y; x[4 * q + 2] = t.z; x[4 * q + 3] = t.w;
      }
      const float4 tv = *reinterpret_cast<const float4 *>(value + i0);
      const float4 tl = *reinterpret_cast<const float4 *>(logp_old + i0);
      const float4 ta = *reinterpret_cast<const float4 *>(adv + i0);
      const float4 tr = *reinterpret_cast<const float4 *>(ret + i0);
      v[0] = tv.x; v[1] = tv.y; v[2] = tv.z; v[3] = tv.w;
      lo[0] = tl.x; lo[1] = tl.y; lo[2] = tl.z; lo[3] = tl.w;
      ad[0] = ta.x; ad[1] = ta.y; ad[2] = ta.z; ad[3] = ta.w;
      rt[0] = tr.x; rt[1] = tr.y; rt[2] = tr.z; rt[3] = tr.w;
      const longlong2 a0 = *reinterpret_cast<const longlong2 *>(action + i0);
      const longlong2 a1 = *reinterpret_cast<const longlong2 *>(action + i0 + 2);
      act[0] = a0.x; act[1] = a0.y; act[2] = a1.x; act[3] = a1.y;
    } else {
#pragma unroll
      for (int s = 0; s < SPT; ++s) {
#pragma unroll
        for (int j = 0; j < K; ++j) x[s * K + j] = logits[(i0 + s) * K + j];
        v[s] = value[i0 + s]; lo[s] = logp_old[i0 + s]; ad[s] = adv[i0 + s];
        rt[s] = ret[i0 + s]; act[s] = action[i0 + s];
      }
    }
    float gx[SPT * K], gv[SPT];
    float part[kLossCols] = {0.0f, 0.0f, 0.0f, 0.0f};  // this group's terms, f32
#pragma unroll
    for (int s = 0; s < SPT; ++s) {
      float xs[K], nl[K], p[K];
#pragma unroll
      for (int j = 0; j < K; ++j) xs[j] = x[s * K + j];
      categorical_normalise_fast<K>(xs, nl, p);
      float logp = nl[0];
#pragma unroll
      for (int j = 1; j < K; ++j) logp = (act[s] == j) ? nl[j] : logp;
      float ent = 0.0f;
      if (with_entropy) {
        float e = 0.0f;
#pragma unroll
        for (int j = 0; j < K; ++j) e += nl[j] * p[j];
        ent = -e;
      }
      const PolicyTerm pt = ppo_policy_term(logp, lo[s], ad[s], hp);
      float dv;
      const float vterm = ppo_vf_term(v[s], rt[s], hp, &dv);
      part[0] += ent;
      part[1] += pt.term;
      part[2] += vterm;
      part[3] += pt.kl;
      if (HAS_GRAD) {
        gv[s] = hp.grad_scale * hp.vf_coeff * dv;
#pragma unroll
        for (int j = 0; j < K; ++j) {
          const float dlogp = (act[s] == j ? 1.0f : 0.0f) - p[j];
          float gj = -pt.dterm_dlogp * dlogp;
          if (with_entropy) gj -= hp.entropy_coeff * (-p[j] * (nl[j] + ent));
          gx[s * K + j] = hp.grad_scale * gj;
        }
        if constexpr (K == 2) {
          // A two-way categorical depends on the logit DIFFERENCE only, so its two gradients are
          // exact negatives in real arithmetic; the two fp32 evaluations above differ in their last
          // bits.  Their antisymmetric mean is as close to either and makes the property exact,
          // which the weight-gradient kernel of a two-output head relies on (rl8_mlp_wgrad_fused_pair_f32:
          // dZ2 = gate * g0 * (W3[0] - W3[1]), one binary operand instead of a dense one).
          const float g = 0.5f * (gx[s * K] - gx[s * K + 1]);
          gx[s * K] = g;
          gx[s * K + 1] = -g;
        }
      }
    }
#pragma unroll
    for (int c = 0; c < kLossCols; ++c) acc[c] += (double)part[c];
    if (HAS_GRAD) {
      if constexpr (SPT == 4) {
        float4 *gp = reinterpret_cast<float4 *>(grad_logits + i0 * K);
#pragma unroll
        for (int q = 0; q < K; ++q)
          gp[q] = make_float4(gx[4 * q + 0], gx[4 * q + 1], gx[4 * q + 2], gx[4 * q + 3]);
        *reinterpret_cast<float4 *>(grad_value + i0) = make_float4(gv[0], gv[1], gv[2], gv[3]);
      } else {
#pragma unroll
        for (int s = 0; s < SPT; ++s) {
#pragma unroll
          for (int j = 0; j < K; ++j) grad_logits[(i0 + s) * K + j] = gx[s * K + j];
          grad_value[i0 + s] = gv[s];
        }
      }
    }
  }
  block_reduce<kLossCols, SumOp>(acc, smem);
  publish_loss_row(acc, partials, blockIdx.x, gridDim.x + extra_rows, (double)m, sums_out, smem);
}

// ---------------------------------------------------------------------------
// Categorical, any A and K <= RL8_MAX_CLASSES: one sample per thread, scalar
// access.  Also mops up the (m % SPT) tail of the vector kernel via `first`.
// ---------------------------------------------------------------------------
template <bool HAS_GRAD>
__global__ __launch_bounds__(kBlock) void ppo_loss_categorical_generic_kernel(
    const float *__restrict__ logits, const float *__restrict__ value,
    const int64_t *__restrict__ action, const float *__restrict__ logp_old,
    const float *__restrict__ adv, const float *__restrict__ ret, int64_t first, int64_t m, int a,
    int k, rl8_ppo_hparams hp, float *__restrict__ grad_logits, float *__restrict__ grad_value,
    double *__restrict__ partials, int partial_row0, double *__restrict__ sums_out) {
  __shared__ double smem[kLossCols * kWavesPerBlock];
  double acc[kLossCols] = {0.0, 0.0, 0.0, 0.0};
  const bool with_entropy = hp.entropy_coeff != 0.0f;
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  float nl[RL8_MAX_CLASSES], p[RL8_MAX_CLASSES];
  for (int64_t i = first + (int64_t)blockIdx.x * kBlock + threadIdx.x; i < m; i += stride) {
    float logp = 0.0f, ent = 0.0f;
    for (int d = 0; d < a; ++d) {
      categorical_normalise_dyn<false>(logits + (i * a + d) * k, k, nl, p);
      int64_t ai = action[i * a + d];
      ai = ai < 0 ? 0 : (ai >= k ? k - 1 : ai);
      const float l = nl[ai];
      logp = d == 0 ? l : logp + l;
      if (with_entropy) {
        float e = 0.0f;
        for (int j = 0; j < k; ++j) e += nl[j] * p[j];
        ent = d == 0 ? -e : ent + (-e);
      }
    }
    const PolicyTerm pt = ppo_policy_term(logp, logp_old[i], adv[i], hp);
    float dv;
    const float vterm = ppo_vf_term(value[i], ret[i], hp, &dv);
    acc[0] += (double)ent;
    acc[1] += (double)pt.term;
    acc[2] += (double)vterm;
    acc[3] += (double)pt.kl;
    if (HAS_GRAD) {
      grad_value[i] = hp.grad_scale * hp.vf_coeff * dv;
      for (int d = 0; d < a; ++d) {
        categorical_normalise_dyn<false>(logits + (i * a + d) * k, k, nl, p);
        float hd = 0.0f;
        if (with_entropy) {
          for (int j = 0; j < k; ++j) hd += nl[j] * p[j];
          hd = -hd;
        }
        const int64_t act = action[i * a + d];
        for (int j = 0; j < k; ++j) {
          const float dlogp = (j == act ? 1.0f : 0.0f) - p[j];
          float gj = -pt.dterm_dlogp * dlogp;
          if (with_entropy) gj -= hp.entropy_coeff * (-p[j] * (nl[j] + hd));
          grad_logits[(i * a + d) * k + j] = hp.grad_scale * gj;
        }
        if (k == 2) {  // exact antisymmetry of a two-way categorical's gradients (see the K = 2 kernel above)
          float *gp = grad_logits + (i * a + d) * 2;
          const float g = 0.5f * (gp[0] - gp[1]);
          gp[0] = g;
          gp[1] = -g;
        }
      }
    }
  }
  block_reduce<kLossCols, SumOp>(acc, smem);
  if (sums_out) {  // sole launch: publish + finalise
    publish_loss_row(acc, partials, blockIdx.x, gridDim.x, (double)m, sums_out, smem);
  } else if (threadIdx.x == 0) {  // tail helper of the vector kernel: row only
#pragma unroll
    for (int c = 0; c < kLossCols; ++c)
      partials[(int64_t)(partial_row0 + blockIdx.x) * kPartialWidth + c] = acc[c];
  }
}

// ---------------------------------------------------------------------------
// Normal / SquashedNormal, any A: one sample per thread.  Also mops up the
// (m % 4) tail of the A == 1 vector kernel via `first`.
// ---------------------------------------------------------------------------
template <bool HAS_GRAD>
__global__ __launch_bounds__(kBlock) void ppo_loss_normal_kernel(
    const float *__restrict__ mean, const float *__restrict__ log_std,
    const float *__restrict__ value, const float *__restrict__ action,
    const float *__restrict__ logp_old, const float *__restrict__ adv,
    const float *__restrict__ ret, int64_t first, int64_t m, int a, int squashed,
    rl8_ppo_hparams hp, float *__restrict__ grad_mean, float *__restrict__ grad_log_std,
    float *__restrict__ grad_value, double *__restrict__ partials, int partial_row0,
    double *__restrict__ sums_out) {
  __shared__ double smem[kLossCols * kWavesPerBlock];
  double acc[kLossCols] = {0.0, 0.0, 0.0, 0.0};
  const bool with_entropy = hp.entropy_coeff != 0.0f && !squashed;
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t i = first + (int64_t)blockIdx.x * kBlock + threadIdx.x; i < m; i += stride) {
    float logp = 0.0f, corr = 0.0f, ent = 0.0f;
    for (int d = 0; d < a; ++d) {
      const float mu = mean[i * a + d], sc = expf(log_std[i * a + d]);
      const float act = action[i * a + d];
      float l;
      if (squashed) {
        l = fminf(fmaxf(normal_log_prob(squashed_invert(act), mu, sc), -100.0f), 100.0f);
        const float t = logf((1.0f - act * act) + kF32Eps);
        corr = d == 0 ? t : corr + t;
      } else {
        l = normal_log_prob(act, mu, sc);
        const float e = kNormalEntropyConst + logf(sc);
        ent = d == 0 ? e : ent + e;
      }
      logp = d == 0 ? l : logp + l;
    }
    if (squashed) logp = logp - corr;
    const PolicyTerm pt = ppo_policy_term(logp, logp_old[i], adv[i], hp);
    float dv;
    const float vterm = ppo_vf_term(value[i], ret[i], hp, &dv);
    acc[0] += (double)(with_entropy ? ent : 0.0f);
    acc[1] += (double)pt.term;
    acc[2] += (double)vterm;
    acc[3] += (double)pt.kl;
    if (HAS_GRAD) {
      grad_value[i] = hp.grad_scale * hp.vf_coeff * dv;
      for (int d = 0; d < a; ++d) {
        const float mu = mean[i * a + d], sc = expf(log_std[i * a + d]);
        float x = action[i * a + d], pass = 1.0f;
        if (squashed) {
          x = squashed_invert(x);
          const float l = normal_log_prob(x, mu, sc);
          pass = (l >= -100.0f && l <= 100.0f) ? 1.0f : 0.0f;
        }
        const float z = (x - mu) / sc;
        float gm = -pt.dterm_dlogp * (pass * (z / sc));
        float gs = -pt.dterm_dlogp * (pass * (z * z - 1.0f));
        if (with_entropy) gs -= hp.entropy_coeff;
        grad_mean[i * a + d] = hp.grad_scale * gm;
        grad_log_std[i * a + d] = hp.grad_scale * gs;
      }
    }
  }
  block_reduce<kLossCols, SumOp>(acc, smem);
  if (sums_out) {  // sole launch: publish + finalise
    publish_loss_row(acc, partials, blockIdx.x, gridDim.x, (double)m, sums_out, smem);
  } else if (threadIdx.x == 0) {  // tail helper of the vector kernel: row only
#pragma unroll
    for (int c = 0; c < kLossCols; ++c)
      partials[(int64_t)(partial_row0 + blockIdx.x) * kPartialWidth + c] = acc[c];
  }
}

// Normal / SquashedNormal with one action dimension (the built-in continuous
// envs): four samples per lane, 16-byte accesses, and every transcendental of a
// sample evaluated once (exp(log_std), the two log1p of the squash inversion,
// log(scale), log(1 - a^2 + eps)) and shared by the loss and the gradients --
// the generic kernel above recomputes them for the backward half and was
// ALU-bound at 1.8 TB/s.  Same op order per value as the generic kernel.
template <bool HAS_GRAD, bool SQUASHED>
__global__ __launch_bounds__(kBlock) void ppo_loss_normal1_kernel(
    const float4 *__restrict__ mean, const float4 *__restrict__ log_std,
    const float4 *__restrict__ value, const float4 *__restrict__ action,
    const float4 *__restrict__ logp_old, const float4 *__restrict__ adv,
    const float4 *__restrict__ ret, int64_t groups, rl8_ppo_hparams hp,
    float4 *__restrict__ grad_mean, float4 *__restrict__ grad_log_std,
    float4 *__restrict__ grad_value, double *__restrict__ partials, int extra_rows, int64_t m,
    double *__restrict__ sums_out) {
  __shared__ double smem[kLossCols * kWavesPerBlock];
  float accf[kLossCols] = {0.0f, 0.0f, 0.0f, 0.0f};
  double acc[kLossCols] = {0.0, 0.0, 0.0, 0.0};
  const bool with_entropy = hp.entropy_coeff != 0.0f && !SQUASHED;
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  int since_flush = 0;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < groups; i += stride) {
    const float4 mu4 = mean[i], ls4 = log_std[i], v4 = value[i], a4 = action[i], lo4 = logp_old[i],
                 ad4 = adv[i], r4 = ret[i];
    const float mu[4] = {mu4.x, mu4.y, mu4.z, mu4.w}, ls[4] = {ls4.x, ls4.y, ls4.z, ls4.w};
    const float vv[4] = {v4.x, v4.y, v4.z, v4.w}, act[4] = {a4.x, a4.y, a4.z, a4.w};
    const float lo[4] = {lo4.x, lo4.y, lo4.z, lo4.w}, ad[4] = {ad4.x, ad4.y, ad4.z, ad4.w};
    const float rr[4] = {r4.x, r4.y, r4.z, r4.w};
    float gm[4], gs[4], gv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      // (round 4) the hardware exp2 / log2 / rcp units, as the categorical kernel's normalisation: this kernel is
      // tolerance-checked (1e-5 on the sums, 2e-5 on the gradients against the reference's autograd; the exact-order
      // functions stay in the generic kernel, i.e. for several action dims and the m % 4 tail) and ran at 0.35 of HBM on
      // two expf, two log1pf, two logf and three divisions per sample.  log(scale) is log_std itself; atanh(c) is
      // 0.5 log((1 + c) / (1 - c)), one logarithm instead of two log1p's.
      // 1 / scale from the accurate expf and an IEEE division: its relative error comes back multiplied by z^2 in logp
      // (a saturated action sits 8 to 20 scales from the mean: the two ulps of the hardware exp2 were 1.5e-5 of the ratio)
      const float inv_sc = 1.0f / expf(ls[u]);
      float x = act[u];
      if (SQUASHED) {
        const float c = fminf(fmaxf(act[u], -1.0f + kF32Eps), 1.0f - kF32Eps);
        // (the accurate logf: x enters logp through z^2 / 2 as well)
        x = 0.5f * logf((1.0f + c) * __builtin_amdgcn_rcpf(1.0f - c));
      }
      const float z = (x - mu[u]) * inv_sc;
      const float raw = (-0.5f * (z * z) - ls[u]) - kLogSqrt2Pi;
      float logp, ent = 0.0f, pass = 1.0f;
      if (SQUASHED) {
        const float l = fminf(fmaxf(raw, -100.0f), 100.0f);
        logp = l - __logf((1.0f - act[u] * act[u]) + kF32Eps);
        pass = (raw >= -100.0f && raw <= 100.0f) ? 1.0f : 0.0f;
      } else {
        logp = raw;
        ent = kNormalEntropyConst + ls[u];
      }
      const PolicyTerm pt = ppo_policy_term(logp, lo[u], ad[u], hp);
      float dv;
      const float vterm = ppo_vf_term(vv[u], rr[u], hp, &dv);
      accf[0] += with_entropy ? ent : 0.0f;
      accf[1] += pt.term;
      accf[2] += vterm;
      accf[3] += pt.kl;
      if (HAS_GRAD) {
        float m_ = -pt.dterm_dlogp * (pass * (z * inv_sc));
        float s_ = -pt.dterm_dlogp * (pass * (z * z - 1.0f));
        if (with_entropy) s_ -= hp.entropy_coeff;
        gm[u] = hp.grad_scale * m_;
        gs[u] = hp.grad_scale * s_;
        gv[u] = hp.grad_scale * hp.vf_coeff * dv;
      }
    }
    if (HAS_GRAD) {
      grad_mean[i] = make_float4(gm[0], gm[1], gm[2], gm[3]);
      grad_log_std[i] = make_float4(gs[0], gs[1], gs[2], gs[3]);
      grad_value[i] = make_float4(gv[0], gv[1], gv[2], gv[3]);
    }
    if (++since_flush == 16) {  // fp32 partials of <= 64 samples, then into fp64
#pragma unroll
      for (int c = 0; c < kLossCols; ++c) {
        acc[c] += (double)accf[c];
        accf[c] = 0.0f;
      }
      since_flush = 0;
    }
  }
#pragma unroll
  for (int c = 0; c < kLossCols; ++c) acc[c] += (double)accf[c];
  block_reduce<kLossCols, SumOp>(acc, smem);
  publish_loss_row(acc, partials, blockIdx.x, gridDim.x + extra_rows, (double)m, sums_out, smem);
}

// Workgroups of the 4-samples-per-lane kernel (RL8_LOSS_GRID_CAP: tuning knob).
static int vec_loss_grid(int64_t m) {
  const int64_t groups = m / 4;
  // One workgroup per CU, grid-striding: measured 5.1 TB/s vs 4.2 TB/s with 8
  // workgroups per CU at 8M samples (fewer concurrent streams per HBM channel).
  int grid = grid_for(groups, kBlock);
  if (grid > kCUs) grid = kCUs;
  static int cap = -1;
  if (cap < 0) {
    const char *v = getenv("RL8_LOSS_GRID_CAP");
    cap = v ? atoi(v) : 0;
  }
  if (cap > 0) {
    const int64_t gg = (groups + kBlock - 1) / kBlock;
    grid = (int)(gg < cap ? gg : cap);
    if (grid > RL8_MAX_PARTIALS) grid = RL8_MAX_PARTIALS;
    if (grid < 1) grid = 1;
  }
  return grid;
}

template <int K>
static int launch_categorical_vec(const float *logits, const float *value, const int64_t *action,
                                  const float *logp_old, const float *adv, const float *ret,
                                  int64_t m, const rl8_ppo_hparams &hp, float *grad_logits,
                                  float *grad_value, double *partials, int extra_rows,
                                  double *sums_out, hipStream_t s) {
  const int grid = vec_loss_grid(m);
  if (grad_logits)
    ppo_loss_categorical_kernel<K, 4, true><<<grid, kBlock, 0, s>>>(
        logits, value, action, logp_old, adv, ret, m, hp, grad_logits, grad_value, partials,
        extra_rows, sums_out);
  else
    ppo_loss_categorical_kernel<K, 4, false><<<grid, kBlock, 0, s>>>(
        logits, value, action, logp_old, adv, ret, m, hp, nullptr, nullptr, partials, extra_rows,
        sums_out);
  return launch_status();
}

}  // namespace rl8

using namespace rl8;

static int check_hp(const rl8_ppo_hparams *hp) {
  if (!hp) return RL8_ENULL;
  if (!(hp->clip_param > 0.0f && hp->clip_param < 1.0f)) return RL8_ECONFIG;
  if (!(hp->vf_clip_param > 0.0f)) return RL8_ECONFIG;
  return RL8_OK;
}

RL8_API int rl8_ppo_loss_categorical_fwd_bwd_f32(
    const float *logits, const float *value, const int64_t *action, const float *logp_old,
    const float *adv, const float *ret, int64_t m, int a, int k, const rl8_ppo_hparams *hp,
    float *grad_logits, float *grad_value, double *loss_sums_out, void *scratch, void *stream) {
  if (!logits || !value || !action || !logp_old || !adv || !ret || !loss_sums_out || !scratch)
    return RL8_ENULL;
  if ((grad_logits == nullptr) != (grad_value == nullptr)) return RL8_ENULL;
  if (m <= 0 || a <= 0 || k <= 1 || k > RL8_MAX_CLASSES) return RL8_ESIZE;
  int st = check_hp(hp);
  if (st != RL8_OK) return st;
  hipStream_t s = (hipStream_t)stream;
  double *partials = (double *)scratch;
  const bool vec_ok = a == 1 && (k == 2 || k == 3) && m >= 4 && aligned16(logits) &&
                      aligned16(value) && aligned16(action) && aligned16(logp_old) &&
                      aligned16(adv) && aligned16(ret) &&
                      (!grad_logits || (aligned16(grad_logits) && aligned16(grad_value)));
  if (vec_ok) {
    // Up to 3 tail samples go through the generic kernel first; it deposits its
    // row after the vector kernel's rows and the vector kernel finalises both.
    const int64_t first = (m / 4) * 4;
    const int extra = first < m ? 1 : 0;
    const int grid = vec_loss_grid(m);
    if (extra) {
      if (grad_logits)
        ppo_loss_categorical_generic_kernel<true><<<1, kBlock, 0, s>>>(
            logits, value, action, logp_old, adv, ret, first, m, a, k, *hp, grad_logits,
            grad_value, partials, grid, nullptr);
      else
        ppo_loss_categorical_generic_kernel<false><<<1, kBlock, 0, s>>>(
            logits, value, action, logp_old, adv, ret, first, m, a, k, *hp, nullptr, nullptr,
            partials, grid, nullptr);
      st = launch_status();
      if (st != RL8_OK) return st;
    }
    return (k == 2) ? launch_categorical_vec<2>(logits, value, action, logp_old, adv, ret, m, *hp,
                                                grad_logits, grad_value, partials, extra,
                                                loss_sums_out, s)
                    : launch_categorical_vec<3>(logits, value, action, logp_old, adv, ret, m, *hp,
                                                grad_logits, grad_value, partials, extra,
                                                loss_sums_out, s);
  }
  const int grid = grid_for(m, kBlock);
  if (grad_logits)
    ppo_loss_categorical_generic_kernel<true><<<grid, kBlock, 0, s>>>(
        logits, value, action, logp_old, adv, ret, 0, m, a, k, *hp, grad_logits, grad_value,
        partials, 0, loss_sums_out);
  else
    ppo_loss_categorical_generic_kernel<false><<<grid, kBlock, 0, s>>>(
        logits, value, action, logp_old, adv, ret, 0, m, a, k, *hp, nullptr, nullptr, partials, 0,
        loss_sums_out);
  return launch_status();
}

RL8_API int rl8_ppo_loss_normal_fwd_bwd_f32(
    const float *mean, const float *log_std, const float *value, const float *action,
    const float *logp_old, const float *adv, const float *ret, int64_t m, int a, int squashed,
    const rl8_ppo_hparams *hp, float *grad_mean, float *grad_log_std, float *grad_value,
    double *loss_sums_out, void *scratch, void *stream) {
  if (!mean || !log_std || !value || !action || !logp_old || !adv || !ret || !loss_sums_out ||
      !scratch)
    return RL8_ENULL;
  const int ng = (grad_mean != nullptr) + (grad_log_std != nullptr) + (grad_value != nullptr);
  if (ng != 0 && ng != 3) return RL8_ENULL;
  if (m <= 0 || a <= 0) return RL8_ESIZE;
  int st = check_hp(hp);
  if (st != RL8_OK) return st;
  if (squashed && hp->entropy_coeff != 0.0f) return RL8_ECONFIG;  // distributions.py:153-157
  hipStream_t s = (hipStream_t)stream;
  double *partials = (double *)scratch;
  const bool vec_ok = a == 1 && m >= 4 && aligned16(mean) && aligned16(log_std) && aligned16(value) &&
                      aligned16(action) && aligned16(logp_old) && aligned16(adv) && aligned16(ret) &&
                      (!ng || (aligned16(grad_mean) && aligned16(grad_log_std) && aligned16(grad_value)));
  if (vec_ok) {
    const int64_t groups = m / 4, first = groups * 4;
    const int extra = first < m ? 1 : 0;
    const int grid = grid_for(groups, kBlock, kMaxGrid - 1);  // + the tail's row
    if (extra) {  // up to 3 tail samples: a row behind the vector kernel's rows
      if (ng)
        ppo_loss_normal_kernel<true><<<1, kBlock, 0, s>>>(mean, log_std, value, action, logp_old, adv, ret, first,
                                                      m, a, squashed, *hp, grad_mean, grad_log_std,
                                                      grad_value, partials, grid, nullptr);
      else
        ppo_loss_normal_kernel<false><<<1, kBlock, 0, s>>>(mean, log_std, value, action, logp_old, adv, ret, first,
                                                       m, a, squashed, *hp, nullptr, nullptr, nullptr,
                                                       partials, grid, nullptr);
      st = launch_status();
      if (st != RL8_OK) return st;
    }
    auto f4 = [](const float *p) { return reinterpret_cast<const float4 *>(p); };
    auto g4 = [](float *p) { return reinterpret_cast<float4 *>(p); };
#define RL8_LAUNCH_NORMAL1(G, S)                                                                      \
  ppo_loss_normal1_kernel<G, S><<<grid, kBlock, 0, s>>>(f4(mean), f4(log_std), f4(value), f4(action),  \
                                                       f4(logp_old), f4(adv), f4(ret), groups, *hp,   \
                                                       g4(grad_mean), g4(grad_log_std),               \
                                                       g4(grad_value), partials, extra, m,            \
                                                       loss_sums_out)
    if (ng && squashed) RL8_LAUNCH_NORMAL1(true, true);
    else if (ng) RL8_LAUNCH_NORMAL1(true, false);
    else if (squashed) RL8_LAUNCH_NORMAL1(false, true);
    else RL8_LAUNCH_NORMAL1(false, false);
#undef RL8_LAUNCH_NORMAL1
    return launch_status();
  }
  const int grid = grid_for(m, kBlock);
  if (ng)
    ppo_loss_normal_kernel<true><<<grid, kBlock, 0, s>>>(
        mean, log_std, value, action, logp_old, adv, ret, 0, m, a, squashed, *hp, grad_mean,
        grad_log_std, grad_value, partials, 0, loss_sums_out);
  else
    ppo_loss_normal_kernel<false><<<grid, kBlock, 0, s>>>(
        mean, log_std, value, action, logp_old, adv, ret, 0, m, a, squashed, *hp, nullptr, nullptr,
        nullptr, partials, 0, loss_sums_out);
  return launch_status();
}
