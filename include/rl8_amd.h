/*
 * rl8_amd.h -- C ABI of the MI355X (gfx950) PPO hot path.
 *
 * One shared library, librl8_amd.so, built from rl8_amd/csrc with
 * `hipcc --offload-arch=gfx950`. Plain pointers and sizes only; every pointer is
 * a DEVICE pointer owned by the caller (PyTorch in the Python host, hipMalloc in
 * a C host) unless it says "host". Nothing here allocates, frees or
 * synchronises: each call enqueues kernels on `stream` (a hipStream_t passed as
 * void*; NULL = the default stream) and returns.
 *
 * Return value: 0 on success; RL8_E* (negative) when an argument check fails
 * (nothing was launched); a positive hipError_t if the launch itself failed.
 *
 * The reference (theOGognf/rl8) has no FFI on this path: its boundary is Python
 * call signatures. Each entry point below cites the reference function whose
 * arithmetic it replaces (paths relative to the reference checkout);
 * INTEGRATION.md shows the ctypes binding a reference maintainer would add.
 *
 * Layouts.  N = num_envs, H = horizon, M = samples in a (mini)batch,
 * A = action dims, K = classes per action dim.
 *   "env-major"  buffer leaf: [N][H+1] contiguous  (the reference's [N, H+1, 1])
 *   "time-major" buffer leaf: [H+1][N] contiguous  (rl8_amd's own rollout
 *                buffer; exposed to Python as a transposed [N, H+1, 1] view)
 *
 * Which entry runs where (round 6; every entry is exported, declared here and called by rl8_amd/hip.py --
 * tests/test_host_logic.py checks the three lists against each other; none is dead):
 *   PRODUCT, default path (AlgorithmConfig / RecurrentAlgorithmConfig defaults, built-in envs, default models):
 *     rollout    rl8_rollout_step_{dummy,cartpole,mountain_car,pendulum}_f32, rl8_rollout_step_dummy_heads_f32,
 *                rl8_rollout_scatter_f32 (user envs), rl8_rollout_stats_f32, rl8_*_reset_f32, rl8_*_step_f32 (Env.step()
 *                called by the user), rl8_categorical_sample_logp_f32, rl8_normal_sample_logp_f32 (Distribution.sample())
 *     update     rl8_gae_scan_f32, rl8_advantage_normalise_f32, rl8_ppo_loss_{categorical,normal}_fwd_bwd_f32,
 *                rl8_pack_samples, rl8_gather_packed, rl8_gather_minibatch
 *     towers     rl8_mlp_tower_forward_f16_f32, rl8_mlp_tower_backward_gate_f16_f32, rl8_mlp_tower_backward_f16_f32,
 *                rl8_mlp_wgrad_gate_bits_f32, rl8_mlp_wgrad_fused_split_f32, rl8_mlp_wgrad_fused_pair_f32,
 *                rl8_mlp_pack_w2_f16, rl8_mlp_pack_w2_f16_gate, rl8_mlp_dout_pair_check, the *_supports / *_bytes /
 *                *_floats / *_max_rows queries
 *     recurrent  rl8_lstm_pack_split, rl8_lstm_split_state[_bound], rl8_lstm_step_split_f32,
 *                rl8_lstm_rows_backward_pack, rl8_lstm_rows_backward[_heads]_f32, rl8_lstm_wgrad_f16_f32,
 *                rl8_linear_heads_{forward,forward_pair,backward}_f32
 *   PRODUCT, fallback (shapes outside the plane kernels' envelope -- d_in > 16, n_out > 8, hidden != 256 go to eager --
 *   or a switch: RL8_AMD_TOWER_GEMM=f32, RL8_AMD_LSTM_GEMM=f32, RL8_WGRAD_PLANES=bf16, RL8_AMD_LSTM_WGRAD_PLANES=bf16,
 *   RL8_AMD_LSTM_BACKWARD_ROWS=0), and THE YARDSTICK the parity tests hold the plane kernels against beside fp64:
 *     rl8_mlp_tower_forward_f32, rl8_mlp_tower_backward_f32, rl8_mlp_wgrad_f32, rl8_mlp_wgrad_split_f32,
 *     rl8_mlp_pack_w2_f32, rl8_lstm_pack_f32, rl8_lstm_forward_f32, rl8_lstm_backward_f32,
 *     rl8_mlp_wgrad_strided_f32, rl8_mlp_wgrad_split_strided_f32, rl8_mlp_wgrad_f16_strided_f32 (one gate per launch)
 *   OPT-IN prototype (RL8_AMD_TOWERS=piecewise; never the default, never the headline): rl8_pw_*
 */
#ifndef RL8_AMD_H
#define RL8_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RL8_OK 0
#define RL8_ENULL (-1)   /* required pointer is NULL */
#define RL8_ESIZE (-2)   /* size / shape argument out of range */
#define RL8_EALIGN (-3)  /* pointer not aligned as required */
#define RL8_ECONFIG (-4) /* unsupported combination of options */

#define RL8_MAX_CLASSES 64 /* K upper bound for the categorical kernels */
#define RL8_MAX_PARTIALS 2048 /* upper bound on per-launch partial rows */

/* ABI / build identification: returns RL8_ABI_VERSION of the library's build; `arch` (host pointer, may be NULL)
 * receives "gfx950".  A binding compares it with the header it was written against (rl8_amd/hip.py refuses a
 * mismatch).  History: 100 rounds 1-2; 103 round 3 (bf16-plane forward / data-gradient entries removed, reduction
 * scratch doubled with two-level tickets -- bump owed since then); 104 round 4 (weight-gradient
 * workspace: guard words and lifetime counters behind the slabs; caller-owned outputs of the forward unchanged);
 * 105 round 4 (rl8_gather_minibatch takes index = NULL: all samples in order); 106 round 6 (no signature changed:
 * rl8_lstm_rows_backward_pack writes the fp16 planes of W_hh^T behind the bf16 ones -- rl8_lstm_rows_backward_pack_bytes
 * grew by 1 MiB + 16, and rl8_lstm_rows_backward_heads_f32 reads them; rl8_mlp_backward_f16_supports and
 * rl8_lstm_split_supports report wider envelopes; the weight-gradient workspace's tail words moved). */
#define RL8_ABI_VERSION 106
int rl8_abi_version(char *arch, int arch_len);

/* Scratch the reductions need (bytes); the caller allocates it once per stream
 * (device memory, 16-byte aligned), ZEROES it once, and passes it to the calls
 * that take `scratch`.  It holds per-block partial rows and the arrival ticket
 * of the single-launch reductions (re-zeroed by each call); two calls sharing
 * one scratch must be ordered on one stream. */
int64_t rl8_scratch_bytes(void);

/* ---------------------------------------------------------------------- *
 * a-1  DummyEnv.step                  src/rl8/env.py:224-230, 253-259
 * state [N] f32 updated in place; action [N] int64 (discrete: state += 2a-1)
 * or f32 (continuous: state += a); reward_out [N] = -|state|.
 * ---------------------------------------------------------------------- */
int rl8_dummy_env_step_f32(float *state, const void *action, int is_discrete,
                           float *reward_out, int64_t n, void *stream);

/* DummyEnv.reset                      src/rl8/env.py:197-203
 * state ~ U(-bounds, bounds) from the build's Philox stream (rl8_philox.h);
 * env_offset = global index of this shard's first env. */
int rl8_dummy_env_reset_f32(float *state, int64_t n, float bounds, uint64_t seed,
                            uint64_t reset_count, int64_t env_offset, void *stream);

/* ---------------------------------------------------------------------- *
 * a-2  CartPole step                  examples/cartpole/env.py:12-64
 * state SoA [4][N] (x, x_dot, theta, theta_dot) in place; action [N] int64 in
 * {0,1,2}; obs_out [N][5] rows (x, x_dot, cos, sin, theta_dot) written with
 * row stride obs_stride floats (5 for a dense [N,5]); reward_out [N].
 * ---------------------------------------------------------------------- */
typedef struct {
  float force_mag, gravity, length, pole_mass, pole_mass_length, total_mass, tau;
  int32_t semi_implicit; /* 0: "euler" (:43-47), 1: semi-implicit (:48-52) */
} rl8_cartpole_cfg;

int rl8_cartpole_step_f32(float *state, const int64_t *action, const rl8_cartpole_cfg *cfg /*host*/,
                          float *obs_out, int64_t obs_stride, float *reward_out, int64_t n,
                          void *stream);

/* CartPole.reset                      examples/cartpole/env.py:128-136
 * state ~ N(0, std) [4][N]; obs_out as above (may be NULL). */
int rl8_cartpole_reset_f32(float *state, int64_t n, float std, uint64_t seed,
                           uint64_t reset_count, int64_t env_offset, float *obs_out,
                           int64_t obs_stride, void *stream);

/* ---------------------------------------------------------------------- *
 * a-6  Distribution.sample / logp     src/rl8/distributions.py:113-170
 * (-> torch.distributions.Categorical.sample == argmax(p / q), q ~ Exp(1))
 * logits [M][A][K] f32; noise: injected q [M][A][K] (may be NULL -> Philox
 * (seed, row_offset + row, step)); action_out [M][A] int64; logp_out [M] f32.
 * deterministic != 0 -> mode.
 * ---------------------------------------------------------------------- */
int rl8_categorical_sample_logp_f32(const float *logits, const float *noise, int64_t *action_out,
                                    float *logp_out, int64_t m, int a, int k, uint64_t seed,
                                    uint64_t step, int64_t row_offset, int deterministic,
                                    void *stream);

/* Normal / SquashedNormal: mean, log_std [M][A]; noise: injected eps [M][A]
 * (may be NULL -> Philox); action_out [M][A] f32; logp_out [M]. */
int rl8_normal_sample_logp_f32(const float *mean, const float *log_std, const float *noise,
                               float *action_out, float *logp_out, int64_t m, int a,
                               int squashed, uint64_t seed, uint64_t step, int64_t row_offset,
                               int deterministic, void *stream);

/* ---------------------------------------------------------------------- *
 * a-3  Rollout bookkeeping            src/rl8/algorithms/_feedforward.py:378-393
 * One launch per timestep t for a generic Env: rdr[t+1] = gamma*rdr[t] + r and
 * the five column writes.  All destinations are columns of TIME-MAJOR leaves,
 * i.e. contiguous [N][d] slabs; sources are the policy / env outputs [N][d].
 * rdr_t / rdr_t1 may be NULL (normalize_rewards = False).
 * ---------------------------------------------------------------------- */
int rl8_rollout_scatter_f32(const void *action, int64_t action_row_bytes, const float *logp,
                            const float *value, const float *reward, const float *obs,
                            int64_t obs_dim, void *action_col, float *logp_col, float *value_col,
                            float *reward_col, float *obs_col_next, const float *rdr_t,
                            float *rdr_t1, float gamma, int64_t n, void *stream);

/* Fused per-timestep kernel for the built-in dummy envs: sampler (a-6) +
 * env.step (a-1) + bookkeeping (a-3) in ONE launch.
 *   discrete:   features = logits [N][1][2];  noise = q [N][1][2] or NULL
 *   continuous: features = mean [N][1], features2 = log_std [N][1]; noise = eps
 *               [N][1] or NULL; squashed selects SquashedNormal.
 * state [N] in place (the env's state; also written to obs_col_next). */
int rl8_rollout_step_dummy_f32(int is_discrete, int squashed, const float *features,
                               const float *features2, const float *value, const float *noise,
                               float *state, void *action_col, float *logp_col, float *value_col,
                               float *reward_col, float *obs_col_next, const float *rdr_t,
                               float *rdr_t1, float gamma, int64_t n, uint64_t seed, uint64_t step,
                               int64_t env_offset, int deterministic, void *stream);

/* The discrete form for the default recurrent model (models/_recurrent.py:287-321: a two-way logits head and a value
 * head on the LSTM's output), heads included: h [N][256] = h_t, w_pol [2][256], b_pol [2], w_vf [1][256], b_vf [1];
 * logits and value are formed in the kernel exactly as rl8_linear_heads_forward_f32 forms them, the rest as
 * rl8_rollout_step_dummy_f32(is_discrete = 1): one launch per rollout timestep behind the LSTM step instead of three. */
int rl8_rollout_step_dummy_heads_f32(const float *h, const float *w_pol, const float *b_pol, const float *w_vf,
                                     const float *b_vf, const float *noise, float *state, int64_t *action_col,
                                     float *logp_col, float *value_col, float *reward_col, float *obs_col_next,
                                     const float *rdr_t, float *rdr_t1, float gamma, int64_t n, uint64_t seed,
                                     uint64_t step, int64_t env_offset, int deterministic, void *stream);

/* Fused per-timestep kernel for CartPole (K = 3): sampler + physics +
 * bookkeeping.  obs_col_next is a [N][5] slab. */
int rl8_rollout_step_cartpole_f32(const float *logits, const float *value, const float *noise,
                                  float *state, const rl8_cartpole_cfg *cfg /*host*/,
                                  int64_t *action_col, float *logp_col, float *value_col,
                                  float *reward_col, float *obs_col_next, const float *rdr_t,
                                  float *rdr_t1, float gamma, int64_t n, uint64_t seed,
                                  uint64_t step, int64_t env_offset, int deterministic,
                                  void *stream);

/* ---------------------------------------------------------------------- *
 * N2  The other example environments, same SoA-state template
 *
 * MountainCar                         examples/mountain_car/env.py:12-38, :96-104
 * state SoA [2][N] (position, velocity) in place; action [N] int64 in {0,1,2};
 * obs_out [N][2] rows (position, velocity) with row stride obs_stride floats;
 * reward_out [N].  reset: position ~ N(-0.5, 0.05), velocity ~ N(0, 0.05) from the
 * build's Philox stream; obs_out may be NULL.
 * ---------------------------------------------------------------------- */
typedef struct {
  float force_mag, goal_position, goal_velocity, gravity, max_position, max_speed, min_position;
} rl8_mountain_car_cfg;

int rl8_mountain_car_step_f32(float *state, const int64_t *action,
                              const rl8_mountain_car_cfg *cfg /*host*/, float *obs_out,
                              int64_t obs_stride, float *reward_out, int64_t n, void *stream);
int rl8_mountain_car_reset_f32(float *state, int64_t n, uint64_t seed, uint64_t reset_count,
                               int64_t env_offset, float *obs_out, int64_t obs_stride,
                               void *stream);
/* Fused per-timestep kernel (K = 3 sampler + physics + bookkeeping), arguments as
 * rl8_rollout_step_cartpole_f32; obs_col_next is a [N][2] slab. */
int rl8_rollout_step_mountain_car_f32(const float *logits, const float *value, const float *noise,
                                      float *state, const rl8_mountain_car_cfg *cfg /*host*/,
                                      int64_t *action_col, float *logp_col, float *value_col,
                                      float *reward_col, float *obs_col_next, const float *rdr_t,
                                      float *rdr_t1, float gamma, int64_t n, uint64_t seed,
                                      uint64_t step, int64_t env_offset, int deterministic,
                                      void *stream);

/* Pendulum                            examples/pendulum/env.py:12-39, :101-113
 * state SoA [2][N] (th, thdot) in place; action [N] f32 (clipped to +-max_torque
 * inside); obs_out [N][3] rows (cos th', sin th', thdot'); reward_out [N] = minus
 * the cost of the state and torque BEFORE the step.  gravity_coeff = 3g/(2l) and
 * torque_coeff = 3/(m l^2), formed in double by the caller as the reference does
 * (:32).  reset: th ~ U(-pi, pi), thdot ~ U(-1, 1). */
typedef struct {
  float dt, gravity_coeff, torque_coeff, max_speed, max_torque;
} rl8_pendulum_cfg;

int rl8_pendulum_step_f32(float *state, const float *action, const rl8_pendulum_cfg *cfg /*host*/,
                          float *obs_out, int64_t obs_stride, float *reward_out, int64_t n,
                          void *stream);
int rl8_pendulum_reset_f32(float *state, int64_t n, uint64_t seed, uint64_t reset_count,
                           int64_t env_offset, float *obs_out, int64_t obs_stride, void *stream);
/* Fused per-timestep kernel: Normal / SquashedNormal sampler (mean, log_std [N],
 * noise: injected standard normals or NULL) + physics + bookkeeping;
 * action_col [N] f32, obs_col_next [N][3]. */
int rl8_rollout_step_pendulum_f32(int squashed, const float *mean, const float *log_std,
                                  const float *value, const float *noise, float *state,
                                  const rl8_pendulum_cfg *cfg /*host*/, float *action_col,
                                  float *logp_col, float *value_col, float *reward_col,
                                  float *obs_col_next, const float *rdr_t, float *rdr_t1,
                                  float gamma, int64_t n, uint64_t seed, uint64_t step,
                                  int64_t env_offset, int deterministic, void *stream);

/* ---------------------------------------------------------------------- *
 * a-4  Collect statistics             src/rl8/algorithms/_feedforward.py:411-436
 * rewards / rdr leaves with element strides (env_stride, time_stride) so both
 * layouts are accepted; rdr may be NULL.  Writes 12 doubles to stats_out:
 *   [0] n_envs  [1] sum(ret) [2] sum(ret^2) [3] min(ret) [4] max(ret)
 *   [5] n*h     [6] sum(r)   [7] sum(r^2)   [8] min(r)   [9] max(r)
 *   [10] sum(rdr[:,1:]) [11] sum(rdr[:,1:]^2)
 * (raw moments, so that shards can be combined with one all-reduce each for
 * SUM / MIN / MAX before the host forms mean and unbiased std).
 * ---------------------------------------------------------------------- */
int rl8_rollout_stats_f32(const float *rewards, const float *rdr, int64_t n, int64_t h,
                          int64_t env_stride, int64_t time_stride, double *stats_out,
                          void *scratch, void *stream);

/* ---------------------------------------------------------------------- *
 * a-5  generalized_advantage_estimate src/rl8/nn/functional.py:50-123
 * Scan: rewards are divided by `reward_denominator` (= f32(reward_scale+1e-8),
 * :106) on load -- and written back when write_scaled_rewards != 0, as the
 * reference does -- then for t = H-1..0
 *   delta = r[t] + (gamma*v[t+1] - v[t]);  adv[t] = delta + gamma_lambda*adv[t+1]
 * adv[H] = 0, ret[t] = adv[t] + v[t] for all H+1 columns (:116-117).
 * moments_out (3 doubles: count, sum, sum of squares of adv[:, :H]) feeds the
 * normalisation; all-reduce it across env shards before normalising.
 * layout: 0 = env-major (LDS-staged tiles), 1 = time-major (register scan).
 * rewards and adv_out / values and ret_out may not alias each other.
 * ---------------------------------------------------------------------- */
int rl8_gae_scan_f32(float *rewards, const float *values, float *adv_out, float *ret_out,
                     int64_t n, int64_t h, int layout, float gamma, float gamma_lambda,
                     float reward_denominator, int write_scaled_rewards, double *moments_out,
                     void *scratch, void *stream);

/* adv[:, :H] = (adv - mean) / (std + 1e-8), mean / unbiased std formed from
 * `moments` (device, 3 doubles) (:118-122). */
int rl8_advantage_normalise_f32(float *adv, int64_t n, int64_t h, int layout,
                                const double *moments, void *stream);

/* ---------------------------------------------------------------------- *
 * a-7  ppo_losses + approximate KL, forward AND backward
 *      src/rl8/nn/functional.py:259-363, algorithms/_feedforward.py:545-559
 * Per sample i: features, value[i], action[i], logp_old[i], adv[i], ret[i].
 * Writes grad_* = d(total)/d(input) * grad_scale (grad_scale =
 * 1 / (M_global * grad_accumulation_steps)), and loss_sums_out (device, 5
 * doubles): sum over samples of entropy, policy, vf terms, the sample count,
 * and the KL terms -- raw sums so that env shards combine with one all-reduce;
 * the host divides.  grad pointers may be NULL (forward only).
 * ---------------------------------------------------------------------- */
typedef struct {
  float clip_param;
  float dual_clip_param; /* <= 0: disabled (reference: None) */
  float entropy_coeff;
  float vf_clip_param;
  float vf_coeff;
  float grad_scale;
} rl8_ppo_hparams;

int rl8_ppo_loss_categorical_fwd_bwd_f32(const float *logits, const float *value,
                                         const int64_t *action, const float *logp_old,
                                         const float *adv, const float *ret, int64_t m, int a,
                                         int k, const rl8_ppo_hparams *hp /*host*/,
                                         float *grad_logits, float *grad_value,
                                         double *loss_sums_out, void *scratch, void *stream);

int rl8_ppo_loss_normal_fwd_bwd_f32(const float *mean, const float *log_std, const float *value,
                                    const float *action, const float *logp_old, const float *adv,
                                    const float *ret, int64_t m, int a, int squashed,
                                    const rl8_ppo_hparams *hp /*host*/, float *grad_mean,
                                    float *grad_log_std, float *grad_value,
                                    double *loss_sums_out, void *scratch, void *stream);

/* ---------------------------------------------------------------------- *
 * a-8  Batcher gather                 src/rl8/_utils.py:211-225
 * index [M] int64 holds REFERENCE sample ids s = env*H + t (the flattening of
 * _feedforward.py:481).  Each of the `n_fields` sources is a buffer leaf with
 * element strides; the matching destination is a dense [M][row_elems] array.
 * Elements are 4 or 8 bytes wide (f32 / int64).
 * index = NULL (ABI 105): every sample of the buffer in order, s = 0 .. m - 1 with m = N h -- the sequence-major
 * copy of a time-major buffer that the recurrent algorithm trains on -- as a transposition through LDS tiles
 * (each byte read and written once, where the indexed kernel moves 6x that for 4-byte leaves); rows of at most
 * 128 bytes summed over the fields, else RL8_ESIZE.
 * ---------------------------------------------------------------------- */
typedef struct {
  const void *src;
  void *dst;
  int64_t env_stride;  /* in elements */
  int64_t time_stride; /* in elements */
  int32_t row_elems;   /* trailing elements per (env, t) cell */
  int32_t elem_bytes;  /* 4 or 8 */
} rl8_gather_field;

#define RL8_MAX_GATHER_FIELDS 8

int rl8_gather_minibatch(const int64_t *index, int64_t m, int64_t h,
                         const rl8_gather_field *fields /*host*/, int n_fields, void *stream);

/* The same gather for a buffer that is shuffled many times per step()
 * (num_sgd_iters x num_minibatches): rl8_pack_samples lays the fields of every
 * sample side by side once -- packed[s][...] for the reference's sample id
 * s = env*H + t, t < h, row_words 4-byte words per sample (a multiple of 4, at most
 * 32, at least the sum of the fields' row words, fields in the order given; `dst` of the
 * fields is ignored) -- and rl8_gather_packed then reads one row per sample
 * (`src`, `env_stride`, `time_stride` ignored): one random sector per sample
 * instead of one per field. */
int rl8_pack_samples(const rl8_gather_field *fields /*host*/, int n_fields, int64_t n, int64_t h,
                     void *packed, int row_words, void *stream);
int rl8_gather_packed(const int64_t *index, int64_t m, const void *packed, int row_words,
                      const rl8_gather_field *fields /*host*/, int n_fields, void *stream);

/* ---------------------------------------------------------------------- *
 * N1  Default policy / value tower, fused
 *      src/rl8/models/_feedforward.py:336-375 (DefaultDiscreteModel),
 *      :263-310 (DefaultContinuousModel), src/rl8/nn/modules/mlp.py:12-52
 * One tower = Linear(d_in, 256) ReLU Linear(256, 256) ReLU Linear(256, n_out),
 * fp32 throughout (layer 2 on v_mfma_f32_32x32x2_f32), activations kept on
 * chip.  d_in <= 16, n_out <= 8.  Weights in torch.nn.Linear layout
 * ([out][in] row-major); the 256x256 matrix is passed in MFMA fragment order,
 * produced by rl8_mlp_pack_w2_f32 (transposed = 0 for the forward pass and the
 * weight-gradient pass, 1 for the data-gradient pass of the backward).
 * ---------------------------------------------------------------------- */
int rl8_mlp_pack_w2_f32(const float *w2 /*[256][256]*/, float *w2_packed /*[65536]*/,
                        int transposed, void *stream);

/* out [M][n_out] = tower(x [M][d_in]).  For training, save_h1 / save_h2
 * ([M][256] each, 16-byte aligned) receive the post-ReLU activations; both NULL
 * for inference. */
int rl8_mlp_tower_forward_f32(const float *x, int64_t m, int d_in, const float *w1,
                              const float *b1, const float *w2_packed, const float *b2,
                              const float *w3, const float *b3, int n_out, float *out,
                              float *save_h1, float *save_h2, void *stream);

/* The same forward with the 256x256 product as THREE fp16 MFMAs per 16 k: each
 * activation row of h1 is scaled by a power of two chosen from a bound on its
 * magnitude (max|b1| + sum_i |x_i| max|w1[:, i]|, placed at 2^14) and W2 by one
 * power of two for the matrix, so both operands sit in the upper fp16 exponent
 * range; each is then split into hi = fp16(v), lo = fp16(v - hi) (22 significand
 * bits) and lo.hi + hi.lo + hi.hi is accumulated in fp32; the epilogue multiplies
 * by the inverse powers of two (exact).  Inputs, outputs and saved activations are fp32
 * exactly as for rl8_mlp_tower_forward_f32 (h1 bit-identical at d_in <= 3 and equal to fp32 rounding above -- there layer 1
 * runs on the matrix pipe too; out / h2 equal to fp32 rounding).
 * save_h1 may be NULL while save_h2 is given: the plane backward kernels recompute h1 from the
 * observations and never read it.  save_gate2 (optional) [M][8] words receives the ReLU gate of
 * layer 2, bit j of row s = (h2[s][j] > 0): the data-gradient kernels need only that bit of h2
 * (32 B per row instead of 1 KiB), and it may be given WITHOUT save_h2 (rank-one heads: see
 * rl8_mlp_wgrad_gate_bits_f32).  Widths (round 5): any d_in <= 8 with n_out <= 8, and d_in 9..16 with n_out <= 4
 * (rl8_mlp_forward_f16_supports) -- the
 * kernels are compiled for width CLASSES d_in {1, 2, 3, 8, 16} x n_out {1, 2, 4, 8} and a class serves every run-time
 * width it holds (weights past d_in zero, observations past d_in not loaded, output rows past n_out zero and not stored;
 * class 8, d_in = 4..8, forms z1 = W1 x as ONE 16x16x32 MFMA per sixteen units and rows that holds the four fp16 plane
 * products of eight inputs; class 16 chains two of them) -- else RL8_ESIZE: wider towers run rl8_mlp_tower_forward_f32.  The plane BACKWARD entries
 * below serve d_in 1..5 x n_out 1..4 (rl8_mlp_backward_f16_supports; the weight-gradient kernels are compiled per width,
 * the data-gradient kernels per class -- d_in 4, 5 in class 8, which recomputes the forward's z1 product bit for bit and
 * forms dW1 / db1 as MFMAs over the wave's rows); a tower inside the forward's envelope but outside theirs is trained
 * through the fp32-MFMA data gradient + bf16-plane weight gradient (save_h1 given).  (The bf16-plane forward / data-gradient
 * entries of rounds 1-2, rl8_mlp_tower_{forward,backward}_split_f32, were removed in round 3.)
 * w2_f16 (rl8_mlp_f16_packed_bytes() bytes:
 * two fp16 planes in fragment order + {scale, 1/scale}) comes from
 * rl8_mlp_pack_w2_f16.  The fragment order depends on `transposed`: 0 (this forward; since round 3 a kernel of
 * v_mfma_f32_16x16x32_f16, a wave per 32 rows) packs 16-byte unit ((hs*8 + ct)*2 + plane)*64 + lane with
 * B(col = 16 (8 (hs & 1) + ct) + (lane & 15), k = 32 (hs >> 1) + 8 (lane >> 4) + e), e = 0..7; 1 (the data-gradient
 * kernels, 32x32x16) keeps unit ((s*8 + ct)*2 + plane)*64 + lane with B(col = 32 ct + (lane & 31),
 * k = 16 s + 8 (lane >> 5) + e) of the transposed matrix.  A pack made for one is not valid for the other.
 * Replaces the same reference call sites
 * (rl8/models/_feedforward.py:115-132 forward of the 256-256 towers). */
int64_t rl8_mlp_f16_packed_bytes(void);
int rl8_mlp_forward_f16_supports(int d_in, int n_out);
int rl8_mlp_pack_w2_f16(const float *w2 /*[256][256]*/, int transposed, void *w2_f16, void *stream);
int rl8_mlp_tower_forward_f16_f32(const float *x, int64_t m, int d_in, const float *w1,
                                  const float *b1, const void *w2_f16, const float *b2,
                                  const float *w3, const float *b3, int n_out, float *out,
                                  float *save_h1, float *save_h2, uint32_t *save_gate2, void *stream);

/* A head of TWO outputs whose gradients are exact negatives of each other (a two-way categorical:
 * rl8_ppo_loss_categorical_fwd_bwd_f32 emits them so) has dZ2 = gate * dOut[.][0] * (W3[0] - W3[1]):
 * the weight gradient then needs one binary operand and three plane products per 16 samples
 * instead of six (as for single-output towers, which rl8_mlp_wgrad_fused_split_f32 handles by
 * itself).  rl8_mlp_dout_pair_check leaves 0 in *flag_out (device int) iff dout [m][2] has
 * dout[s][1] == -dout[s][0] bit for bit in every row; only then may rl8_mlp_wgrad_fused_pair_f32
 * replace rl8_mlp_wgrad_fused_split_f32(..., n_out = 2, ...): same arguments, same outputs. */
int rl8_mlp_dout_pair_check(const float *dout /*[m][2]*/, int64_t m, int *flag_out /*device*/, void *stream);
int rl8_mlp_wgrad_fused_pair_f32(const float *h2, const float *dout, const float *x, const float *w1,
                                 const float *b1, const float *w3, int64_t m, int d_in,
                                 float *workspace, float *dw2_out, float *partials, void *stream);

/* Rank-one heads without h2.  rl8_mlp_tower_forward_f16_f32 with save_h2 = NULL and save_gate2 given stores the
 * gate bits alone (32 bytes per row instead of 1 KiB, none of the h2 store path); the data gradient's gate mode
 * never needed h2; and the weight gradient's only other use of it, dW3 = dOut^T h2, follows from the sums M the
 * gate-plane kernel forms anyway (dW2[j][i] = W3[j] M[j][i]):  with h2 = gate * (h1 W2^T + b2),
 *   dW3[j] = sum_i W2[j][i] M[j][i] + b2[j] * sum_s gate[s][j] dOut[s].
 * rl8_mlp_wgrad_gate_bits_f32: outputs of rl8_mlp_wgrad_fused_split_f32 (n_out = 1) / _pair_f32 (n_out = 2, after
 * a clean rl8_mlp_dout_pair_check) from gate2 [m][8] instead of h2; w2 [256][256] and b2 [256] are the layer's
 * own parameters.  In fp32 the result differs from the direct sum by the rounding the forward's own 256-term dot
 * products put into h2 -- what another fp32 evaluation of h2 (the reference's) differs from this one by.
 * Since round 3 the second operand dOut[s] * h1[s][i] is carried as two fp16 planes times a power of two per column i
 * (two plane products per 16 rows; the column bounds come from one pass over dout and x in front of the first
 * segment and live in the 256 bytes behind the slabs of `workspace`); RL8_WGRAD_GATE_PLANES=bf16 in the environment
 * selects the three exact bf16 planes of round 2.  dW2 stays within the fp32 kernels' error against the tensor's
 * largest entry; an entry made only of rows 2^17 below its column's bound keeps ~1e-5 of its own size. */
int rl8_mlp_wgrad_gate_bits_f32(const uint32_t *gate2, const float *dout, const float *x, const float *w1,
                                const float *b1, const float *w2, const float *b2, const float *w3,
                                int64_t m, int d_in, int n_out, float *workspace, float *dw2_out,
                                float *partials, void *stream);

/* The data-gradient half of the fused backward pass on the same fp16 planes (three MFMAs
 * per 16 k): no dZ2 store; instead of the saved h1 the kernel takes layer 1 itself (w1, b1) and recomputes
 * the ReLU gate h1 > 0 <=> b1 + x . w1 > 0 with the forward pass's own fma chain; gate2 -- the
 * forward's save_gate2 -- is required; w2t_f16 from rl8_mlp_pack_w2_f16(..., transposed = 1).
 * dZ2 rows are scaled by a power of two from the row bound sum_q |dOut_q| max|W3_q|, W2^T
 * by the pack's, both undone on the accumulators.  Fills [dW1 | db1] of `*partial_rows_out`
 * partial rows (layout of rl8_mlp_tower_backward_f32); the caller follows with
 * rl8_mlp_wgrad_fused_split_f32 for dW2 and the head segments.
 * Replaces the autograd backward of the 256-256 towers
 * (rl8/algorithms/_feedforward.py:374-386 loss.backward()). */
int rl8_mlp_backward_f16_supports(int d_in, int n_out);
int rl8_mlp_tower_backward_f16_f32(const float *x, const float *w1, const float *b1, const float *dout,
                                   int64_t m, int d_in, const void *w2t_f16, const float *w3, int n_out,
                                   float *partials, int *partial_rows_out /*host*/, const uint32_t *gate2,
                                   void *stream);

/* Gate mode of the fp16 data-gradient kernel, for heads whose dZ2 is gate * d[s] * w3e[k] -- n_out = 1
 * (w3e = W3[0]), or n_out = 2 with dout[s][1] == -dout[s][0] in every row (rl8_mlp_dout_pair_check;
 * w3e = W3[0] - W3[1], d = dout[.][0]):  dH1[s][i] = d[s] * sum_k gate[s][k] * (w3e[k] W2[k][i]).
 * The gate is the A operand (one fp16 plane of zeros and ones straight from the gate bits), B the two
 * planes of w3e[k] W2[k][i] from rl8_mlp_pack_w2_f16_gate (rl8_mlp_f16_packed_bytes() bytes; re-made
 * when W2 or W3 change): two plane products per 16 k instead of three.  Other arguments, partial rows
 * and the follow-up weight-gradient call as rl8_mlp_tower_backward_f16_f32. */
int rl8_mlp_pack_w2_f16_gate(const float *w2 /*[256][256]*/, const float *w3 /*[n_out][256]*/, int n_out,
                             void *w2t_gate, void *stream);
int rl8_mlp_tower_backward_gate_f16_f32(const float *x, const float *w1, const float *b1, const float *dout,
                                        int64_t m, int d_in, const void *w2t_gate, int n_out,
                                        float *partials, int *partial_rows_out /*host*/,
                                        const uint32_t *gate2, void *stream);

/* Backward of one tower ("dgrad" half): given dOut [M][n_out] and the saved
 * activations h1 / h2, writes dZ2 [M][256] (input of rl8_mlp_wgrad_f32, which
 * forms dW2 = dZ2^T h1) and `*partial_rows_out` rows (<= rl8_mlp_backward_max_rows()) of
 * rl8_mlp_backward_partial_floats(d_in, n_out) floats each into `partials`:
 *   [dW1 (256*d_in) | db1 (256) | db2 (256) | dW3 (n_out*256) | db3 (n_out)]
 * whose column sums are the parameter gradients.  w2t_packed is
 * rl8_mlp_pack_w2_f32(..., transposed = 1).  partial_rows_out is a host pointer. */
int64_t rl8_mlp_backward_partial_floats(int d_in, int n_out);
int rl8_mlp_backward_max_rows(void);
int rl8_mlp_tower_backward_f32(const float *x, const float *h1, const float *h2,
                               const float *dout, int64_t m, int d_in, const float *w2t_packed,
                               const float *w3, int n_out, float *dz2_out, float *partials,
                               int *partial_rows_out /*host*/, void *stream);

/* dW2 (+)= dZ2^T h1 on bf16 planes, with h1 = relu(x W1^T + b1) recomputed from the
 * observations instead of read back (workspace: rl8_mlp_wgrad_workspace_bytes()). */
int rl8_mlp_wgrad_split_f32(const float *dz2, const float *x, const float *w1, const float *b1,
                            int64_t m, int d_in, float *workspace, float *dw2_out, int accumulate,
                            void *stream);

/* The fused backward: rl8_mlp_tower_backward_f16_f32 followed by this call, whose weight-gradient
 * kernel re-forms dZ2 = (dOut x W3) * (h2 > 0) and h1 itself (thread = column) and
 * accumulates the head gradients on the way.  dw2_out [256][256] receives dW2; the
 * head segments [db2 | dW3] of `partials` (the same buffer and row count the first
 * call reported) are filled and the db3 segment is zeroed: db3 = the column sums of
 * dOut, which the caller forms itself.  Widths: rl8_mlp_backward_f16_supports.  Since round 3 the
 * products run on two fp16 planes per operand, each scaled by a power of two per column of the
 * output it indexes (three plane products per 16 rows; RL8_WGRAD_PLANES=bf16: the six bf16 ones).
 * Inputs longer than 2^23 rows are summed in 2^23-row segments, in order (this call and
 * rl8_mlp_wgrad_split_f32): the fp32 accumulation chains do not grow with m. */
int rl8_mlp_wgrad_fused_split_f32(const float *h2, const float *dout, const float *x, const float *w1,
                                  const float *b1, const float *w3, int64_t m, int d_in, int n_out,
                                  float *workspace, float *dw2_out, float *partials, void *stream);

/* Weight gradient of the 256x256 layer: dw2_out [256][256] (+)= dZ2^T h1 over M
 * rows (fp32 MFMA; per-workgroup partial slabs in `workspace`, summed in a fixed
 * order).  workspace: rl8_mlp_wgrad_workspace_bytes() bytes (one 256 x 256 slab per CU + 256 bytes for the
 * fp16-plane kernels: operand bounds, the guard's sample counts and decision -- written per call -- and, in the
 * last 128 bytes, two uint32 counters that only grow: [32] calls that consulted the guard, [33] calls it sent to
 * the exact bf16 planes).  No initialisation needed; zero the last 256 bytes once if the counters are to be read.
 *
 * The guard (round 4): the fp16 planes of rl8_mlp_wgrad_fused_split_f32 / rl8_mlp_wgrad_gate_bits_f32 carry 22 bits
 * of a term within 2^-17 of its column's bound.  Each call samples dOut (one KiB in sixteen) behind the pass that
 * takes the bounds; if more than 2^-7 of the non-zero entries lie below 2^-12 of the largest, the call's sums are
 * formed on three exact bf16 planes instead (both generations are launched, one leaves at once: no host round
 * trip).  RL8_WGRAD_PLANES / RL8_WGRAD_GATE_PLANES = bf16: always the exact planes; = f16!: never (diagnostics). */
int64_t rl8_mlp_wgrad_workspace_bytes(void);
int rl8_mlp_wgrad_f32(const float *dz2, const float *h1, int64_t m, float *workspace,
                      float *dw2_out, int accumulate, void *stream);
/* The same product over rows that sit `pitch` floats apart (multiples of 4,
 * >= 256): e.g. one gate of an LSTM's [M][4][256] gate gradients, pitch 1024. */
int rl8_mlp_wgrad_strided_f32(const float *dz2, int64_t dz2_pitch, const float *h1, int64_t h1_pitch,
                              int64_t m, float *workspace, float *dw2_out, int accumulate,
                              void *stream);
/* The same product on the bf16 matrix pipe (fp32-accurate bf16-plane products, both
 * operands prefetched from memory): the LSTM's dW_hh[q] = dG_q^T h_{t-1}
 * (autograd of src/rl8/models/_recurrent.py:312-333).  With x / colsums it also leaves, per
 * workgroup, the column sums  sum_rows dZ[row][col] * x[row][i]  and  sum_rows dZ[row][col]:
 * the same gate's dW_ih and bias gradients, from the dZ values its producers hold anyway
 * (d_in in {1, 2, 3, 5}; the caller adds the *colsum_rows_out rows in order). */
int rl8_mlp_wgrad_split_strided_f32(const float *dz, int64_t dz_pitch, const float *h, int64_t h_pitch,
                                    int64_t m, float *workspace, float *dw_out, int accumulate,
                                    const float *x /* [m][d_in] dense, or NULL */, int d_in,
                                    float *colsums /* [<= 256 rows][256 * (d_in + 1)], or NULL */,
                                    int *colsum_rows_out, void *stream);

/* ---------------------------------------------------------------------- *
 * a-9, second generation: ONE LSTM timestep with the recurrent product as an
 *      fp32-accurate product on the fp16 matrix pipe (the towers' two-plane scheme: |h| < 1, so
 *      the state's planes are those of h * 2^14; W_hh carries one power of two for the matrix)
 *      src/rl8/models/_recurrent.py:312-333 (the nn.LSTM call of the default recurrent
 *      models), algorithms/_recurrent.py:385-431 (its per-timestep use in collect())
 * The time loop is the caller's: one launch per timestep, states exchanged through HBM.
 *
 * rl8_lstm_pack_split: torch-layout parameters -> `packed` (rl8_lstm_split_packed_bytes()
 *   bytes: W_hh as two fp16 planes in the kernel's fragment order + its power of two) and `wb`
 *   (rl8_lstm_split_wb_floats() floats: [1024][8] = [w_ih row | 0.. | b_ih + b_hh]).
 * rl8_lstm_split_state: h [B][pitch] fp32 -> `planes` (rl8_lstm_split_state_bytes(B) bytes),
 *   the step kernel's A operand.  rl8_lstm_split_state_bound: the same, and max |h| over the rows as a float's bit
 *   pattern in *bound_out (what rl8_lstm_wgrad_f16_f32 takes as h_bound for the first timestep).
 * rl8_lstm_step_split_f32: x row r at x + r * x_pitch (d_in floats); h_planes of h_{t-1};
 *   c_prev row r at c_prev + r * c_prev_pitch; writes h_t, c_t rows at the given pitches
 *   (floats) and, when `gates` is not NULL, the post-activation gates i, f, g, o as
 *   [4][256] per row at gates + r * gates_pitch (what rl8_lstm_backward_f32 reads); when
 *   `planes_out` is not NULL, h_t also as fp16 planes (rl8_lstm_split_state's layout) for the
 *   next timestep's call, so that only the first step of a sequence needs rl8_lstm_split_state.
 *   d_in in {1, 2, 3, 5} (rl8_lstm_split_supports); other widths keep rl8_lstm_forward_f32.
 * ---------------------------------------------------------------------- */
int rl8_lstm_split_supports(int d_in);
int64_t rl8_lstm_split_packed_bytes(void);
int64_t rl8_lstm_split_wb_floats(void);
int64_t rl8_lstm_split_state_bytes(int64_t b);
int rl8_lstm_pack_split(const float *w_ih, const float *w_hh, const float *b_ih, const float *b_hh,
                        int d_in, void *packed, float *wb, void *stream);
int rl8_lstm_split_state(const float *h, int64_t pitch, int64_t b, void *planes, void *stream);
int rl8_lstm_split_state_bound(const float *h, int64_t pitch, int64_t b, void *planes, uint32_t *bound_out,
                               void *stream);
int rl8_lstm_step_split_f32(const float *x, int64_t x_pitch, int d_in, const void *h_planes,
                            const float *c_prev, int64_t c_prev_pitch, const void *w_planes,
                            const float *wb, int64_t b, float *h_out, int64_t h_out_pitch, float *c_out,
                            int64_t c_out_pitch, float *gates, int64_t gates_pitch,
                            void *planes_out /* h_t as planes for the next step (another buffer than h_planes), or NULL */,
                            void *stream);

/* ---------------------------------------------------------------------- *
 * a-9  Default recurrent models' LSTM, fused
 *      src/rl8/models/_recurrent.py:201-321 (torch.nn.LSTM(d_in, 256, num_layers=1,
 *      batch_first=True) inside DefaultContinuous/DiscreteRecurrentModel)
 * One layer, hidden 256, fp32 (matrix products on v_mfma_f32_32x32x2_f32), gate
 * order i, f, g, o as in torch, 1 <= d_in <= 7 (rl8_lstm_supports).
 *
 * rl8_lstm_pack_f32: torch-layout parameters (w_ih [1024][d_in], w_hh [1024][256],
 * b_ih, b_hh [1024]) -> `packed` (rl8_lstm_pack_floats() floats): per gate, the
 * 256 x (256 + 8) matrix [w_hh | w_ih | b_ih + b_hh | 0] in MFMA fragment order.
 *
 * rl8_lstm_forward_f32: x [B][L][d_in] dense; h0, c0 [B][256] -> hs [B][L][256]
 * (every h_t), hn, cn [B][256].  For training, save_gates [B][L][4][256]
 * (post-activation i, f, g, o) and save_c [B][L][256] receive what
 * rl8_lstm_backward_f32 needs; both NULL for rollouts.
 * ---------------------------------------------------------------------- */
int rl8_lstm_supports(int d_in);
int64_t rl8_lstm_pack_floats(void);
int rl8_lstm_pack_f32(const float *w_ih, const float *w_hh, const float *b_ih, const float *b_hh,
                      int d_in, float *packed, void *stream);
int rl8_lstm_forward_f32(const float *x, int64_t b, int l, int d_in, const float *h0, const float *c0,
                         const float *w_packed, float *hs, float *hn, float *cn, float *save_gates,
                         float *save_c, void *stream);

/* Backward through time ("dgrad" half): given dhs [B][L][256] (gradient of every
 * h_t output) and what the forward saved, writes the pre-activation gate
 * gradients dgates [B][L][4][256] and `*partial_rows_out` rows
 * (<= rl8_lstm_backward_max_rows()) of rl8_lstm_backward_partial_floats(d_in)
 * floats into `partials`: [dW_ih (1024 * d_in) | db (1024)], whose column sums are
 * the gradients of w_ih and of each bias vector.  whht_packed [4][65536]: block q =
 * rl8_mlp_pack_w2_f32(w_hh[256q : 256q + 256], transposed = 1).  The recurrent
 * weight gradient is dW_hh[256q : 256q + 256] = rl8_mlp_wgrad_strided_f32(dgates +
 * 256q, 1024, h_prev, 256, B*L, ...) with h_prev[b][t] = h_{t-1} (h0 at t = 0).
 * No gradient is produced for x, h0, c0 or the final states.  * partials == NULL: the data-gradient kernel only (dG out); the input-weight and bias
 * gradients then come from rl8_mlp_wgrad_split_strided_f32's column sums.
 */
int64_t rl8_lstm_backward_partial_floats(int d_in);
int rl8_lstm_backward_max_rows(void);
int rl8_lstm_backward_f32(const float *x, int64_t b, int l, int d_in, const float *c0,
                          const float *gates, const float *cs, const float *dhs,
                          const float *whht_packed, float *dgates, float *partials,
                          int *partial_rows_out /*host*/, void *stream);

/* The same data gradient with the recurrent product dL/dh_{t-1} = sum_q dG_q x W_hh[q] on the
 * 16-bit matrix pipe (bf16 planes, three per operand, six plane products: fp32-accurate), a
 * wave per 32 sequences for all L steps (lstm_rows_kernels.hip).  What torch.nn.LSTM's backward
 * computes for the module of src/rl8/models/_recurrent.py:201-321 when
 * src/rl8/algorithms/_recurrent.py calls loss.backward().
 * rl8_lstm_rows_backward_pack: w_hh [1024][256] (torch layout, gates i, f, g, o) -> `packed`
 *   (rl8_lstm_rows_backward_pack_bytes() bytes, 16-byte aligned): the planes of W_hh^T in the
 *   order the kernel streams them.  Re-pack when w_hh changes.
 * rl8_lstm_rows_backward_f32: c0 [B][256], gates [B][L][4][256] and cs [B][L][256] as the
 *   forward saved them, dhs [B][L][256] -> dgates [B][L][4][256]; `dc_scratch` [B][256] floats
 *   (contents irrelevant on entry, undefined on return); dg_bound_out (device word, may be NULL) receives the bit
 *   pattern of max |dgates|: the bound rl8_mlp_wgrad_f16_strided_f32 scales its fp16 planes by.  All arrays 16-byte
 *   aligned; 32 * L * 4096 < 2^31.  The input-weight, bias and recurrent-weight gradients come from
 *   rl8_mlp_wgrad_split_strided_f32 on dgates as before. */
int64_t rl8_lstm_rows_backward_pack_bytes(void);
int rl8_lstm_rows_backward_pack(const float *w_hh, void *packed, void *stream);
int rl8_lstm_rows_backward_f32(int64_t b, int l, const float *c0, const float *gates, const float *cs,
                               const float *dhs, const void *packed, float *dgates, float *dc_scratch,
                               uint32_t *dg_bound_out, void *stream);
/* The same with dL/dh_t of the output heads formed inside from heads_dout [b][l][4] (the gradient of the heads' outputs,
 * zero-padded to four per row-step) and heads_w [4][256] (their nn.Linear weights stacked, zero rows past their number):
 * the [b][l][256] array is neither written by rl8_linear_heads_backward_f32 (dh_out NULL) nor read here. */
int rl8_lstm_rows_backward_heads_f32(int64_t b, int l, const float *c0, const float *gates, const float *cs,
                                     const float *heads_dout, const float *heads_w, const void *packed, float *dgates,
                                     float *dc_scratch, uint32_t *dg_bound_out, void *stream);
/* rl8_mlp_wgrad_split_strided_f32 on fp16 planes (three plane products instead of six): dz and h each scaled by one
 * power of two taken from *dz_bound / *h_bound -- device words holding a float >= max |dz| / max |h| over all rows
 * read (dg_bound_out above; 1.0f for an LSTM's outputs). */
int rl8_mlp_wgrad_f16_strided_f32(const float *dz, int64_t dz_pitch, const uint32_t *dz_bound, const float *h,
                                  int64_t h_pitch, const uint32_t *h_bound, int64_t m, float *workspace,
                                  float *dw_out, int accumulate, const float *x, int d_in, float *colsums,
                                  int *colsum_rows_out, void *stream);

/* The four gates of one LSTM timestep in one launch of the fp16-plane weight gradient: dz = the step's dG rows ([m] rows
 * of dz_pitch >= 1024 floats, gate q at columns [256 q, 256 q + 256)), dw_out [4][256][256] (+)= dG_q^T h per gate,
 * colsums (optional, with x / d_in) [4][*colsum_rows_out][256 (d_in + 1)]; h is then read from HBM once instead of four
 * times.  m >= 128 (RL8_ESIZE below: use rl8_mlp_wgrad_f16_strided_f32 per gate); bounds as there. */
int rl8_lstm_wgrad_f16_f32(const float *dz, int64_t dz_pitch, const uint32_t *dz_bound, const float *h, int64_t h_pitch,
                           const uint32_t *h_bound, int64_t m, float *workspace, float *dw_out, int accumulate,
                           const float *x, int d_in, float *colsums, int *colsum_rows_out, void *stream);

/* The recurrent models' output heads (src/rl8/models/_recurrent.py:230-236, 287-292),
 * all of them at once: out [M][n] = h [M][256] x w^T + b, w [n][256] (the heads'
 * nn.Linear weights stacked), n <= 8.  Backward: dh_out [M][256] = dout x w and
 * `*partial_rows_out` rows (<= rl8_linear_heads_max_rows()) of [dW (n*256) | db (n)]
 * whose column sums are the parameter gradients; dh_out NULL: the parameter gradients only (the recurrent models'
 * backward forms dh inside rl8_lstm_rows_backward_heads_f32).
 * rl8_linear_heads_forward_pair_f32: two layers that stay separate arrays (the rollout's logits head and value head,
 * algorithms/_recurrent.py:400-420 via policy.sample()) in one pass over h: out_a [M][n_a], out_b [M][n_b],
 * n_a + n_b <= 8. */
int rl8_linear_heads_max_rows(void);
int rl8_linear_heads_forward_f32(const float *h, int64_t m, const float *w, const float *b, int n_out,
                                 float *out, void *stream);
int rl8_linear_heads_forward_pair_f32(const float *h, int64_t m, const float *w_a, const float *b_a, int n_a,
                                      float *out_a, const float *w_b, const float *b_b, int n_b, float *out_b,
                                      void *stream);
int rl8_linear_heads_backward_f32(const float *h, const float *dout, int64_t m, const float *w,
                                  int n_out, float *dh_out, float *partials,
                                  int *partial_rows_out /*host*/, void *stream);

/* ---- OPT-IN: towers of a SCALAR observation as exact piecewise-linear tables (round 4 prototype; VERDICT r3 item 10,
 * DESIGN.md section 9).  Special to d_in = 1 -- the dummy envs of BASELINE configs 2 and 4 -- and never a replacement
 * for the general tower kernels above, which every other shape runs and against which this path is tested.  The
 * table (breakpoints ascending, then per interval an anchor, n_out values at the anchor, n_out slopes) is built by
 * the caller in fp64 from the tower's weights (rl8_amd/nn/piecewise_mlp.py; reference arithmetic:
 * src/rl8/models/_feedforward.py:336-375) and handed over as floats:
 *   table = [ breaks[p] | anchor[p + 1] | value[(p + 1) * n_out] | slope[(p + 1) * n_out] ]
 *   out[r][q] = value[i][q] + slope[i][q] * (x[r] - anchor[i]),   i = number of breakpoints < x[r].
 * p <= rl8_pw_max_breaks(), n_out <= 3. */
int rl8_pw_max_breaks(void);
int rl8_pw_tower_forward_f32(const float *x, int64_t m, const float *table, int p, int n_out, float *out, void *stream);
/* Per interval i the sums over its rows of dOut and of dOut * x, all the backward pass of such a tower needs from the
 * rows: sums_out[i][0][q] = sum dout[r][q], sums_out[i][1][q] = sum dout[r][q] * x[r]  (fp64, [(p + 1)][2][n_out]).
 * Accumulated EXACTLY on a common power-of-two scale in 64-bit integers (two per sum, 74 bits), so the result does not
 * depend on the order of the atomics: bitwise reproducible.  m <= 2^25 per call.  workspace:
 * rl8_pw_workspace_bytes(p, n_out) bytes, 16-byte aligned, no initialisation.  A non-finite dOut or x makes every sum
 * NaN.  RL8_ESIZE when the accumulators do not fit LDS (p * n_out beyond ~3000): the caller keeps the general path. */
int64_t rl8_pw_workspace_bytes(int p, int n_out);
int rl8_pw_segment_sums_f32(const float *x, const float *dout, int64_t m, int n_out, const float *breaks, int p,
                            void *workspace, double *sums_out, void *stream);

#ifdef __cplusplus
}
#endif

#endif /* RL8_AMD_H */
