// K3: generalized advantage estimate, reverse scan over the rollout buffer.
// Restates src/rl8/nn/functional.py:100-123 of the reference.
//
// HBM-bound: per (env, t) cell the scan reads reward + value (8 B) and writes
// advantage + return (8 B); the reference's 5H+6 strided launches become one.
//
//  * time-major [H+1][N]: column t is a contiguous slab, so consecutive lanes
//    read consecutive envs (16 B/lane with 4 envs per thread); the per-env
//    recurrence lives in registers and values[t+1] is carried, not re-read.
//  * env-major [N][H+1] (the reference's own layout): a workgroup stages a
//    [E envs x Tc steps] tile of rewards and values through LDS with coalesced
//    loads along each env's contiguous row, each lane then scans one env out of
//    LDS (row stride kept odd => no bank conflicts) and the tile is written
//    back coalesced.  Long horizons are walked in time chunks, carrying
//    adv[t+1] and values[t+1] in registers across chunks.
//
// Both variants also emit fp64 (count, sum, sum of squares) of adv[:, :H] as
// per-block partials reduced in a fixed order (bitwise reproducible).
#include "common.hip.h"

namespace rl8 {

struct f4 {
  float v[4];
};

template <int VEC>
struct VecIO;

template <>
struct VecIO<4> {
  __device__ __forceinline__ static f4 load(const float *p) {
    const float4 t = *reinterpret_cast<const float4 *>(p);
    return {{t.x, t.y, t.z, t.w}};
  }
  __device__ __forceinline__ static void store(float *p, const f4 &a) {
    *reinterpret_cast<float4 *>(p) = make_float4(a.v[0], a.v[1], a.v[2], a.v[3]);
  }
};

template <>
struct VecIO<1> {
  __device__ __forceinline__ static f4 load(const float *p) { return {{*p, 0.f, 0.f, 0.f}}; }
  __device__ __forceinline__ static void store(float *p, const f4 &a) { *p = a.v[0]; }
};

template <int VEC>
__global__ __launch_bounds__(kBlock) void gae_scan_time_major_kernel(
    float *__restrict__ rewards, const float *__restrict__ values, float *__restrict__ adv,
    float *__restrict__ ret, int64_t n, int64_t h, float gamma, float gamma_lambda, float denom,
    int write_back, double *__restrict__ partials) {
  __shared__ double smem[2 * kWavesPerBlock];
  double acc[2] = {0.0, 0.0};
  const int64_t stride_e = (int64_t)gridDim.x * kBlock * VEC;
  for (int64_t e0 = ((int64_t)blockIdx.x * kBlock + threadIdx.x) * VEC; e0 < n; e0 += stride_e) {
    f4 v_next = VecIO<VEC>::load(values + h * n + e0);
    f4 prev = {{0.f, 0.f, 0.f, 0.f}};
    {
      f4 ret_h;
#pragma unroll
      for (int i = 0; i < VEC; ++i) ret_h.v[i] = 0.0f + v_next.v[i];
      VecIO<VEC>::store(adv + h * n + e0, prev);
      VecIO<VEC>::store(ret + h * n + e0, ret_h);
      if (write_back) {
        f4 r = VecIO<VEC>::load(rewards + h * n + e0);
#pragma unroll
        for (int i = 0; i < VEC; ++i) r.v[i] = r.v[i] / denom;
        VecIO<VEC>::store(rewards + h * n + e0, r);
      }
    }
#pragma unroll 4
    for (int64_t t = h - 1; t >= 0; --t) {
      f4 r = VecIO<VEC>::load(rewards + t * n + e0);
      const f4 v = VecIO<VEC>::load(values + t * n + e0);
      f4 q;
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        r.v[i] = r.v[i] / denom;
        const float delta = r.v[i] + (gamma * v_next.v[i] - v.v[i]);
        prev.v[i] = delta + gamma_lambda * prev.v[i];
        q.v[i] = prev.v[i] + v.v[i];
        acc[0] += (double)prev.v[i];
        acc[1] += (double)prev.v[i] * (double)prev.v[i];
      }
      VecIO<VEC>::store(adv + t * n + e0, prev);
      VecIO<VEC>::store(ret + t * n + e0, q);
      if (write_back) VecIO<VEC>::store(rewards + t * n + e0, r);
      v_next = v;
    }
  }
  block_reduce<2, SumOp>(acc, smem);
  if (threadIdx.x == 0) {
    partials[(int64_t)blockIdx.x * kPartialWidth + 0] = acc[0];
    partials[(int64_t)blockIdx.x * kPartialWidth + 1] = acc[1];
  }
}

// Env-major, LDS-staged.  Dynamic LDS: two [E][S] float tiles (rewards->adv,
// values->ret).  blockDim = kBlock; lanes [0, E) scan.
__global__ __launch_bounds__(kBlock) void gae_scan_env_major_kernel(
    float *__restrict__ rewards, const float *__restrict__ values, float *__restrict__ adv,
    float *__restrict__ ret, int64_t n, int64_t h, float gamma, float gamma_lambda, float denom,
    int write_back, int envs_per_block, int chunk, int lds_stride,
    double *__restrict__ partials) {
  extern __shared__ float lds[];
  __shared__ double smem[2 * kWavesPerBlock];
  float *tile_r = lds;
  float *tile_v = lds + (int64_t)envs_per_block * lds_stride;
  const int64_t stride = h + 1;
  const int tid = threadIdx.x;
  double acc[2] = {0.0, 0.0};
  const int64_t tiles = (n + envs_per_block - 1) / envs_per_block;
  for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
  const int64_t e0 = tile * envs_per_block;
  const int ne = (int)((n - e0) < envs_per_block ? (n - e0) : envs_per_block);
  float prev = 0.0f, v_next = 0.0f;
  // Walk the H+1 columns from the end in chunks of `chunk` steps.
  for (int64_t t1 = stride; t1 > 0; t1 -= chunk) {
    const int64_t t0 = t1 - chunk > 0 ? t1 - chunk : 0;
    const int tc = (int)(t1 - t0);
    const int count = ne * tc;
    __syncthreads();  // previous chunk fully written out before the tile is reused
    for (int idx = tid; idx < count; idx += kBlock) {
      const int e = idx / tc, j = idx - e * tc;
      const int64_t g = (e0 + e) * stride + t0 + j;
      const float r = rewards[g] / denom;
      if (write_back) rewards[g] = r;
      tile_r[e * lds_stride + j] = r;
      tile_v[e * lds_stride + j] = values[g];
    }
    __syncthreads();
    if (tid < ne) {
      float *row_r = tile_r + tid * lds_stride;
      float *row_v = tile_v + tid * lds_stride;
      for (int j = tc - 1; j >= 0; --j) {
        const float v = row_v[j];
        if (t0 + j == h) {  // column H: adv = 0, ret = values (:105, :117)
          prev = 0.0f;
          row_r[j] = 0.0f;
          row_v[j] = 0.0f + v;
        } else {
          const float delta = row_r[j] + (gamma * v_next - v);
          prev = delta + gamma_lambda * prev;
          row_r[j] = prev;
          row_v[j] = prev + v;
          acc[0] += (double)prev;
          acc[1] += (double)prev * (double)prev;
        }
        v_next = v;
      }
    }
    __syncthreads();
    for (int idx = tid; idx < count; idx += kBlock) {
      const int e = idx / tc, j = idx - e * tc;
      const int64_t g = (e0 + e) * stride + t0 + j;
      adv[g] = tile_r[e * lds_stride + j];
      ret[g] = tile_v[e * lds_stride + j];
    }
  }
  }
  block_reduce<2, SumOp>(acc, smem);
  if (tid == 0) {
    partials[(int64_t)blockIdx.x * kPartialWidth + 0] = acc[0];
    partials[(int64_t)blockIdx.x * kPartialWidth + 1] = acc[1];
  }
}

__global__ void gae_moments_kernel(const double *__restrict__ partials, int rows, double count,
                                   double *__restrict__ out) {
  __shared__ double smem[2 * kWavesPerBlock];
  double acc[2] = {0.0, 0.0};
  for (int r = threadIdx.x; r < rows; r += kBlock) {
    acc[0] += partials[(int64_t)r * kPartialWidth + 0];
    acc[1] += partials[(int64_t)r * kPartialWidth + 1];
  }
  block_reduce<2, SumOp>(acc, smem);
  if (threadIdx.x == 0) {
    out[0] = count;
    out[1] = acc[0];
    out[2] = acc[1];
  }
}

struct NormConsts {
  float mean, sd;
};

__device__ __forceinline__ NormConsts norm_consts(const double *moments) {
  const double cnt = moments[0], s = moments[1], sq = moments[2];
  const double mean = s / cnt;
  double var = (sq - s * mean) / (cnt - 1.0);
  var = var > 0.0 ? var : 0.0;
  NormConsts c;
  c.mean = (float)mean;
  c.sd = (float)sqrt(var) + 1e-8f;
  return c;
}

// time-major: adv[:H] is the contiguous prefix [0, h*n).
template <int VEC>
__global__ __launch_bounds__(kBlock) void advantage_normalise_flat_kernel(
    float *__restrict__ adv, int64_t count, const double *__restrict__ moments) {
  const NormConsts c = norm_consts(moments);
  const int64_t step = (int64_t)gridDim.x * kBlock * VEC;
  for (int64_t i = ((int64_t)blockIdx.x * kBlock + threadIdx.x) * VEC; i < count; i += step) {
    f4 a = VecIO<VEC>::load(adv + i);
#pragma unroll
    for (int k = 0; k < VEC; ++k) a.v[k] = (a.v[k] - c.mean) / c.sd;
    VecIO<VEC>::store(adv + i, a);
  }
}

// env-major: every row's last column is left alone.
__global__ __launch_bounds__(kBlock) void advantage_normalise_env_major_kernel(
    float *__restrict__ adv, int64_t n, int64_t h, const double *__restrict__ moments) {
  const NormConsts c = norm_consts(moments);
  const int64_t total = n * (h + 1);
  const int64_t step = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += step) {
    if (i % (h + 1) != h) adv[i] = (adv[i] - c.mean) / c.sd;
  }
}

}  // namespace rl8

using namespace rl8;

RL8_API int rl8_gae_scan_f32(float *rewards, const float *values, float *adv_out, float *ret_out,
                             int64_t n, int64_t h, int layout, float gamma, float gamma_lambda,
                             float reward_denominator, int write_scaled_rewards,
                             double *moments_out, void *scratch, void *stream) {
  if (!rewards || !values || !adv_out || !ret_out || !moments_out || !scratch) return RL8_ENULL;
  if (n <= 0 || h <= 0) return RL8_ESIZE;
  if (layout != 0 && layout != 1) return RL8_ECONFIG;
  hipStream_t s = (hipStream_t)stream;
  double *partials = (double *)scratch;
  int rows;
  if (layout == 1) {
    const bool vec = (n % 4 == 0) && aligned16(rewards) && aligned16(values) &&
                     aligned16(adv_out) && aligned16(ret_out);
    const int per_block = kBlock * (vec ? 4 : 1);
    rows = grid_for(n, per_block);
    if (vec)
      gae_scan_time_major_kernel<4><<<rows, kBlock, 0, s>>>(
          rewards, values, adv_out, ret_out, n, h, gamma, gamma_lambda, reward_denominator,
          write_scaled_rewards, partials);
    else
      gae_scan_time_major_kernel<1><<<rows, kBlock, 0, s>>>(
          rewards, values, adv_out, ret_out, n, h, gamma, gamma_lambda, reward_denominator,
          write_scaled_rewards, partials);
  } else {
    const int64_t cols = h + 1;
    const int chunk = (int)(cols < 127 ? cols : 127);
    const int lds_stride = chunk | 1;  // odd => conflict-free column walks
    int e = (int)(65536 / ((int64_t)lds_stride * 8) / kWave) * kWave;
    if (e > kBlock) e = kBlock;
    if (e < kWave) e = kWave;
    rows = grid_for(n, e);
    const size_t lds_bytes = (size_t)2 * e * lds_stride * sizeof(float);
    gae_scan_env_major_kernel<<<rows, kBlock, lds_bytes, s>>>(
        rewards, values, adv_out, ret_out, n, h, gamma, gamma_lambda, reward_denominator,
        write_scaled_rewards, e, chunk, lds_stride, partials);
  }
  int st = launch_status();
  if (st != RL8_OK) return st;
  gae_moments_kernel<<<1, kBlock, 0, s>>>(partials, rows, (double)n * (double)h, moments_out);
  return launch_status();
}

RL8_API int rl8_advantage_normalise_f32(float *adv, int64_t n, int64_t h, int layout,
                                        const double *moments, void *stream) {
  if (!adv || !moments) return RL8_ENULL;
  if (n <= 0 || h <= 0) return RL8_ESIZE;
  hipStream_t s = (hipStream_t)stream;
  if (layout == 1) {
    const int64_t count = n * h;
    if (count % 4 == 0 && aligned16(adv))
      advantage_normalise_flat_kernel<4><<<grid_for(count, kBlock * 4), kBlock, 0, s>>>(adv, count,
                                                                                    moments);
    else
      advantage_normalise_flat_kernel<1><<<grid_for(count, kBlock), kBlock, 0, s>>>(adv, count,
                                                                                moments);
  } else if (layout == 0) {
    advantage_normalise_env_major_kernel<<<grid_for(n * (h + 1), kBlock), kBlock, 0, s>>>(adv, n, h,
                                                                                      moments);
  } else {
    return RL8_ECONFIG;
  }
  return launch_status();
}
