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
#include <stdlib.h>

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

// Publishes (sum, sum_sq) of this block; the last block to arrive adds the rows
// in order and writes moments_out = (count, sum, sum_sq).
__device__ __forceinline__ void publish_moments(double (&acc)[2], double *partials, double count,
                                                double *moments_out, double *smem) {
  if (threadIdx.x == 0) {
    publish_partial(partials + (int64_t)blockIdx.x * kPartialWidth + 0, acc[0]);
    publish_partial(partials + (int64_t)blockIdx.x * kPartialWidth + 1, acc[1]);
  }
  if (!last_block_arrives(ticket_word(partials))) return;
  double tot[2] = {0.0, 0.0};
  fold_partial_rows<2>(partials, (int)gridDim.x, [&](const double(&row)[2]) {
    tot[0] += row[0];
    tot[1] += row[1];
  });
  block_reduce<2, SumOp>(tot, smem);
  if (threadIdx.x == 0) {
    moments_out[0] = count;
    moments_out[1] = tot[0];
    moments_out[2] = tot[1];
    *ticket_word(partials) = 0u;
  }
}

template <int VEC>
__global__ __launch_bounds__(kBlock) void gae_scan_time_major_kernel(
    float *__restrict__ rewards, const float *__restrict__ values, float *__restrict__ adv,
    float *__restrict__ ret, int64_t n, int64_t h, float gamma, float gamma_lambda, float denom,
    int write_back, double *__restrict__ partials, double *__restrict__ moments_out) {
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
  publish_moments(acc, partials, (double)n * (double)h, moments_out, smem);
}

// Env-major, LDS-staged.  One lane per env: blockDim.x == envs_per_block.
// Dynamic LDS: two [E][S] float tiles (rewards -> advantages, values -> returns).
//
// FLAT == true (whole rows in one chunk and S == H+1, i.e. H+1 odd): the tile's
// global footprint is one contiguous run of E*(H+1) floats whose layout equals
// the LDS layout, so staging is a straight 16-byte-per-lane copy with several
// loads in flight per lane, and the odd row stride keeps the per-env column
// walk conflict-free.  Otherwise rows are padded to an odd stride and copied
// element-wise (time chunks of long horizons, even H+1).
constexpr int kStageUnroll = 4;

template <bool FLAT>
__global__ void gae_scan_env_major_kernel(
    float *__restrict__ rewards, const float *__restrict__ values, float *__restrict__ adv,
    float *__restrict__ ret, int64_t n, int64_t h, float gamma, float gamma_lambda, float denom,
    int write_back, int chunk, int lds_stride, double *__restrict__ partials,
    double *__restrict__ moments_out) {
  extern __shared__ float lds[];
  __shared__ double smem[2 * kWavesPerBlock];
  const int envs_per_block = blockDim.x;
  const int nthreads = blockDim.x;
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
      __syncthreads();  // previous tile / chunk fully written out before reuse
      if (FLAT) {
        const int64_t base = e0 * stride;
        const int nvec = count >> 2;
        const float4 *gr = reinterpret_cast<const float4 *>(rewards + base);
        const float4 *gv = reinterpret_cast<const float4 *>(values + base);
        float4 *lr = reinterpret_cast<float4 *>(tile_r);
        float4 *lv = reinterpret_cast<float4 *>(tile_v);
        // kStageUnroll loads in flight per lane.  The guarded form of this loop (`if (i < nvec)` around each load
        // into r4[u] / v4[u]) left the two arrays in SCRATCH (80 bytes per thread: round 2's "WRITE_SIZE 1.44x the
        // bytes it stores, cause not found" -- rocprof counted the spill traffic); full groups are therefore
        // unguarded, the last partial group goes one element at a time.
        int i0 = tid;
        for (; i0 + (kStageUnroll - 1) * nthreads < nvec; i0 += nthreads * kStageUnroll) {
          float4 r4[kStageUnroll], v4[kStageUnroll];
#pragma unroll
          for (int u = 0; u < kStageUnroll; ++u) {
            r4[u] = gr[i0 + u * nthreads];
            v4[u] = gv[i0 + u * nthreads];
          }
#pragma unroll
          for (int u = 0; u < kStageUnroll; ++u) {
            const int i = i0 + u * nthreads;
            r4[u].x = r4[u].x / denom; r4[u].y = r4[u].y / denom;
            r4[u].z = r4[u].z / denom; r4[u].w = r4[u].w / denom;
            lr[i] = r4[u];
            lv[i] = v4[u];
            if (write_back) reinterpret_cast<float4 *>(rewards + base)[i] = r4[u];
          }
        }
        for (int i = i0; i < nvec; i += nthreads) {
          float4 r = gr[i];
          const float4 v = gv[i];
          r.x = r.x / denom; r.y = r.y / denom; r.z = r.z / denom; r.w = r.w / denom;
          lr[i] = r;
          lv[i] = v;
          if (write_back) reinterpret_cast<float4 *>(rewards + base)[i] = r;
        }
        for (int i = (nvec << 2) + tid; i < count; i += nthreads) {  // < 4 leftovers
          const float r = rewards[base + i] / denom;
          if (write_back) rewards[base + i] = r;
          tile_r[i] = r;
          tile_v[i] = values[base + i];
        }
      } else {
        for (int i0 = tid; i0 < count; i0 += nthreads * kStageUnroll) {
          float r1[kStageUnroll], v1[kStageUnroll];
#pragma unroll
          for (int u = 0; u < kStageUnroll; ++u) {
            const int idx = i0 + u * nthreads;
            if (idx < count) {
              const int e = idx / tc, j = idx - e * tc;
              const int64_t g = (e0 + e) * stride + t0 + j;
              r1[u] = rewards[g];
              v1[u] = values[g];
            }
          }
#pragma unroll
          for (int u = 0; u < kStageUnroll; ++u) {
            const int idx = i0 + u * nthreads;
            if (idx < count) {
              const int e = idx / tc, j = idx - e * tc;
              const float r = r1[u] / denom;
              if (write_back) rewards[(e0 + e) * stride + t0 + j] = r;
              tile_r[e * lds_stride + j] = r;
              tile_v[e * lds_stride + j] = v1[u];
            }
          }
        }
      }
      __syncthreads();
      if (tid < ne) {
        float *row_r = tile_r + tid * lds_stride;
        float *row_v = tile_v + tid * lds_stride;
#pragma unroll 4
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
      if (FLAT) {
        const int64_t base = e0 * stride;
        const int nvec = count >> 2;
        float4 *ga = reinterpret_cast<float4 *>(adv + base);
        float4 *gq = reinterpret_cast<float4 *>(ret + base);
        const float4 *lr = reinterpret_cast<const float4 *>(tile_r);
        const float4 *lv = reinterpret_cast<const float4 *>(tile_v);
        for (int i = tid; i < nvec; i += nthreads) {
          ga[i] = lr[i];
          gq[i] = lv[i];
        }
        for (int i = (nvec << 2) + tid; i < count; i += nthreads) {
          adv[base + i] = tile_r[i];
          ret[base + i] = tile_v[i];
        }
      } else {
        for (int idx = tid; idx < count; idx += nthreads) {
          const int e = idx / tc, j = idx - e * tc;
          const int64_t g = (e0 + e) * stride + t0 + j;
          adv[g] = tile_r[e * lds_stride + j];
          ret[g] = tile_v[e * lds_stride + j];
        }
      }
    }
  }
  block_reduce<2, SumOp>(acc, smem);
  publish_moments(acc, partials, (double)n * (double)h, moments_out, smem);
}

// The flat case again (whole rows in one chunk, H + 1 odd and <= 4 KMAX), software-pipelined over the workgroup's tiles.
// The kernel above runs its three phases one behind the other, and a wave's vector-memory operations retire through ONE
// in-order counter: the wait for a tile's loads also waits for the stores of the tile before it, so the chip alternates
// between reading and writing (loads alone 71 us, loads + stores 147 us per 2^20 x 32; RL8_GAE_DIAG runs of round 3).
// Here the NEXT tile's rows are requested into registers before this tile's scan: they are older than this tile's
// stores, land under the scan, and the wait for them (written by the compiler: all but the stores issued behind them)
// no longer sees the stores.  Buffer descriptors end at the tile's last float: no guarded tails.
template <int KMAX>
__global__ __launch_bounds__(256) void gae_scan_env_major_pipelined_kernel(
    float *__restrict__ rewards, const float *__restrict__ values, float *__restrict__ adv,
    float *__restrict__ ret, int64_t n, int64_t h, float gamma, float gamma_lambda, float denom,
    int write_back, double *__restrict__ partials, double *__restrict__ moments_out) {
  extern __shared__ float lds[];
  __shared__ double smem[2 * kWavesPerBlock];
  const int envs_per_block = blockDim.x, nthreads = blockDim.x, tid = threadIdx.x;
  const int stride = (int)h + 1;  // = the LDS row stride (odd: conflict-free column walks)
  float *tile_r = lds, *tile_v = lds + envs_per_block * stride;
  typedef unsigned int u4 __attribute__((ext_vector_type(4)));
  double acc[2] = {0.0, 0.0};
  const int64_t tiles = (n + envs_per_block - 1) / envs_per_block;
  auto envs_of = [&](int64_t tile) {
    const int64_t left = tile < tiles ? n - tile * envs_per_block : 0;
    return (int)(left < envs_per_block ? left : envs_per_block);
  };
  auto rsrc_of = [&](const float *base, int64_t tile) {
    const int ne = envs_of(tile);
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(ne > 0 ? base + tile * envs_per_block * stride : base), 0,
                                             ne * stride * 4, 0x00020000);
  };
  u4 r4[KMAX], v4[KMAX];
  auto request = [&](int64_t tile) {
    const __amdgpu_buffer_rsrc_t rr = rsrc_of(rewards, tile), vr = rsrc_of(values, tile);
#pragma unroll
    for (int u = 0; u < KMAX; ++u) {
      r4[u] = __builtin_amdgcn_raw_buffer_load_b128(rr, (tid + u * nthreads) * 16, 0, 0);
      v4[u] = __builtin_amdgcn_raw_buffer_load_b128(vr, (tid + u * nthreads) * 16, 0, 0);
    }
  };
  int64_t tile = blockIdx.x;
  request(tile);
  for (; tile < tiles; tile += gridDim.x) {
    const int ne = envs_of(tile);
    const int nvec = (ne * stride + 3) >> 2;
    const __amdgpu_buffer_rsrc_t wr = rsrc_of(rewards, tile);
    // registers -> LDS (the previous tile's stores have read it: barrier at the end of the loop body)
#pragma unroll
    for (int u = 0; u < KMAX; ++u) {
      const int i = tid + u * nthreads;
      if (i < nvec) {
        float4 r;
        r.x = __uint_as_float(r4[u][0]) / denom, r.y = __uint_as_float(r4[u][1]) / denom;
        r.z = __uint_as_float(r4[u][2]) / denom, r.w = __uint_as_float(r4[u][3]) / denom;
        reinterpret_cast<float4 *>(tile_r)[i] = r;
        reinterpret_cast<float4 *>(tile_v)[i] =
            make_float4(__uint_as_float(v4[u][0]), __uint_as_float(v4[u][1]), __uint_as_float(v4[u][2]), __uint_as_float(v4[u][3]));
        if (write_back)
          __builtin_amdgcn_raw_buffer_store_b128(
              u4{__float_as_uint(r.x), __float_as_uint(r.y), __float_as_uint(r.z), __float_as_uint(r.w)}, wr, i * 16, 0, 0);
      }
    }
    __syncthreads();
    request(tile + gridDim.x);  // in flight through the scan and the stores below; zero records past the last tile
    if (tid < ne) {
      float *row_r = tile_r + tid * stride, *row_v = tile_v + tid * stride;
      float prev = 0.0f, v_next = 0.0f;
#pragma unroll 4
      for (int j = stride - 1; j >= 0; --j) {
        const float v = row_v[j];
        if (j == (int)h) {  // column H: adv = 0, ret = values (:105, :117)
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
    const __amdgpu_buffer_rsrc_t ar = rsrc_of(adv, tile), qr = rsrc_of(ret, tile);
#pragma unroll
    for (int u = 0; u < KMAX; ++u) {
      const int i = tid + u * nthreads;
      if (i < nvec) {
        const float4 a4 = reinterpret_cast<const float4 *>(tile_r)[i], q4 = reinterpret_cast<const float4 *>(tile_v)[i];
        __builtin_amdgcn_raw_buffer_store_b128(
            u4{__float_as_uint(a4.x), __float_as_uint(a4.y), __float_as_uint(a4.z), __float_as_uint(a4.w)}, ar, i * 16, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(
            u4{__float_as_uint(q4.x), __float_as_uint(q4.y), __float_as_uint(q4.z), __float_as_uint(q4.w)}, qr, i * 16, 0, 0);
      }
    }
    __syncthreads();  // the tile has been read out: the next one may overwrite it
  }
  block_reduce<2, SumOp>(acc, smem);
  publish_moments(acc, partials, (double)n * (double)h, moments_out, smem);
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
    // 2 workgroups per CU measured best (5.3 TB/s vs 5.1 at 4/CU, 3.8 at 1/CU).
    static const int tm_cap = env_int("RL8_GAE_GRID_CAP");
    rows = grid_for(n, per_block, tm_cap > 0 ? tm_cap : 2 * kCUs);
    if (vec)
      gae_scan_time_major_kernel<4><<<rows, kBlock, 0, s>>>(
          rewards, values, adv_out, ret_out, n, h, gamma, gamma_lambda, reward_denominator,
          write_scaled_rewards, partials, moments_out);
    else
      gae_scan_time_major_kernel<1><<<rows, kBlock, 0, s>>>(
          rewards, values, adv_out, ret_out, n, h, gamma, gamma_lambda, reward_denominator,
          write_scaled_rewards, partials, moments_out);
  } else {
    const int64_t cols = h + 1;
    const int chunk = (int)(cols < 127 ? cols : 127);
    const int lds_stride = chunk | 1;  // odd => conflict-free column walks
    // One lane per env; as many envs per block as ~72 KB of LDS allow (2 blocks
    // per CU), in whole waves.  RL8_GAE_ENVS_PER_BLOCK overrides (tuning).
    static int env_override = -1;
    if (env_override < 0) {
      const char *v = getenv("RL8_GAE_ENVS_PER_BLOCK");
      env_override = v ? atoi(v) : 0;
    }
    int e = (int)(73728 / ((int64_t)lds_stride * 8) / kWave) * kWave;
    if (e > 256) e = 256;
    // (the pipelined flat kernel below: 128 envs per workgroup, two workgroups per CU measured best -- 122.6-124.2 us per
    // 2^20 x 32 against 126.5-128 for the other shapes)
    const bool pipelined_shape = chunk == cols && (cols & 1) && cols <= 36;
    if (pipelined_shape && e > 128) e = 128;
    if (env_override >= kWave && env_override <= 256 && env_override % kWave == 0) e = env_override;
    if (e < kWave) e = kWave;
    const size_t lds_bytes = (size_t)2 * e * lds_stride * sizeof(float);
    const bool flat = chunk == cols && lds_stride == cols && (e % 4 == 0) && aligned16(rewards) &&
                      aligned16(values) && aligned16(adv_out) && aligned16(ret_out);
    static LdsOptIn lds_attr_set_0;
    if (const int e_lds_attr_set_0 = allow_dynamic_lds(lds_attr_set_0, reinterpret_cast<const void *>(&gae_scan_env_major_kernel<true>), 160 * 1024)) return e_lds_attr_set_0;
    static LdsOptIn lds_attr_set_1;
    if (const int e_lds_attr_set_1 = allow_dynamic_lds(lds_attr_set_1, reinterpret_cast<const void *>(&gae_scan_env_major_kernel<false>), 160 * 1024)) return e_lds_attr_set_1;
    rows = grid_for(n, e);
    if (flat && pipelined_shape) {
      static LdsOptIn lds_attr9_0;
      if (const int e_lds_attr9_0 = allow_dynamic_lds(lds_attr9_0, reinterpret_cast<const void *>(&gae_scan_env_major_pipelined_kernel<9>), 160 * 1024)) return e_lds_attr9_0;
      // as many workgroups as are resident at once (each runs its tiles back to back, the next one's rows in flight):
      // a second round of workgroups would start with nothing requested
      static const int per_cu_cap = env_int("RL8_GAE_BLOCKS_PER_CU");
      int per_cu = (int)((160 * 1024) / (lds_bytes + 16 + 1024));
      if (per_cu > 2) per_cu = 2;
      if (per_cu_cap > 0 && per_cu_cap < per_cu) per_cu = per_cu_cap;
      if (per_cu < 1) per_cu = 1;
      if (rows > kCUs * per_cu) rows = kCUs * per_cu;
      // (a lane's sixteen-byte piece of the tile's tail may reach past E * cols floats by up to 12 bytes: + one vector)
      gae_scan_env_major_pipelined_kernel<9><<<rows, e, lds_bytes + 16, s>>>(
          rewards, values, adv_out, ret_out, n, h, gamma, gamma_lambda, reward_denominator, write_scaled_rewards, partials,
          moments_out);
    } else if (flat)
      gae_scan_env_major_kernel<true><<<rows, e, lds_bytes, s>>>(
          rewards, values, adv_out, ret_out, n, h, gamma, gamma_lambda, reward_denominator,
          write_scaled_rewards, chunk, lds_stride, partials, moments_out);
    else
      gae_scan_env_major_kernel<false><<<rows, e, lds_bytes, s>>>(
          rewards, values, adv_out, ret_out, n, h, gamma, gamma_lambda, reward_denominator,
          write_scaled_rewards, chunk, lds_stride, partials, moments_out);
  }
  return launch_status();
}

RL8_API int rl8_advantage_normalise_f32(float *adv, int64_t n, int64_t h, int layout,
                                        const double *moments, void *stream) {
  if (!adv || !moments) return RL8_ENULL;
  if (n <= 0 || h <= 0) return RL8_ESIZE;
  hipStream_t s = (hipStream_t)stream;
  if (layout == 1) {
    const int64_t count = n * h;
    static const int cap = env_int("RL8_NORM_GRID_CAP");
    if (count % 4 == 0 && aligned16(adv))
      advantage_normalise_flat_kernel<4>
          <<<grid_for(count, kBlock * 4, cap > 0 ? cap : kMaxGrid), kBlock, 0, s>>>(adv, count,
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
