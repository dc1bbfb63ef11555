// K6 rollout statistics, K5 minibatch gather, and the ABI housekeeping calls.
//
// K6 restates src/rl8/algorithms/_feedforward.py:411-436: per-env returns
// (sum over the horizon), min / max / mean / unbiased std of returns and of
// rewards, and std of the reversed discounted returns -- eight eager reductions
// and nine host syncs in the reference, one pass over 8 B per transition here.
// Raw fp64 moments are emitted (not means) so that env shards on several GPUs
// combine with plain SUM / MIN / MAX all-reduces.
//
// K5 restates src/rl8/_utils.py:211-225 (Batcher): index-gather of every buffer
// leaf for one minibatch, all leaves in one launch.
#include "common.hip.h"

namespace rl8 {

// partial columns: 0 sum(ret) 1 sum(ret^2) 2 min(ret) 3 max(ret)
//                  4 sum(r)   5 sum(r^2)   6 min(r)   7 max(r)
//                  8 sum(rdr) 9 sum(rdr^2)

// VEC == 4: time-major leaves (env_stride == 1): one lane walks 4 adjacent envs
// with 16-byte loads per column.  VEC == 1: any strides, one env per lane.
template <int VEC>
__global__ __launch_bounds__(kBlock) void rollout_stats_kernel(
    const float *__restrict__ rewards, const float *__restrict__ rdr, int64_t n, int64_t h,
    int64_t env_stride, int64_t time_stride, double *__restrict__ partials,
    double *__restrict__ out) {
  __shared__ double smem[6 * kWavesPerBlock];
  double sums[6] = {0, 0, 0, 0, 0, 0};  // ret, ret^2, r, r^2, rdr, rdr^2
  double mins[2] = {INFINITY, INFINITY}, maxs[2] = {-INFINITY, -INFINITY};
  const int64_t stride = (int64_t)gridDim.x * kBlock * VEC;
  for (int64_t e = ((int64_t)blockIdx.x * kBlock + threadIdx.x) * VEC; e < n; e += stride) {
    float ret[VEC];  // torch.sum(rewards[:, :-1], dim=1) accumulates in f32
#pragma unroll
    for (int i = 0; i < VEC; ++i) ret[i] = 0.0f;
    float rmin = INFINITY, rmax = -INFINITY;
#pragma unroll 4
    for (int64_t t = 0; t < h; ++t) {
      float r[VEC];
      if constexpr (VEC == 4) {
        const float4 q = *reinterpret_cast<const float4 *>(rewards + t * time_stride + e);
        r[0] = q.x; r[1] = q.y; r[2] = q.z; r[3] = q.w;
      } else {
        r[0] = rewards[e * env_stride + t * time_stride];
      }
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        ret[i] = ret[i] + r[i];
        sums[2] += (double)r[i];
        sums[3] += (double)r[i] * (double)r[i];
        rmin = fminf(rmin, r[i]);
        rmax = fmaxf(rmax, r[i]);
      }
      if (rdr) {
        float d[VEC];
        if constexpr (VEC == 4) {
          const float4 q = *reinterpret_cast<const float4 *>(rdr + (t + 1) * time_stride + e);
          d[0] = q.x; d[1] = q.y; d[2] = q.z; d[3] = q.w;
        } else {
          d[0] = rdr[e * env_stride + (t + 1) * time_stride];
        }
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
          sums[4] += (double)d[i];
          sums[5] += (double)d[i] * (double)d[i];
        }
      }
    }
    mins[1] = (double)rmin < mins[1] ? (double)rmin : mins[1];
    maxs[1] = (double)rmax > maxs[1] ? (double)rmax : maxs[1];
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      sums[0] += (double)ret[i];
      sums[1] += (double)ret[i] * (double)ret[i];
      mins[0] = ret[i] < mins[0] ? (double)ret[i] : mins[0];
      maxs[0] = ret[i] > maxs[0] ? (double)ret[i] : maxs[0];
    }
  }
  double a4[4] = {sums[0], sums[1], sums[2], sums[3]};
  block_reduce<4, SumOp>(a4, smem);
  double b2[2] = {sums[4], sums[5]};
  block_reduce<2, SumOp>(b2, smem);
  block_reduce<2, MinOp>(mins, smem);
  block_reduce<2, MaxOp>(maxs, smem);
  if (threadIdx.x == 0) {
    double *row = partials + (int64_t)blockIdx.x * kPartialWidth;
    const double vals[10] = {a4[0], a4[1], mins[0], maxs[0], a4[2], a4[3], mins[1], maxs[1],
                             b2[0], b2[1]};
#pragma unroll
    for (int c = 0; c < 10; ++c) publish_partial(row + c, vals[c]);
  }
  if (!last_block_arrives(ticket_word(partials))) return;
  // Last block: combine every row in order (fixed order => reproducible).
  for (int i = 0; i < 6; ++i) sums[i] = 0.0;
  mins[0] = mins[1] = INFINITY;
  maxs[0] = maxs[1] = -INFINITY;
  const int rows = (int)gridDim.x;
  const double nn = (double)n, nh = (double)n * (double)h;
  for (int r = threadIdx.x; r < rows; r += kBlock) {
    double row[10];
#pragma unroll
    for (int c = 0; c < 10; ++c) row[c] = read_partial(partials + (int64_t)r * kPartialWidth + c);
    sums[0] += row[0]; sums[1] += row[1]; sums[2] += row[4];
    sums[3] += row[5]; sums[4] += row[8]; sums[5] += row[9];
    mins[0] = row[2] < mins[0] ? row[2] : mins[0];
    maxs[0] = row[3] > maxs[0] ? row[3] : maxs[0];
    mins[1] = row[6] < mins[1] ? row[6] : mins[1];
    maxs[1] = row[7] > maxs[1] ? row[7] : maxs[1];
  }
  block_reduce<6, SumOp>(sums, smem);
  block_reduce<2, MinOp>(mins, smem);
  block_reduce<2, MaxOp>(maxs, smem);
  if (threadIdx.x == 0) {
    out[0] = nn;  out[1] = sums[0]; out[2] = sums[1]; out[3] = mins[0]; out[4] = maxs[0];
    out[5] = nh;  out[6] = sums[2]; out[7] = sums[3]; out[8] = mins[1]; out[9] = maxs[1];
    out[10] = sums[4]; out[11] = sums[5];
    *ticket_word(partials) = 0u;
  }
}

struct GatherArgs {
  rl8_gather_field f[RL8_MAX_GATHER_FIELDS];
  int n_fields;
};

// One lane per (sample, field-element).  Reads are random 4/8-byte accesses
// (this is a shuffle), writes are dense.
__global__ __launch_bounds__(kBlock) void gather_minibatch_kernel(
    const int64_t *__restrict__ index, int64_t m, int64_t h, GatherArgs args) {
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < m; i += stride) {
    const int64_t s = index[i];
    const int64_t env = s / h, t = s - env * h;
#pragma unroll 1
    for (int f = 0; f < args.n_fields; ++f) {
      const rl8_gather_field &fd = args.f[f];
      const int64_t src0 = env * fd.env_stride + t * fd.time_stride;
      if (fd.elem_bytes == 4) {
        const uint32_t *src = static_cast<const uint32_t *>(fd.src) + src0;
        uint32_t *dst = static_cast<uint32_t *>(fd.dst) + i * fd.row_elems;
        for (int c = 0; c < fd.row_elems; ++c) dst[c] = src[c];
      } else {
        const uint64_t *src = static_cast<const uint64_t *>(fd.src) + src0;
        uint64_t *dst = static_cast<uint64_t *>(fd.dst) + i * fd.row_elems;
        for (int c = 0; c < fd.row_elems; ++c) dst[c] = src[c];
      }
    }
  }
}

// Wide rows (recurrent states: 1 KiB per layer): one wave per gathered row, 16 B
// per lane, so each row moves as whole 1-KiB wave instructions.
__global__ __launch_bounds__(kBlock) void gather_wide_rows_kernel(
    const int64_t *__restrict__ index, int64_t m, int64_t h, rl8_gather_field fd) {
  const int lane = threadIdx.x & (kWave - 1);
  const int64_t wave0 = ((int64_t)blockIdx.x * kBlock + threadIdx.x) / kWave;
  const int64_t waves = (int64_t)gridDim.x * kWavesPerBlock;
  const int vecs = (int)((int64_t)fd.row_elems * fd.elem_bytes / 16);
  for (int64_t i = wave0; i < m; i += waves) {
    const int64_t s = index[i];
    const int64_t env = s / h, t = s - env * h;
    const char *src = static_cast<const char *>(fd.src) +
                      (env * fd.env_stride + t * fd.time_stride) * fd.elem_bytes;
    char *dst = static_cast<char *>(fd.dst) + i * (int64_t)fd.row_elems * fd.elem_bytes;
    for (int c = lane; c < vecs; c += kWave)
      reinterpret_cast<float4 *>(dst)[c] = reinterpret_cast<const float4 *>(src)[c];
  }
}

}  // namespace rl8

using namespace rl8;

RL8_API int rl8_abi_version(char *arch, int arch_len) {
  if (arch && arch_len > 0) {
    const char *a = "gfx950";
    int i = 0;
    for (; a[i] && i < arch_len - 1; ++i) arch[i] = a[i];
    arch[i] = 0;
  }
  return 100;
}

RL8_API int64_t rl8_scratch_bytes(void) {
  return (int64_t)(RL8_MAX_PARTIALS + 8) * kPartialWidth * (int64_t)sizeof(double);
}

RL8_API int rl8_rollout_stats_f32(const float *rewards, const float *rdr, int64_t n, int64_t h,
                                  int64_t env_stride, int64_t time_stride, double *stats_out,
                                  void *scratch, void *stream) {
  if (!rewards || !stats_out || !scratch) return RL8_ENULL;
  if (n <= 0 || h <= 0 || env_stride <= 0 || time_stride <= 0) return RL8_ESIZE;
  hipStream_t s = (hipStream_t)stream;
  double *partials = (double *)scratch;
  const bool vec = env_stride == 1 && n % 4 == 0 && time_stride % 4 == 0 && aligned16(rewards) &&
                   (!rdr || aligned16(rdr));
  static const int cap = env_int("RL8_STATS_GRID_CAP");
  if (vec)
    rollout_stats_kernel<4><<<grid_for(n, kBlock * 4, cap > 0 ? cap : kMaxGrid), kBlock, 0, s>>>(
        rewards, rdr, n, h, env_stride, time_stride, partials, stats_out);
  else
    rollout_stats_kernel<1><<<grid_for(n, kBlock), kBlock, 0, s>>>(
        rewards, rdr, n, h, env_stride, time_stride, partials, stats_out);
  return launch_status();
}

RL8_API int rl8_gather_minibatch(const int64_t *index, int64_t m, int64_t h,
                                 const rl8_gather_field *fields, int n_fields, void *stream) {
  if (!index || !fields) return RL8_ENULL;
  if (m <= 0 || h <= 0 || n_fields <= 0 || n_fields > RL8_MAX_GATHER_FIELDS) return RL8_ESIZE;
  GatherArgs args;
  args.n_fields = 0;
  hipStream_t s = (hipStream_t)stream;
  for (int f = 0; f < n_fields; ++f) {
    const rl8_gather_field &fd = fields[f];
    if (!fd.src || !fd.dst) return RL8_ENULL;
    if (fd.elem_bytes != 4 && fd.elem_bytes != 8) return RL8_ECONFIG;
    if (fd.row_elems <= 0) return RL8_ESIZE;
    const int64_t row_bytes = (int64_t)fd.row_elems * fd.elem_bytes;
    const bool wide = row_bytes >= 256 && row_bytes % 16 == 0 && aligned16(fd.src) &&
                      aligned16(fd.dst) && (fd.env_stride * fd.elem_bytes) % 16 == 0 &&
                      (fd.time_stride * fd.elem_bytes) % 16 == 0;
    if (wide) {
      gather_wide_rows_kernel<<<grid_for(m, kWavesPerBlock), kBlock, 0, s>>>(index, m, h, fd);
      const int st = launch_status();
      if (st != RL8_OK) return st;
    } else {
      args.f[args.n_fields++] = fd;
    }
  }
  if (args.n_fields > 0)
    gather_minibatch_kernel<<<grid_for(m, kBlock), kBlock, 0, s>>>(index, m, h, args);
  return launch_status();
}
