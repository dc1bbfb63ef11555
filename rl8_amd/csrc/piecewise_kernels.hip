// OPT-IN, special to scalar observations (d_in = 1: the dummy envs of BASELINE configs 2 and 4) -- VERDICT r3 item 10,
// DESIGN.md section 9.  The general tower kernels (mlp_rows / mlp_f16 / mlp_split / mlp_kernels) stay what every other shape
// runs and what this path is checked against.
//
// A two-layer ReLU tower of one scalar, out(x) = W3 relu(W2 relu(w1 x + b1) + b2) + b3, is piecewise linear in x: 256
// layer-1 kinks at x = -b1_i / w1_i, and inside each segment between them the layer-2 pre-activations are A_jk x + B_jk,
// with their own sign changes.  All breakpoints sorted, every interval has ONE gate pattern for both layers, hence one
// slope and one value per output: a table of P + 1 intervals (P a few hundred in practice, 66 048 at most), rebuilt in
// fp64 whenever the weights change (rl8_amd/nn/piecewise_mlp.py).  Then
//   forward:  a search of x in the breakpoints + one multiply-add per output              (12-16 bytes per row)
//   backward: every parameter gradient is a small fp64 product of the per-interval gate patterns with the per-interval
//             sums  S0_p = sum of dOut over the rows of interval p,  S1_p = sum of dOut * x  -- two numbers per output
//             and interval, formed here in ONE pass over (x, dOut)                         (8-12 bytes per row)
// instead of 3 x 131 072 executed FLOP per row, forward alone.  The sums are accumulated EXACTLY: every term is split into
// two 64-bit integers on a common power-of-two scale (76 bits below the largest |dOut| * max(|x|, 1) of the call) and added
// with integer atomics, first in LDS, then into the global table -- integer addition commutes, so the result does not
// depend on the order the rows arrive in (bitwise reproducible, no sort, no fixed reduction tree).
#include "common.hip.h"

namespace rl8 {

constexpr int kPwMaxBreaks = 2048;   // table resident in LDS: (P + 1) * (2 + 2 n_out) floats <= 64 KiB at n_out = 3
constexpr int kPwMaxOut = 3;

// number of breakpoints strictly below v (torch.searchsorted, right = False): interval index of v
__device__ __forceinline__ int pw_interval(const float *breaks, int p, int top /* power of two >= p */, float v) {
  int lo = 0;
  for (int step = top; step > 0; step >>= 1) {
    const int probe = lo + step;
    if (probe <= p && breaks[probe - 1] < v) lo = probe;
  }
  return lo;
}

template <int NOUT>
__global__ __launch_bounds__(kBlock) void pw_forward_kernel(const float *__restrict__ x, int64_t m,
                                                            const float *__restrict__ table, int p, int top,
                                                            float *__restrict__ out) {
  extern __shared__ float lds[];
  const int floats = p + (p + 1) * (1 + 2 * NOUT);
  for (int i = threadIdx.x; i < floats; i += kBlock) lds[i] = table[i];
  __syncthreads();
  const float *breaks = lds, *anchor = lds + p, *value = anchor + (p + 1), *slope = value + (p + 1) * NOUT;
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < m; i += stride) {
    const float v = x[i];
    const int at = pw_interval(breaks, p, top, v);
    const float d = v - anchor[at];
#pragma unroll
    for (int q = 0; q < NOUT; ++q) out[i * NOUT + q] = value[at * NOUT + q] + slope[at * NOUT + q] * d;
  }
}

// max |x| and max |dOut| of the call (bit patterns of non-negative floats, atomicMax on words the caller zeroed);
// a non-finite dOut or x raises bounds[2].
__global__ __launch_bounds__(kBlock) void pw_bounds_kernel(const float *__restrict__ x, const float *__restrict__ dout,
                                                           int64_t m, int n_out, uint32_t *__restrict__ bounds) {
  float mx = 0.0f, md = 0.0f;
  bool bad = false;
  const int64_t tid = (int64_t)blockIdx.x * kBlock + threadIdx.x, threads = (int64_t)gridDim.x * kBlock;
  // both arrays as flat streams of 16-byte vectors (x: m floats, dOut: m * n_out), four loads in flight per lane
  auto scan = [&](const float *p, int64_t n, float &top) {
    const int64_t vecs = ((uintptr_t)p & 15) == 0 ? n / 4 : 0;
    int64_t q = tid;
    for (; q + 3 * threads < vecs; q += 4 * threads) {
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = reinterpret_cast<const float4 *>(p)[q + u * threads];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float a = fmaxf(fmaxf(fabsf(v[u].x), fabsf(v[u].y)), fmaxf(fabsf(v[u].z), fabsf(v[u].w)));
        // (fmaxf drops a NaN operand: test the sum, which keeps it)
        bad |= !(fabsf(v[u].x) + fabsf(v[u].y) + fabsf(v[u].z) + fabsf(v[u].w) < INFINITY);
        top = fmaxf(top, a);
      }
    }
    for (; q < vecs; q += threads) {
      const float4 v = reinterpret_cast<const float4 *>(p)[q];
      bad |= !(fabsf(v.x) + fabsf(v.y) + fabsf(v.z) + fabsf(v.w) < INFINITY);
      top = fmaxf(top, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
    }
    for (int64_t i = 4 * vecs + tid; i < n; i += threads) {
      const float a = fabsf(p[i]);
      bad |= !(a < INFINITY);
      top = fmaxf(top, a);
    }
  };
  scan(x, m, mx);
  scan(dout, m * n_out, md);
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) {
    mx = fmaxf(mx, __shfl_down(mx, off, kWave));
    md = fmaxf(md, __shfl_down(md, off, kWave));
  }
  if ((threadIdx.x & (kWave - 1)) == 0) {
    atomicMax(bounds + 0, __float_as_uint(mx));
    atomicMax(bounds + 1, __float_as_uint(md));
  }
  if (__any(bad) && (threadIdx.x & (kWave - 1)) == 0) atomicOr(bounds + 2, 1u);
}

// The common scale of a call's terms d and d * x: all of them below 2^e.
__device__ __forceinline__ int pw_scale_exponent(const uint32_t *bounds) {
  const float mx = fmaxf(__uint_as_float(bounds[0]), 1.0f), md = __uint_as_float(bounds[1]);
  int e = __builtin_amdgcn_frexp_expf(mx) + __builtin_amdgcn_frexp_expf(md);  // mx < 2^ex, md < 2^ed
  return md > 0.0f ? (e < -900 ? -900 : e) : 0;
}
constexpr int kPwHiBits = 36, kPwLoBits = 38;  // |term| 2^(36 - e) < 2^36; 2^25 rows of them < 2^61.  Residual: 38 more bits

// S[p][kind][q] += (kind ? dOut[r][q] * x[r] : dOut[r][q]) over the rows r of interval p, exactly: hi = rint(v 2^(36 - e)),
// lo = rint((v - hi 2^(e - 36)) 2^(36 - e + 38)) as int64, integer atomics in LDS, then into `acc` (zeroed by the caller).
template <int NOUT>
__global__ __launch_bounds__(kBlock) void pw_segment_sums_kernel(const float *__restrict__ x, const float *__restrict__ dout,
                                                                 int64_t m, const float *__restrict__ breaks_in, int p, int top,
                                                                 const uint32_t *__restrict__ bounds,
                                                                 unsigned long long *__restrict__ acc) {
  extern __shared__ unsigned long long lacc[];  // [(p + 1)][kind][q][hi | lo], then the breakpoints
  const int slots = (p + 1) * 2 * NOUT * 2;
  float *breaks = reinterpret_cast<float *>(lacc + slots);
  for (int i = threadIdx.x; i < slots; i += kBlock) lacc[i] = 0ull;
  for (int i = threadIdx.x; i < p; i += kBlock) breaks[i] = breaks_in[i];
  __syncthreads();
  const int e = pw_scale_exponent(bounds);
  const double up = ldexp(1.0, kPwHiBits - e), down = ldexp(1.0, e - kPwHiBits), up_lo = ldexp(1.0, kPwHiBits - e + kPwLoBits);
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  auto add_row = [&](float xv, int at, const float *d) {
    unsigned long long *row = lacc + (int64_t)at * (2 * NOUT * 2);
#pragma unroll
    for (int q = 0; q < NOUT; ++q) {
      const double dq = (double)d[q];
#pragma unroll
      for (int kind = 0; kind < 2; ++kind) {
        const double v = kind ? dq * (double)xv : dq;  // (exact: 24 x 24 bits)
        if (v != 0.0) {
          const long long hi = __double2ll_rn(v * up);
          const long long lo = __double2ll_rn((v - (double)hi * down) * up_lo);
          atomicAdd(row + (kind * NOUT + q) * 2, (unsigned long long)hi);
          if (lo) atomicAdd(row + (kind * NOUT + q) * 2 + 1, (unsigned long long)lo);
        }
      }
    }
  };
  // four rows per lane and trip: four independent searches in flight (the search is a chain of dependent LDS reads)
  const int64_t quads = m / 4;
  for (int64_t g = (int64_t)blockIdx.x * kBlock + threadIdx.x; g < quads; g += stride) {
    float xv[4], dv[4][NOUT];
    int at[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      xv[u] = x[4 * g + u];
#pragma unroll
      for (int q = 0; q < NOUT; ++q) dv[u][q] = dout[(4 * g + u) * NOUT + q];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) at[u] = 0;
    for (int step = top; step > 0; step >>= 1) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int probe = at[u] + step;
        if (probe <= p && breaks[probe - 1] < xv[u]) at[u] = probe;
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) add_row(xv[u], at[u], dv[u]);
  }
  for (int64_t i = 4 * quads + (int64_t)blockIdx.x * kBlock + threadIdx.x; i < m; i += stride) {
    float dv[NOUT];
#pragma unroll
    for (int q = 0; q < NOUT; ++q) dv[q] = dout[i * NOUT + q];
    add_row(x[i], pw_interval(breaks, p, top, x[i]), dv);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < slots; i += kBlock)
    if (lacc[i]) atomicAdd(acc + i, lacc[i]);
}

// (hi, lo) integer pairs -> doubles; NaN throughout if the call saw a non-finite input
__global__ __launch_bounds__(kBlock) void pw_sums_finish_kernel(const unsigned long long *__restrict__ acc, int count,
                                                                const uint32_t *__restrict__ bounds, double *__restrict__ out) {
  const int e = pw_scale_exponent(bounds);
  const double down = ldexp(1.0, e - kPwHiBits), down_lo = ldexp(1.0, e - kPwHiBits - kPwLoBits);
  const bool bad = bounds[2] != 0u;
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < count; i += gridDim.x * kBlock) {
    const double v = (double)(long long)acc[2 * i] * down + (double)(long long)acc[2 * i + 1] * down_lo;
    out[i] = bad ? __longlong_as_double(0x7ff8000000000000ll) : v;
  }
}

static int pw_top(int p) {
  int top = 1;
  while (top < p) top <<= 1;
  return p > 0 ? top : 0;
}

}  // namespace rl8

using namespace rl8;

RL8_API int rl8_pw_max_breaks(void) { return kPwMaxBreaks; }

RL8_API int64_t rl8_pw_workspace_bytes(int p, int n_out) {
  // the integer accumulators [(p + 1)][2][n_out][hi | lo] + 64 bytes of bounds
  return (int64_t)(p + 1) * 2 * n_out * 2 * 8 + 64;
}

RL8_API int rl8_pw_tower_forward_f32(const float *x, int64_t m, const float *table, int p, int n_out, float *out,
                                     void *stream) {
  if (!x || !table || !out) return RL8_ENULL;
  if (m <= 0 || p < 0 || p > kPwMaxBreaks || n_out < 1 || n_out > kPwMaxOut) return RL8_ESIZE;
  const int lds = (p + (p + 1) * (1 + 2 * n_out)) * 4;
  const int grid = grid_for(m, kBlock * 4, 8 * kCUs);
  hipStream_t s = (hipStream_t)stream;
  static LdsOptIn opt[kPwMaxOut];
#define RL8_PW_FWD(N) \
  if (n_out == N) { \
    if (const int e = allow_dynamic_lds(opt[N - 1], reinterpret_cast<const void *>(&pw_forward_kernel<N>), 160 * 1024)) return e; \
    pw_forward_kernel<N><<<grid, kBlock, lds, s>>>(x, m, table, p, pw_top(p), out); \
  }
  RL8_PW_FWD(1) RL8_PW_FWD(2) RL8_PW_FWD(3)
#undef RL8_PW_FWD
  return launch_status();
}

RL8_API int rl8_pw_segment_sums_f32(const float *x, const float *dout, int64_t m, int n_out, const float *breaks, int p,
                                    void *workspace, double *sums_out, void *stream) {
  if (!x || !dout || !workspace || !sums_out || (p > 0 && !breaks)) return RL8_ENULL;
  if (m <= 0 || m > ((int64_t)1 << 25) || p < 0 || p > kPwMaxBreaks || n_out < 1 || n_out > kPwMaxOut) return RL8_ESIZE;
  if (((uintptr_t)workspace & 15) != 0) return RL8_EALIGN;
  hipStream_t s = (hipStream_t)stream;
  const int slots = (p + 1) * 2 * n_out * 2;
  unsigned long long *acc = static_cast<unsigned long long *>(workspace);
  uint32_t *bounds = reinterpret_cast<uint32_t *>(acc + slots);
  if (const hipError_t e = hipMemsetAsync(workspace, 0, (size_t)slots * 8 + 64, s); e != hipSuccess) return (int)e;
  pw_bounds_kernel<<<grid_for(m, kBlock * 4, 4 * kCUs), kBlock, 0, s>>>(x, dout, m, n_out, bounds);
  const int lds = slots * 8 + p * 4;
  if (lds > 150 * 1024) return RL8_ESIZE;  // (the accumulators live in LDS: the caller keeps the general kernels)
  const int per_cu = (160 * 1024) / (lds + 1024) < 1 ? 1 : (160 * 1024) / (lds + 1024);
  const int64_t want = (m + kBlock * 8 - 1) / (kBlock * 8);
  const int cap = (per_cu > 8 ? 8 : per_cu) * kCUs;
  const int grid = (int)(want < 1 ? 1 : want > cap ? cap : want);
  static LdsOptIn opt[kPwMaxOut];
#define RL8_PW_SUMS(N) \
  if (n_out == N) { \
    if (const int e = allow_dynamic_lds(opt[N - 1], reinterpret_cast<const void *>(&pw_segment_sums_kernel<N>), 160 * 1024)) return e; \
    pw_segment_sums_kernel<N><<<grid, kBlock, lds, s>>>(x, dout, m, breaks, p, pw_top(p), bounds, acc); \
  }
  RL8_PW_SUMS(1) RL8_PW_SUMS(2) RL8_PW_SUMS(3)
#undef RL8_PW_SUMS
  pw_sums_finish_kernel<<<(slots / 2 + kBlock - 1) / kBlock, kBlock, 0, s>>>(acc, slots / 2, bounds, sums_out);
  return launch_status();
}
