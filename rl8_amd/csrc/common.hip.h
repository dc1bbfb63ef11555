// Shared device helpers for the gfx950 PPO kernels: 64-lane wave reductions,
// deterministic two-stage block reductions in fp64, launch-shape helpers.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rl8_amd.h"

#define RL8_API extern "C" __attribute__((visibility("default")))

namespace rl8 {

constexpr int kWave = 64;        // CDNA wavefront
constexpr int kBlock = 256;      // 4 waves: one per SIMD of a CU
constexpr int kWavesPerBlock = kBlock / kWave;
constexpr int kCUs = 256;        // MI355X
constexpr int kMaxGrid = 2048;   // 8 blocks/CU: grid-stride beyond this
constexpr int kPartialWidth = 16;  // doubles per partial row in scratch

static_assert(kMaxGrid <= RL8_MAX_PARTIALS, "partials must fit the scratch");

struct SumOp {
  __device__ __forceinline__ static double apply(double a, double b) { return a + b; }
};
struct MinOp {
  __device__ __forceinline__ static double apply(double a, double b) { return a < b ? a : b; }
};
struct MaxOp {
  __device__ __forceinline__ static double apply(double a, double b) { return a > b ? a : b; }
};

template <class Op>
__device__ __forceinline__ double wave_reduce(double v) {
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) v = Op::apply(v, __shfl_down(v, off, kWave));
  return v;
}

// Reduces NV values per thread across a 256-thread block.  Result valid in
// thread 0.  `smem` must hold NV * kWavesPerBlock doubles.  Fixed order =>
// bitwise reproducible.
template <int NV, class Op>
__device__ __forceinline__ void block_reduce(double (&v)[NV], double *smem) {
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x / kWave;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    v[i] = wave_reduce<Op>(v[i]);
    if (lane == 0) smem[i * kWavesPerBlock + wave] = v[i];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      double acc = smem[i * kWavesPerBlock];
#pragma unroll
      for (int w = 1; w < kWavesPerBlock; ++w) acc = Op::apply(acc, smem[i * kWavesPerBlock + w]);
      v[i] = acc;
    }
  }
  __syncthreads();
}

inline int grid_for(int64_t work_items, int items_per_block) {
  int64_t g = (work_items + items_per_block - 1) / items_per_block;
  if (g < 1) g = 1;
  if (g > kMaxGrid) g = kMaxGrid;
  return (int)g;
}

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

inline int launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? RL8_OK : (int)e;
}

}  // namespace rl8
