// Shared device helpers for the gfx950 PPO kernels: 64-lane wave reductions,
// deterministic two-stage block reductions in fp64, launch-shape helpers.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

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

// Reduces NV values per thread across a block of up to 256 threads (whole
// waves).  Result valid in thread 0.  `smem` must hold NV * kWavesPerBlock
// doubles.  Fixed order => bitwise reproducible.
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
    const int waves = (blockDim.x + kWave - 1) / kWave;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      double acc = smem[i * kWavesPerBlock];
      for (int w = 1; w < waves; ++w) acc = Op::apply(acc, smem[i * kWavesPerBlock + w]);
      v[i] = acc;
    }
  }
  __syncthreads();
}

// ---- single-launch reductions ------------------------------------------------
// Every block publishes one row of partials; the block that arrives last (a
// ticket from one agent-scope atomic) reduces all rows in row order, so the
// result is bitwise reproducible and no second launch is needed.
//
// Hand-off, per the gfx950 rules for inter-workgroup visibility (per-CU L1s are
// never refreshed by other CUs' stores, per-XCD L2s are not coherent with each
// other): the row is written with agent-scope (sc1, write-through) stores by
// ONE lane, that lane drains them (s_waitcnt vmcnt(0)) and only then takes its
// ticket; the last arriver issues an agent-scope acquire, the block meets at a
// barrier, and the rows are read with agent-scope (sc1) loads.  No release fence
// is needed because every handed-off byte is an sc1 store -- a release would
// write back the whole XCD L2, i.e. every streaming output line the other
// blocks have in flight.
// The ticket word lives in the caller's scratch, must be zero before the first
// launch (rl8_scratch_bytes) and is re-zeroed by the last block.
__device__ __forceinline__ void publish_partial(double *slot, double v) {
  __hip_atomic_store(slot, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ double read_partial(const double *slot) {
  return __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Call from every thread after thread 0 has published the block's row.
__device__ __forceinline__ bool last_block_arrives(unsigned *ticket_word) {
  __shared__ int is_last;
  if (threadIdx.x == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned ticket =
        __hip_atomic_fetch_add(ticket_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    is_last = ticket == gridDim.x - 1;
    if (is_last) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
  __syncthreads();
  return is_last != 0;
}

__device__ __forceinline__ unsigned *ticket_word(double *scratch) {
  return reinterpret_cast<unsigned *>(scratch + (int64_t)RL8_MAX_PARTIALS * 16);
}

inline int grid_for(int64_t work_items, int items_per_block, int cap = kMaxGrid) {
  int64_t g = (work_items + items_per_block - 1) / items_per_block;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (int)g;
}

// Tuning knob: integer environment variable read once (0 / unset = default).
inline int env_int(const char *name) {
  const char *v = getenv(name);
  return v ? atoi(v) : 0;
}

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

inline int launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? RL8_OK : (int)e;
}

}  // namespace rl8
