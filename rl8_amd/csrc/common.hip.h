// Shared device helpers for the gfx950 PPO kernels: 64-lane wave reductions,
// deterministic two-stage block reductions in fp64, launch-shape helpers.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <atomic>

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

// Dynamic-LDS opt-in of a kernel on the device that is current NOW: once per device and kernel (the attribute belongs to
// the device's copy of the function), safe to race from several host threads, and the error comes back instead of
// being dropped (a process-wide `static bool` set the attribute on the first device only and launched on
// the others without it).  One `static LdsOptIn` per call site and kernel.
struct LdsOptIn {
  std::atomic<unsigned long long> devices{0};
};
inline int allow_dynamic_lds(LdsOptIn &state, const void *kernel, int bytes) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return (int)e;
  const unsigned long long bit = 1ull << (dev & 63);
  if (state.devices.load(std::memory_order_acquire) & bit) return 0;
  // (a kernel's static LDS counts against the CU's 160 KiB too: ask for what is left at most)
  hipFuncAttributes attr;
  if (hipFuncGetAttributes(&attr, kernel) == hipSuccess) {
    const int left = 160 * 1024 - (int)attr.sharedSizeBytes;
    if (bytes > left) bytes = left;
  } else {
    (void)hipGetLastError();
  }
  e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return (int)e;
  }
  state.devices.fetch_or(bit, std::memory_order_release);
  return 0;
}

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

// The arrival count is kept on two levels: 32 consecutive blocks share a group
// word, and only the last arriver of a group takes a ticket on the top word.
// One word for the whole grid serialises every block's atomic on one address
// at the memory side: 11.6 ns each, 13.6 us for 1024 blocks against 0.6 us here
// (tools/probes/ticket_probe.hip, profiles/r03_experiments.md).  Group words
// sit 4 KiB apart so that different groups' atomics land on different channels.
constexpr int kTicketGroup = 32;
constexpr int kTicketWordPitch = 1024;  // unsigneds between group words (4 KiB)

__device__ __forceinline__ unsigned *ticket_word(double *scratch) {
  return reinterpret_cast<unsigned *>(scratch + (int64_t)RL8_MAX_PARTIALS * 16);
}

// Call from every thread after thread 0 has published the block's row.  True in
// every thread of the one block that arrived after all others; that block
// zeroes the top word when it is done (`*ticket_word(scratch) = 0u`), the group
// words are zeroed here by their last arrivers.
__device__ __forceinline__ bool last_block_arrives(unsigned *top) {
  __shared__ int is_last;
  if (threadIdx.x == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned group = blockIdx.x / kTicketGroup;
    const unsigned groups = (gridDim.x + kTicketGroup - 1) / kTicketGroup;
    const unsigned members =
        group == groups - 1 ? gridDim.x - group * kTicketGroup : (unsigned)kTicketGroup;
    unsigned *word = top + (int64_t)(1 + group) * kTicketWordPitch;
    int last = 0;
    if (__hip_atomic_fetch_add(word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ==
        members - 1) {
      __hip_atomic_store(word, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      last = __hip_atomic_fetch_add(top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ==
             groups - 1;
    }
    is_last = last;
    if (last) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
  __syncthreads();
  return is_last != 0;
}

// The last block's walk over the published rows: thread t takes rows t,
// t + blockDim, ... in that order (the order the sums have always been taken
// in), with the loads of kFoldRows rows in the air at once instead of one
// row's (6.8 us per 1024 rows of 10 columns otherwise: latency, not bytes).
constexpr int kFoldRows = 4;

template <int NC, class F>
__device__ __forceinline__ void fold_partial_rows(const double *partials, int rows, F &&f) {
  const int step = (int)blockDim.x;
  for (int r0 = threadIdx.x; r0 < rows; r0 += step * kFoldRows) {
    double v[kFoldRows][NC];
#pragma unroll
    for (int u = 0; u < kFoldRows; ++u) {
      const int r = r0 + u * step < rows ? r0 + u * step : r0;
#pragma unroll
      for (int c = 0; c < NC; ++c) v[u][c] = read_partial(partials + (int64_t)r * kPartialWidth + c);
    }
#pragma unroll
    for (int u = 0; u < kFoldRows; ++u)
      if (r0 + u * step < rows) f(v[u]);
  }
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
