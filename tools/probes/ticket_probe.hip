// What a single-launch reduction's hand-off costs on MI355X (common.hip.h: every block publishes a row of partials with
// sc1 stores, drains them, takes a ticket from ONE agent-scope atomic; the last arriver folds the rows).  The rollout
// statistics kernel spends 31 us on an 8 MB input (h = 1) and 1.3 us per further 8 MB: the fixed part is this hand-off.
// Variants, G blocks of 256 threads each, no other work:
//   flat       one ticket word for all G blocks (what common.hip.h does)
//   two_level  G / 32 group words 4 KiB apart, the last arriver of a group takes a ticket on the top word
//   no_ticket  publish only (the floor: launch + stores)
// and the fold of G rows x 10 doubles by the last block, timed with the ticket (fold = 1) or without.
//   hipcc --offload-arch=gfx950 -O3 -o ticket_probe ticket_probe.hip && ./ticket_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

constexpr int kWidth = 16;

__device__ inline void publish(double *row, int n) {
  for (int c = 0; c < n; ++c) __hip_atomic_store(row + c, (double)c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

__device__ inline void fold(const double *rows, int g, double *out) {
  double acc = 0.0;
  for (int r = threadIdx.x; r < g; r += blockDim.x)
    for (int c = 0; c < 10; ++c)
      acc += __hip_atomic_load(rows + (int64_t)r * kWidth + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if ((threadIdx.x & 63) == 0) atomicAdd(out, acc);
}

template <int MODE, bool FOLD>
__global__ __launch_bounds__(256) void handoff(double *rows, unsigned *words, double *out) {
  __shared__ int last;
  if (threadIdx.x == 0) {
    publish(rows + (int64_t)blockIdx.x * kWidth, 10);
    last = 0;
    if (MODE == 0) {
      const unsigned t = __hip_atomic_fetch_add(words, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      last = t == gridDim.x - 1;
      if (last) *words = 0u;
    } else if (MODE == 1) {
      const unsigned group = blockIdx.x >> 5, groups = (gridDim.x + 31) >> 5;
      const unsigned members = group == groups - 1 ? gridDim.x - (group << 5) : 32u;
      unsigned *gw = words + 1024 + group * 1024;  // 4 KiB apart
      const unsigned t = __hip_atomic_fetch_add(gw, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (t == members - 1) {
        *gw = 0u;
        const unsigned u = __hip_atomic_fetch_add(words, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = u == groups - 1;
        if (last) *words = 0u;
      }
    }
    if (last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
  if (FOLD && last) fold(rows, gridDim.x, out);
}

template <int MODE, bool FOLD>
static float run(int g, double *rows, unsigned *words, double *out) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  for (int i = 0; i < 5; ++i) handoff<MODE, FOLD><<<g, 256>>>(rows, words, out);
  hipEventRecord(a);
  for (int i = 0; i < 50; ++i) handoff<MODE, FOLD><<<g, 256>>>(rows, words, out);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0.f;
  hipEventElapsedTime(&ms, a, b);
  return ms * 1000.f / 50.f;
}

int main() {
  double *rows, *out;
  unsigned *words;
  hipMalloc(&rows, (size_t)8192 * kWidth * sizeof(double));
  hipMalloc(&out, 64);
  hipMalloc(&words, (size_t)(1024 + 300 * 1024) * sizeof(unsigned));
  hipMemset(words, 0, (size_t)(1024 + 300 * 1024) * sizeof(unsigned));
  hipMemset(out, 0, 64);
  printf("%6s %12s %12s %12s %12s %12s  (us per launch, back to back)\n", "blocks", "no_ticket", "flat", "flat+fold",
         "two_level", "two_lvl+fold");
  for (int g : {64, 256, 512, 1024, 2048, 4096, 8192}) {
    printf("%6d %12.2f %12.2f %12.2f %12.2f %12.2f\n", g, run<2, false>(g, rows, words, out),
           run<0, false>(g, rows, words, out), run<0, true>(g, rows, words, out), run<1, false>(g, rows, words, out),
           run<1, true>(g, rows, words, out));
  }
  return 0;
}
