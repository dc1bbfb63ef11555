// Hardware probe (kernel-tuning aid, not part of the library): how much VALU work
// can ride under a stream of v_mfma_f32_32x32x2_f32 on a gfx950 SIMD --
//   same   : K VALU fmas per MFMA issued by the SAME wave
//   other  : a second wave on the same SIMD issues the VALU fmas
// and what the sustained fp32 MFMA rate is (which pins the clock under load).
//   hipcc --offload-arch=gfx950 -O3 -o mfma_valu_probe mfma_valu_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int K>
__device__ __forceinline__ void valu_block(float (&v)[8], float a, float b) {
#pragma unroll
  for (int k = 0; k < K; ++k)
    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v[k & 7]) : "v"(a), "v"(b));
}

// MODE 0: every wave does MFMA + K fmas per MFMA.  MODE 1: waves 0..3 MFMA only,
// waves 4..7 (same SIMDs) K fmas per (the other wave's) MFMA.  MODE 2: LDS reads
// (ds_read_b32) instead of fmas, same wave.
template <int K, int MODE>
__global__ __launch_bounds__(512) void probe(float *out, int iters, int flag) {
  __shared__ float lds[4096];
  const int wave = threadIdx.x >> 6;
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
  float v[8] = {1, 2, 3, 4, 5, 6, 7, 8};
  float a = 1.0f + threadIdx.x * 1e-9f, b = 1e-9f;
  lds[threadIdx.x] = a;
  __syncthreads();
  const bool mfma_wave = MODE != 1 || wave < 4;
  const bool valu_wave = MODE != 1 || wave >= 4;
  if (mfma_wave && valu_wave) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
        if (MODE == 2) {
#pragma unroll
          for (int k = 0; k < K; ++k) {
            float t;
            asm volatile("ds_read_b32 %0, %1" : "=v"(t) : "v"((threadIdx.x * 4 + k * 64) & 16383));
            v[k & 7] = t;
          }
        } else {
          valu_block<K>(v, a, b);
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)");
  } else if (mfma_wave) {
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
  } else {
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 4; ++i) valu_block<K>(v, a, b);
  }
  if (flag) {
    float s = 0;
    for (int i = 0; i < 4; ++i)
      for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int k = 0; k < 8; ++k) s += v[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  }
}

template <int K, int MODE>
void run(const char *name, int threads, float *out, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int grid = 256;
  for (int w = 0; w < 3; ++w) probe<K, MODE><<<grid, threads>>>(out, iters, 0);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0);
    probe<K, MODE><<<grid, threads>>>(out, iters, 0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const int mfma_waves = MODE == 1 ? 4 : threads / 64;
  const double mfmas = (double)grid * mfma_waves * iters * 4;
  const double tflops = mfmas * 4096.0 / (best * 1e-3) / 1e12;
  // cycles per MFMA per SIMD if the clock were 2.4 GHz
  const double mfma_per_simd = (double)iters * 4 * (MODE == 1 ? 1 : threads / 256);
  printf("%-28s threads=%3d K=%2d  %8.3f ms  %7.1f TFLOP/s  %6.1f ns/MFMA/SIMD\n", name, threads, K, best,
         tflops, best * 1e6 / mfma_per_simd);
}

// K buffer loads (dword, or dwordx4 with MODE4 = true) per MFMA from the same
// wave, out of a 64-KiB L2-resident buffer: what a vector-memory instruction
// costs the matrix pipe.
template <int K, bool X4>
__global__ __launch_bounds__(256) void probe_vmem(const float *src, float *out, int iters, int flag) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
  float a = 1.0f + threadIdx.x * 1e-9f, b = 1e-9f;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, 65536, 0x00020000);
  const int voff = (threadIdx.x & 63) * (X4 ? 16 : 4);
  float sink = 0.0f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
#pragma unroll
      for (int k = 0; k < K; ++k) {
        if constexpr (X4) {
          typedef float f4 __attribute__((ext_vector_type(4)));
          f4 t;
          asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen offset:%3" : "=v"(t) : "v"(voff), "s"(rs), "n"(k * 1024));
          asm volatile("" ::"v"(t));
        } else {
          float t;
          asm volatile("buffer_load_dword %0, %1, %2, 0 offen offset:%3" : "=v"(t) : "v"(voff), "s"(rs), "n"(k * 256));
          asm volatile("" ::"v"(t));
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(8)");
  }
  asm volatile("s_waitcnt vmcnt(0)");
  if (flag) {
    float s = sink;
    for (int i = 0; i < 4; ++i)
      for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  }
}

template <int K, bool X4>
void run_vmem(const float *src, float *out, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) probe_vmem<K, X4><<<256, 256>>>(src, out, iters, 0);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0);
    probe_vmem<K, X4><<<256, 256>>>(src, out, iters, 0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double mfmas = 256.0 * 4 * iters * 4;
  printf("%-28s threads=256 K=%2d  %8.3f ms  %7.1f TFLOP/s  %6.1f ns/MFMA/SIMD\n",
         X4 ? "same wave, buffer dwordx4" : "same wave, buffer dword", K, best, mfmas * 4096.0 / (best * 1e-3) / 1e12,
         best * 1e6 / (iters * 4.0));
}

// Power: the same MFMA stream on operands that change every instruction
// (pseudo-random floats) -- what the matrix pipe sustains on real data.
__global__ __launch_bounds__(256) void probe_random(float *out, int iters, int flag) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
  float a[8], b[8];
  unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
  for (int k = 0; k < 8; ++k) {
    h = h * 1664525u + 1013904223u;
    a[k] = (float)(int)(h >> 8) * (1.0f / 8388608.0f) - 1.0f;
    h = h * 1664525u + 1013904223u;
    b[k] = (float)(int)(h >> 8) * (1.0f / 8388608.0f) - 1.0f;
  }
  for (int it = 0; it < iters; it += 16) {  // 64 MFMAs per trip, every operand pair different
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int k = 0; k < 8; ++k)
        asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc[k & 3]) : "v"(a[k]), "v"(b[(k + j) & 7]));
  }
  if (flag) {
    float s = 0;
    for (int i = 0; i < 4; ++i)
      for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  }
}

void run_random(float *out, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int w = 0; w < 10; ++w) probe_random<<<256, 256>>>(out, iters, 0);
  hipDeviceSynchronize();
  float best = 1e30f, sum = 0;
  for (int rep = 0; rep < 10; ++rep) {
    hipEventRecord(e0);
    probe_random<<<256, 256>>>(out, iters, 0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
    sum += ms;
  }
  const double mfmas = 256.0 * 4 * iters * 4;
  printf("%-28s threads=256       %8.3f ms best %8.3f ms mean  %7.1f / %7.1f TFLOP/s\n", "mfma only, random operands", best,
         sum / 10, mfmas * 4096.0 / (best * 1e-3) / 1e12, mfmas * 4096.0 / (sum / 10 * 1e-3) / 1e12);
}

int main() {
  float *out;
  hipMalloc(&out, 256 * 512 * sizeof(float));
  const int iters = 20000;
  // warm the clocks
  for (int i = 0; i < 20; ++i) probe<0, 0><<<256, 256>>>(out, iters, 0);
  hipDeviceSynchronize();
  run<0, 0>("mfma only, 1 wave/SIMD", 256, out, iters);
  run<0, 0>("mfma only, 2 waves/SIMD", 512, out, iters);
  run_random(out, iters);
  run_random(out, iters * 10);
  run<1, 0>("same wave", 256, out, iters);
  run<2, 0>("same wave", 256, out, iters);
  run<4, 0>("same wave", 256, out, iters);
  run<8, 0>("same wave", 256, out, iters);
  run<12, 0>("same wave", 256, out, iters);
  run<16, 0>("same wave", 256, out, iters);
  run<4, 0>("same wave, 2 waves/SIMD", 512, out, iters);
  run<8, 0>("same wave, 2 waves/SIMD", 512, out, iters);
  run<1, 1>("other wave", 512, out, iters);
  run<2, 1>("other wave", 512, out, iters);
  run<4, 1>("other wave", 512, out, iters);
  run<8, 1>("other wave", 512, out, iters);
  run<12, 1>("other wave", 512, out, iters);
  run<16, 1>("other wave", 512, out, iters);
  run<24, 1>("other wave", 512, out, iters);
  run<1, 2>("same wave, LDS reads", 256, out, iters);
  run<2, 2>("same wave, LDS reads", 256, out, iters);
  run<4, 2>("same wave, LDS reads", 256, out, iters);
  run<8, 2>("same wave, LDS reads", 256, out, iters);
  float *src;
  hipMalloc(&src, 65536);
  hipMemset(src, 0, 65536);
  run_vmem<1, false>(src, out, iters);
  run_vmem<2, false>(src, out, iters);
  run_vmem<4, false>(src, out, iters);
  run_vmem<1, true>(src, out, iters);
  run_vmem<2, true>(src, out, iters);
  hipFree(src);
  hipFree(out);
  return 0;
}
