// Hardware probe (kernel-tuning aid, not part of the library): what a 3-way
// bf16 split of fp32 GEMM operands (6 v_mfma_f32_32x32x16_bf16 per 16 k instead
// of 8 v_mfma_f32_32x32x2_f32) could buy on gfx950:
//   * the sustained bf16 MFMA rate on changing operands (clock under load),
//   * whether VALU work (the operand splitting) overlaps with bf16 MFMAs, from
//     the same wave and from the other wave of the SIMD (it does NOT with fp32
//     MFMAs: tools/probes/mfma_valu_probe.hip),
//   * the cost of the split itself (fp32 -> hi/mid/lo bf16, packed).
//   hipcc --offload-arch=gfx950 -O3 -o bf16_split_probe bf16_split_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

template <int K>
__device__ __forceinline__ void valu_block(float (&v)[8], float a, float b) {
#pragma unroll
  for (int k = 0; k < K; ++k)
    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v[k & 7]) : "v"(a), "v"(b));
}

__device__ __forceinline__ bf16x8 make_operand(unsigned &h) {
  bf16x8 r;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    h = h * 1664525u + 1013904223u;
    r[k] = (short)(0x3c00 + ((h >> 20) & 0x3ff));  // bf16 in [~0.0078, ~2) : no denormals / NaNs
    if (h & 0x80000u) r[k] |= (short)0x8000;
  }
  return r;
}

// MODE 0: every wave issues MFMA + K fmas per MFMA.  MODE 1: waves 0..3 MFMA,
// waves 4..7 (same SIMDs) fmas only.
template <int K, int MODE>
__global__ __launch_bounds__(512) void probe(float *out, int iters, int flag) {
  const int wave = threadIdx.x >> 6;
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
  unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
  bf16x8 a[4], b[4];
  for (int k = 0; k < 4; ++k) {
    a[k] = make_operand(h);
    b[k] = make_operand(h);
  }
  float v[8] = {1, 2, 3, 4, 5, 6, 7, 8};
  const float fa = 1.0f + threadIdx.x * 1e-9f, fb = 1e-9f;
  const bool mfma_wave = MODE != 1 || wave < 4;
  const bool valu_wave = MODE != 1 || wave >= 4;
  if (mfma_wave) {
    for (int it = 0; it < iters; it += 4) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a[i]), "v"(b[(i + j) & 3]));
          if (valu_wave) valu_block<K>(v, fa, fb);
        }
    }
  } else {
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 4; ++i) valu_block<K>(v, fa, fb);
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
  for (int w = 0; w < 5; ++w) probe<K, MODE><<<grid, threads>>>(out, iters, 0);
  hipDeviceSynchronize();
  float best = 1e30f, sum = 0;
  const int reps = 8;
  for (int rep = 0; rep < reps; ++rep) {
    hipEventRecord(e0);
    probe<K, MODE><<<grid, threads>>>(out, iters, 0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
    sum += ms;
  }
  const int mfma_waves = MODE == 1 ? 4 : threads / 64;
  const double mfmas = (double)grid * mfma_waves * iters * 4;
  const double per_simd = (double)iters * 4 * (MODE == 1 ? 1 : threads / 256);
  printf("%-30s threads=%3d K=%2d  best %8.3f ms  mean %8.3f ms  %7.1f TFLOP/s(bf16)  %6.2f ns/MFMA/SIMD\n", name, threads,
         K, best, sum / reps, mfmas * 32768.0 / (best * 1e-3) / 1e12, best * 1e6 / per_simd);
}

// The split: fp32 x -> (hi, mid, lo) bf16 by truncation (v_and / v_sub), packed two
// values per dword with v_perm_b32.  Timed alone (VALU-only kernel), per element.
__global__ __launch_bounds__(256) void split_cost(const float *src, unsigned *dst, int iters, int flag) {
  float x[8];
  for (int k = 0; k < 8; ++k) x[k] = src[threadIdx.x * 8 + k];
  unsigned sink = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 8; k += 2) {
      float x0 = x[k], x1 = x[k + 1];
      unsigned h0 = __float_as_uint(x0) & 0xffff0000u, h1 = __float_as_uint(x1) & 0xffff0000u;
      float r0 = x0 - __uint_as_float(h0), r1 = x1 - __uint_as_float(h1);
      unsigned m0 = __float_as_uint(r0) & 0xffff0000u, m1 = __float_as_uint(r1) & 0xffff0000u;
      float l0 = r0 - __uint_as_float(m0), l1 = r1 - __uint_as_float(m1);
      unsigned hi = __builtin_amdgcn_perm(h1, h0, 0x07060302u);
      unsigned mi = __builtin_amdgcn_perm(m1, m0, 0x07060302u);
      unsigned lo = __builtin_amdgcn_perm(__float_as_uint(l1), __float_as_uint(l0), 0x07060302u);
      sink ^= hi + mi * 3u + lo * 5u;
      x[k] = x0 * 1.0001f;
      x[k + 1] = x1 * 0.9999f;
    }
  }
  if (flag) dst[blockIdx.x * blockDim.x + threadIdx.x] = sink;
}

int main() {
  float *out;
  hipMalloc(&out, 256 * 512 * sizeof(float));
  const int iters = 20000;
  run<0, 0>("bf16 MFMA only", 256, out, iters);
  run<0, 0>("bf16 MFMA only", 512, out, iters);
  run<1, 0>("same wave +1 fma/MFMA", 256, out, iters);
  run<2, 0>("same wave +2 fma/MFMA", 256, out, iters);
  run<4, 0>("same wave +4 fma/MFMA", 256, out, iters);
  run<6, 0>("same wave +6 fma/MFMA", 256, out, iters);
  run<8, 0>("same wave +8 fma/MFMA", 256, out, iters);
  run<12, 0>("same wave +12 fma/MFMA", 256, out, iters);
  run<4, 0>("same wave +4 fma/MFMA", 512, out, iters);
  run<8, 0>("same wave +8 fma/MFMA", 512, out, iters);
  run<12, 0>("same wave +12 fma/MFMA", 512, out, iters);
  run<4, 1>("other wave 4 fma/MFMA", 512, out, iters);
  run<6, 1>("other wave 6 fma/MFMA", 512, out, iters);
  run<8, 1>("other wave 8 fma/MFMA", 512, out, iters);
  run<16, 1>("other wave 16 fma/MFMA", 512, out, iters);
  {
    float *src;
    unsigned *dst;
    hipMalloc(&src, 256 * 8 * sizeof(float));
    hipMalloc(&dst, 1024 * 256 * sizeof(unsigned));
    hipMemset(src, 0x3f, 256 * 8 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int it2 = 4000;
    for (int w = 0; w < 3; ++w) split_cost<<<1024, 256>>>(src, dst, it2, 0);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    split_cost<<<1024, 256>>>(src, dst, it2, 0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double elems = 1024.0 * 256 * 8 * it2;
    // 1024 WGs of 4 waves over 256 CUs = 1 wave per SIMD: per-SIMD element rate
    printf("split fp32 -> 3 x bf16 (packed): %.3f ms, %.2f ps/element/chip, %.2f ns per 64-lane element-row per SIMD\n", ms,
           ms * 1e9 / elems, ms * 1e6 / (it2 * 8.0));
  }
  return 0;
}
