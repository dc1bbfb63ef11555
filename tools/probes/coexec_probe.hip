// Hardware probe (kernel-tuning aid, not part of the library; round 5): how much of a wave's vector-ALU work the SIMD
// puts UNDER another wave's (or its own) matrix work.  Every wave repeats
//   MODE 0 (interleaved): 8 x [one v_mfma_f32_16x16x32_f16 + VPM v_fma_f32], independent registers throughout;
//   MODE 1 (phased):      8 x 8 MFMAs back to back, then 8 x 8 x VPM fmas (a matrix loop, then an epilogue);
//   MODE 3 / 4: the matrix / the vector phase of mode 1 alone (what the two cost apart);
//   MODE 2 (phased, alternate waves half a period apart): waves with bit SHIFT of their index set start with the vector phase.
// at 1, 2 and 4 waves per SIMD (workgroups of 256 / 512 / 1024 threads, one per CU: 100 KiB of LDS requested), VPM = 1..4
// vector instructions per MFMA (the tower kernels run 2.5-3.5).  Operands live in registers: nothing but issue is
// measured.  Prints the cycles one wave needs per MFMA (s_memtime deltas, median over waves) beside the two bounds
// "serial" = waves x (16 + 4 VPM) and "perfect" = waves x max(16, 4 VPM).
//   hipcc --offload-arch=gfx950 -O3 -o coexec_probe coexec_probe.hip && ./coexec_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

template <int MODE, int VPM, int SHIFT = 0>
__global__ __launch_bounds__(1024) void probe(float *out, unsigned long long *stamps, int iters) {
  const int tid = threadIdx.x, wave = tid >> 6;
  half8 a, b;
  for (int i = 0; i < 8; ++i) {
    a[i] = (_Float16)(0.25f + 0.001f * ((tid * 7 + i * 13) & 255));
    b[i] = (_Float16)(0.5f - 0.002f * ((tid * 11 + i * 5) & 255));
  }
  f32x4 acc[8];
  float v[8 * 4];
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
  for (int i = 0; i < 32; ++i) v[i] = 1.0f + 0.001f * (tid + i);
  const float m = 0.999f, c = 0.001f;
  auto mfma = [&](int i) { acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0); };
  auto valu = [&](int i) {
#pragma unroll
    for (int k = 0; k < VPM; ++k) v[(i * 4 + k) & 31] = __builtin_fmaf(v[(i * 4 + k) & 31], m, c);
  };
  __syncthreads();
  if (MODE == 2 && ((wave >> SHIFT) & 1)) {  // half a period ahead: the vector phase first
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < 8; ++i) valu(i);
  }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        mfma(i);
        valu(i);
      }
    } else {
      if (MODE != 4) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
          for (int i = 0; i < 8; ++i) mfma(i);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (MODE != 3) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
          for (int i = 0; i < 8; ++i) valu(i);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float total = 0.0f;
  for (int i = 0; i < 8; ++i) total += acc[i][0] + acc[i][3];
  for (int i = 0; i < 32; ++i) total += v[i];
  out[blockIdx.x * blockDim.x + tid] = total;
  if ((tid & 63) == 0) stamps[blockIdx.x * 16 + wave] = t1 - t0;
}

template <int MODE, int VPM, int SHIFT = 0>
static void run(int waves_per_simd, float *out, unsigned long long *stamps) {
  const int threads = 256 * waves_per_simd, grid = 256, iters = MODE == 0 ? 4000 : 500;
  const int mfmas = iters * (MODE == 0 ? 8 : 64);
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&probe<MODE, VPM, SHIFT>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  probe<MODE, VPM, SHIFT><<<grid, threads, 100 * 1024>>>(out, stamps, iters);
  (void)hipDeviceSynchronize();
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0);
  for (int rep = 0; rep < 5; ++rep) probe<MODE, VPM, SHIFT><<<grid, threads, 100 * 1024>>>(out, stamps, iters);
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms = 0.0f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double ns_per_mfma_per_simd = ms / 5 * 1e6 / ((double)mfmas * waves_per_simd);  // wall time per MFMA slot of a SIMD
  std::vector<unsigned long long> h(grid * 16);
  (void)hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost);
  std::vector<double> per;
  for (int g = 0; g < grid; ++g)
    for (int w = 0; w < 4 * waves_per_simd; ++w) per.push_back((double)h[g * 16 + w] / mfmas);
  std::sort(per.begin(), per.end());
  // s_memtime ticks at 100 MHz on this chip: convert with the shader clock measured by the caller? report raw ticks x 24
  // (2.4 GHz nominal) AND the ratio to the serial bound, which needs no clock
  const double med = per[per.size() / 2];
  const double serial = waves_per_simd * (16.0 + 4.0 * VPM), perfect = waves_per_simd * std::max(16.0, 4.0 * VPM);
  std::printf("mode %d  vpm %d  stagger bit %d  waves/simd %d  cycles per mfma per wave %6.2f  (serial %3.0f, perfect %3.0f)   wall ns per MFMA and SIMD %.3f\n",
              MODE, VPM, SHIFT, waves_per_simd, med, serial, perfect, ns_per_mfma_per_simd);
}

int main() {
  float *out;
  unsigned long long *stamps;
  (void)hipMalloc(&out, 256 * 1024 * 4);
  (void)hipMalloc(&stamps, 256 * 16 * 8);
  for (int w : {1, 2, 4}) {
    run<0, 1>(w, out, stamps);
    run<0, 3>(w, out, stamps);
    run<0, 4>(w, out, stamps);
    run<1, 3>(w, out, stamps);
    run<2, 3, 0>(w, out, stamps);
    run<2, 3, 1>(w, out, stamps);
    run<2, 3, 2>(w, out, stamps);
    run<2, 3, 3>(w, out, stamps);
    run<3, 3>(w, out, stamps);
    run<4, 3>(w, out, stamps);
  }
  return 0;
}
