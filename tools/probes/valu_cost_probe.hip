// Hardware probe (kernel-tuning aid; round 5): issue cost of the vector instructions the tower kernels use, 2 and 4 waves
// per SIMD, independent registers, wall clock (HIP events) per instruction and SIMD in shader cycles at the clock the
// MFMA-only run of coexec_probe holds (2.2 GHz: printed ns too).
//   hipcc --offload-arch=gfx950 -O3 -o valu_cost_probe valu_cost_probe.hip && ./valu_cost_probe
#include <hip/hip_runtime.h>
#include <cstdio>

template <int OP>
__global__ __launch_bounds__(1024) void probe(float *out, int iters) {
  const int tid = threadIdx.x;
  float v[16];
  unsigned u[16];
  for (int i = 0; i < 16; ++i) {
    v[i] = 1.0f + 0.001f * (tid + i);
    u[i] = tid * 2654435761u + i;
  }
  const float m = 0.999f, c = 0.001f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(m), "v"(c));
        if (OP == 1) asm volatile("v_max_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c));
        if (OP == 2) asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0" : "+v"(u[i]) : "v"(v[i]), "v"(m));
        if (OP == 3) asm volatile("v_fma_mixhi_f16 %0, %1, %2, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(u[i]) : "v"(v[i]), "v"(m));
        if (OP == 4) asm volatile("v_cndmask_b32 %0, 0, %0, vcc" : "+v"(u[i]));
        if (OP == 5) asm volatile("v_cmp_lt_f32 vcc, 0, %0" ::"v"(v[i]) : "vcc");
        if (OP == 6) asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(u[i]) : "v"(u[(i + 1) & 15]));
        if (OP == 7) asm volatile("v_add_u32 %0, 0x7fffffff, %0" : "+v"(u[i]));
        if (OP == 8) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(u[i]), "+v"(u[(i + 8) & 15]));
        if (OP == 9) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[i]) : "v"(m));
        if (OP == 10) asm volatile("v_fma_mix_f32 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(v[i]) : "v"(v[(i + 1) & 15]), "v"(m), "v"(u[i]));
        if (OP == 11) asm volatile("v_and_b32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 3) & 15]));
        if (OP == 12) asm volatile("v_bfe_i32 %0, %0, 3, 1" : "+v"(u[i]));
        if (OP == 13) asm volatile("v_pk_mul_f16 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 5) & 15]));
        if (OP == 14) asm volatile("v_mov_b32 %0, %1" : "=v"(u[i]) : "v"(u[(i + 7) & 15]));
        if (OP == 16) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(u[i]) : "v"(v[i]), "v"(v[(i + 1) & 15]));
        if (OP == 17) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(u[i]) : "v"(v[i]), "v"(v[(i + 1) & 15]));
        if (OP == 18) asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(u[i]) : "v"(v[i]));
        if (OP == 19) asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(u[i]) : "v"(u[(i + 1) & 15]), "v"(u[(i + 2) & 15]));
        if (OP == 20) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u[i]) : "v"(v[i]), "v"(v[(i + 1) & 15]));
        if (OP == 21) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(u[i]) : "v"(u[(i + 1) & 15]), "v"(u[(i + 2) & 15]), "v"(u[(i + 3) & 15]));
        if (OP == 22) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c));
        if (OP == 23) asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(v[i]) : "v"(u[(i + 1) & 15]));
        if (OP == 24) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(v[i]) : "v"(m), "v"(c));
        if (OP == 25) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(*reinterpret_cast<double *>(&v[(2 * i) & 15])) : "v"(*reinterpret_cast<double *>(&v[(2 * i + 4) & 15])), "v"(*reinterpret_cast<double *>(&v[(2 * i + 8) & 15])));
        if (OP == 26) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*reinterpret_cast<double *>(&v[(2 * i) & 15])) : "v"(*reinterpret_cast<double *>(&v[(2 * i + 4) & 15])));
        if (OP == 27) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
        if (OP == 28) asm volatile("v_rcp_f32 %0, %0" : "+v"(v[i]));
        if (OP == 15) asm volatile("v_cmp_lt_f32 s[20:21], 0, %0\n\tv_cndmask_b32 %1, 0, %1, s[20:21]" : : "v"(v[i]), "v"(u[i]) : "s20", "s21");
      }
  }
  float total = 0.0f;
  for (int i = 0; i < 16; ++i) total += v[i] + (float)u[i];
  out[blockIdx.x * blockDim.x + tid] = total;
}

template <int OP>
static void run(const char *name, int insts_per_slot, float *out) {
  for (int waves : {2, 4}) {
    const int threads = 256 * waves, grid = 256, iters = 2000;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&probe<OP>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    probe<OP><<<grid, threads, 100 * 1024>>>(out, iters);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    for (int rep = 0; rep < 5; ++rep) probe<OP><<<grid, threads, 100 * 1024>>>(out, iters);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms = 0.0f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double ns = ms / 5 * 1e6 / ((double)iters * 64 * insts_per_slot * waves);
    std::printf("%-44s waves/simd %d  %.3f ns = %.2f cycles at 2.2 GHz per instruction and SIMD\n", name, waves, ns, ns * 2.2);
  }
}

int main() {
  float *out;
  (void)hipMalloc(&out, 256 * 1024 * 4);
  run<0>("v_fma_f32", 1, out);
  run<1>("v_max_f32", 1, out);
  run<9>("v_mul_f32", 1, out);
  run<2>("v_fma_mixlo_f16 (f32 x f32 -> f16)", 1, out);
  run<3>("v_fma_mixhi_f16 (.. - f16 -> f16)", 1, out);
  run<10>("v_fma_mix_f32 (f32 x f32 - f16 -> f32)", 1, out);
  run<5>("v_cmp_lt_f32 -> vcc", 1, out);
  run<4>("v_cndmask_b32 (vcc)", 1, out);
  run<15>("v_cmp -> sgpr pair ; v_cndmask (sgpr)", 2, out);
  run<6>("v_alignbit_b32", 1, out);
  run<7>("v_add_u32 (literal)", 1, out);
  run<11>("v_and_b32", 1, out);
  run<12>("v_bfe_i32", 1, out);
  run<13>("v_pk_mul_f16", 1, out);
  run<14>("v_mov_b32", 1, out);
  run<8>("v_permlane32_swap_b32", 1, out);
  run<16>("v_cvt_pk_f16_f32 (two f32 -> packed f16, rne)", 1, out);
  run<17>("v_cvt_pkrtz_f16_f32", 1, out);
  run<18>("v_cvt_f16_f32", 1, out);
  run<20>("v_cvt_pk_bf16_f32", 1, out);
  run<19>("v_pk_fma_f16", 1, out);
  run<21>("v_perm_b32", 1, out);
  run<22>("v_sub_f32", 1, out);
  run<23>("v_ldexp_f32", 1, out);
  run<24>("v_fmac_f32 (vop2)", 1, out);
  run<25>("v_pk_fma_f32 (two fmas)", 1, out);
  run<26>("v_pk_mul_f32 (two muls)", 1, out);
  run<27>("v_exp_f32", 1, out);
  run<28>("v_rcp_f32", 1, out);
  return 0;
}
