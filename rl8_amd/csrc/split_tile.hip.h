// Shared by the bf16-plane kernels (mlp_split_kernels.hip: the towers;
// lstm_split_kernels.hip: the LSTM): tile constants, LDS access through inline asm
// with hand-placed waits, operand-fragment registers, the six-product matrix step and
// the fp32 -> three-bf16-planes split.  See mlp_split_kernels.hip for the scheme.
#pragma once
#include <type_traits>

#include "mfma_tile.hip.h"

namespace rl8 {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// Kernel-tuning builds only (tools/diag_mlp.sh): -DRL8_DIAG_SKIP=<bits> drops one
// memory stream (forward: 8 h2 store, 32 h1 store; backward: 64 h2 loads, 128 dZ2
// stores, 256 h1 / x loads of the epilogue).  The shipped library is built with 0.
#ifndef RL8_DIAG_SKIP
#define RL8_DIAG_SKIP 0
#endif
constexpr int kSplitDiagSkip = RL8_DIAG_SKIP;
#ifndef RL8_H2_STORE_AUX
#define RL8_H2_STORE_AUX (16 | 2)  // cache policy of the training forward's h2 stores: sc1 | nt (streaming)
#endif

constexpr int kSplitRows = 128;                 // rows per macro tile
constexpr int kSplitSteps = kHidden / 16;       // k-steps of 16
// A chunk in LDS: [plane][k-half][row] x 16 B (one MFMA operand fragment per row
// and k-half).  The k-half stride is padded by 64 B so that the producers' 8-byte
// writes (lanes 4i..4i+3 = the four quarter-fragments of row i) fall on disjoint
// banks for the two k-halves.
constexpr int kSplitKhStride = kSplitRows * 16 + 64;
constexpr int kSplitPlaneStride = 2 * kSplitKhStride;
constexpr int kSplitABytes = 3 * kSplitPlaneStride;
constexpr int kSplitBBytes = 3 * 8 * 1024;                 // [column tile][plane] x 1 KiB
constexpr int kSplitStageBytes = kSplitABytes + kSplitBBytes;
constexpr int kSplitPackedBytes = kSplitSteps * kSplitBBytes;  // 393 216

// ---- LDS access, invisible to the compiler's wait-count insertion -------------
typedef __attribute__((address_space(3))) unsigned char lds_byte_t;
__device__ __forceinline__ unsigned lds_offset(const void *p) {
  return (unsigned)(uintptr_t)(lds_byte_t *)p;
}
template <int OFF>
__device__ __forceinline__ u32x4 lds_read_b128(unsigned addr) {
  u32x4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
template <int OFF>
__device__ __forceinline__ void lds_write_b128(unsigned addr, u32x4 v) {
  asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF) : "memory");
}
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
template <int OFF>
__device__ __forceinline__ void lds_write_b64(unsigned addr, u32x2 v) {
  asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF) : "memory");
}
__device__ __forceinline__ void lds_write_b32(unsigned addr, float v) {
  asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
__device__ __forceinline__ float lds_read_b32(unsigned addr) {
  float v;
  asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  return v;
}

// Lane index from the execution mask (no register has to carry threadIdx.x across
// the matrix loop for the code behind it).
__device__ __forceinline__ int lane_id() {
  return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}

// vgpr[lane LANE] = scalar value (this toolchain has no builtin for v_writelane_b32).
template <int LANE>
__device__ __forceinline__ int write_lane(int vgpr, uint32_t value) {
  // (s_nop: the scalar usually comes straight from a VALU compare, and the hazard
  // recognizer does not look inside inline asm -- gfx940-family VALU-writes-SGPR ->
  // VALU-reads-it wait states)
  asm("s_nop 1\n\tv_writelane_b32 %0, %1, %2" : "+v"(vgpr) : "s"(value), "n"(LANE));
  return vgpr;
}

// The operand registers of one k-step.  `m` first holds the mid planes and is
// re-loaded with the lo planes once the mid terms have been issued.
struct SplitFrags {
  u32x4 ah[2], bh[4], am[2], bm[4];
};
__device__ __forceinline__ void wait_lds_all(SplitFrags &f) {
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(f.ah[0]), "+v"(f.ah[1]), "+v"(f.bh[0]), "+v"(f.bh[1]), "+v"(f.bh[2]), "+v"(f.bh[3]),
                 "+v"(f.am[0]), "+v"(f.am[1]), "+v"(f.bm[0]), "+v"(f.bm[1]), "+v"(f.bm[2]), "+v"(f.bm[3]));
}

// lgkmcnt <= N, ordering the named fragments behind it.  LDS returns in order, so with
// ONLY ds_reads outstanding "all but the last N" have landed; a scalar-cache load in
// flight would break that count (it returns out of order) -- kernels using this keep
// their s_loads in front of the step's barrier (tests/test_kernel_resources.py checks
// the instruction stream for it).
template <int N>
__device__ __forceinline__ void wait_lds(u32x4 &a) {
  asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a) : "n"(N));
}
template <int N>
__device__ __forceinline__ void wait_lds(u32x4 &a, u32x4 &b) {
  asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N));
}
template <int N>
__device__ __forceinline__ void wait_lds(u32x4 &a, u32x4 &b, u32x4 &c, u32x4 &d) {
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N));
}
template <int N>
__device__ __forceinline__ void wait_lds(u32x4 &a, u32x4 &b, u32x4 &c, u32x4 &d, u32x4 &e, u32x4 &g) {
  asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(g) : "n"(N));
}

// Scalar-register operands fetched by hand (s_buffer_load, NO wait: the caller's next
// barrier with lgkmcnt(0) is the wait, scalar_tie() behind it the compiler's fence).
typedef float f32x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void scalar_tie(f32x8 &v) { asm volatile("" : "+s"(v)); }
// Raw buffer descriptor over [p, p + bytes) in scalar registers, and eight floats at
// byte OFF of it (floats past `bytes` read as zero).
__device__ __forceinline__ u32x4 scalar_rsrc(const float *p, int bytes) {
  const uint64_t a = (uint64_t)(uintptr_t)p;
  return u32x4{(uint32_t)a, (uint32_t)(a >> 32) & 0xffffu, (uint32_t)bytes, 0x00020000u};
}
template <int OFF>
__device__ __forceinline__ f32x8 scalar_buffer_load_x8(u32x4 rsrc) {
  f32x8 v;
  // (early clobber: a later load through the same descriptor must still find it intact)
  asm volatile("s_buffer_load_dwordx8 %0, %1, %2" : "=&s"(v) : "s"(rsrc), "n"(OFF));
  return v;
}

template <bool FIRST>
__device__ __forceinline__ void split_mma(const u32x4 (&a)[2], const u32x4 (&b)[4], f32x16 (&acc)[2][4]) {
  if constexpr ((kSplitDiagSkip & 2048) != 0) {  // tuning builds: no matrix work (one MFMA per group keeps the data flow)
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[0]), __builtin_bit_cast(bf16x8, b[0]),
                                                        FIRST ? f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0} : acc[0][0], 0, 0, 0);
    if constexpr (FIRST) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
          if (mt + nt) acc[mt][nt] = acc[0][0];
    }
    return;
  }
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      if constexpr (FIRST) {
        const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[mt]),
                                                              __builtin_bit_cast(bf16x8, b[nt]), zero, 0, 0, 0);
      } else {
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[mt]),
                                                              __builtin_bit_cast(bf16x8, b[nt]), acc[mt][nt], 0, 0, 0);
      }
    }
}

// The same products with the operand roles exchanged: the 32x32 block comes out
// TRANSPOSED in the accumulator -- lane = sample row, registers = output columns
// (r -> column (r & 3) + 8 (r >> 2) + 4 (lane >> 5)).  A- and B-fragments have the
// same register layout (lane = index mod 32, k-half = lane / 32), so this costs
// nothing, every dot product is the same sum, and four consecutive registers are four
// consecutive columns of one row: the forward kernel's h2 leaves as 16-byte stores
// and its head needs no cross-lane reduction tree (see the epilogue there).
template <bool FIRST>
__device__ __forceinline__ void split_mma_t(const u32x4 (&a)[2], const u32x4 (&b)[4], f32x16 (&acc)[2][4]) {
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
      acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, b[nt]),
                                                            __builtin_bit_cast(bf16x8, a[mt]),
                                                            FIRST ? zero : acc[mt][nt], 0, 0, 0);
    }
}

template <bool FIRST>
__device__ __forceinline__ void split_mma_row(const u32x4 &a, const u32x4 (&b)[4], f32x16 (&acc)[4]) {
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b[nt]),
                                                      FIRST ? zero : acc[nt], 0, 0, 0);
  }
}

// v (fp32 pair) -> the three packed bf16 pairs (element 0 in the low half).
// Deliberately scalar arithmetic, and this file is built with -fno-slp-vectorize:
// NO PACKED FP32 ARITHMETIC (v_pk_{fma,add,mul}_f32) IN KERNELS THAT INTERLEAVE
// VALU WORK WITH bf16 MFMAs.  What happened (round 1, fused weight-gradient kernel,
// one run in four: a few dW3 accumulators wrong in lanes 48..63, always the low half
// of a pair, off by one fma evaluated on "the value written two instructions
// later"), read off the failing build's listing (commit 7f18347, recompiled to
// ISA; tools/check_inflight_regs.py::packed_war finds 53 such windows in it):
//     v_pk_fma_f32 v[196:197], v[224:225], v[194:195], v[196:197]   ; reads v224 (low half)
//     v_pk_fma_f32 v[226:227], v[226:227], v[206:207], 0
//     v_mov_b32    v224, v225        ; pair-alignment shuffle for the NEXT packed op
// i.e. a packed op's SOURCE register overwritten one or two VALU slots later.  A
// write-after-read in program order is architecturally safe, and for single-pass
// VALU ops it is safe here too (these kernels are full of it, also directly behind
// MFMAs that read the register).  A packed fp32 op issues as two passes; beside
// bf16 MFMAs -- where VALU instructions are slotted between matrix passes instead of
// owning the pipe -- the second pass's operand fetch for the last quarter-wave came
// after the younger v_mov had written v224.  The fp32-MFMA kernels (mlp_kernels.hip)
// contain the same packed ops at the same distance and never failed: fp32 MFMAs do
// not co-issue with the VALU.  LLVM's gfx950 hazard recognizer has no rule for it, so
// the rule lives here: no packed ops (flag above, scalar code below), enforced on the
// shipped ISA by tests/test_kernel_resources.py (regex + packed_war scan).
// The other suspect was ruled out on the same listing: the CFG walk of
// tools/check_inflight_regs.py finds no instruction touching the destination of a
// hand-issued load (ds_read_b128 / s_buffer_load_dwordx8 with the wait in a later
// asm statement) before a covering s_waitcnt -- neither in the failing build nor in
// the shipped one, where the same test now asserts it for every kernel.
__device__ __forceinline__ void split_pair(float x0, float x1, uint32_t &hi, uint32_t &mid, uint32_t &lo) {
  const float r0 = x0 - __uint_as_float(__float_as_uint(x0) & 0xffff0000u);
  const float r1 = x1 - __uint_as_float(__float_as_uint(x1) & 0xffff0000u);
  const float q0 = r0 - __uint_as_float(__float_as_uint(r0) & 0xffff0000u);
  const float q1 = r1 - __uint_as_float(__float_as_uint(r1) & 0xffff0000u);
  hi = __builtin_amdgcn_perm(__float_as_uint(x1), __float_as_uint(x0), 0x07060302u);
  mid = __builtin_amdgcn_perm(__float_as_uint(r1), __float_as_uint(r0), 0x07060302u);
  lo = __builtin_amdgcn_perm(__float_as_uint(q1), __float_as_uint(q0), 0x07060302u);
}

// One level of the sum over the 32 lanes of each half-wave (after levels 0..4 the
// total is valid in lanes 16..31 / 48..63).  Callers run a level over a BATCH of
// independent values: a value's five levels are a dependent chain, and one chain
// at a time (with its LDS write behind it) cost the forward epilogue 3x its
// instruction count in cycles.
template <int LEVEL>
__device__ __forceinline__ float half_wave_sum_level(float v) {
  constexpr int ctrl = LEVEL == 0 ? 0xb1     // quad_perm [1,0,3,2]
                       : LEVEL == 1 ? 0x4e   // quad_perm [2,3,0,1]
                       : LEVEL == 2 ? 0x141  // row_half_mirror
                       : LEVEL == 3 ? 0x140  // row_mirror
                                    : 0x142; // row_bcast:15 -> rows 1, 3
  constexpr int row_mask = LEVEL == 4 ? 0xa : 0xf;
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, row_mask, 0xf, false));
}

// ---- fp16 two-plane scheme (mlp_f16_kernels.hip, lstm_split_kernels.hip) ----------------
// An operand scaled by a power of two into fp16's range is carried as hi = fp16_rn(v),
// lo = fp16_rn(v - hi) (22 significand bits); a*b = ah*bh + ah*bl + al*bh + O(2^-22 |a*b|):
// three v_mfma_f32_32x32x16_f16 per 16 k, products exact, fp32 accumulate.
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));

constexpr int kF16Top = 14;                                // scaled operands stay below 2^14 (fp16 max 65504)

// (x0, x1) -> packed fp16 pairs hi, lo (element 0 in the low half); scalar arithmetic, as
// split_pair(): no packed fp32 ops beside the MFMAs.
// (Round 5, tools/probes/valu_cost_probe.hip: on gfx950 v_fma_mix{lo,hi}_f16 issue at HALF rate -- 7.5 cycles per
// instruction and SIMD where v_fma_f32 / v_fma_mix_f32 take 3.8 and v_mul_f32 2.2 -- while v_cvt_pk_f16_f32 rounds TWO
// floats into a packed pair in 4.1.  So the planes are formed as  t = x s (exact: two muls), hi = cvt_pk(t0, t1), the
// residuals by v_fma_mix_f32 against the halves of hi (exact), lo = cvt_pk(r0, r1): the SAME bits as the four mix
// instructions gave for a power-of-two s, 20.6 cycles per pair instead of 30.2; the wide forms 25 instead of 38.)
__device__ __forceinline__ void f16_pair(float x0, float x1, uint32_t &hi, uint32_t &lo) {
  float r0, r1;
  asm("v_cvt_pk_f16_f32 %0, %4, %5\n\t"
      "v_fma_mix_f32 %2, %4, 1.0, -%0 op_sel_hi:[0,0,1]\n\t"
      "v_fma_mix_f32 %3, %5, 1.0, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
      "v_cvt_pk_f16_f32 %1, %2, %3"
      : "=&v"(hi), "=&v"(lo), "=&v"(r0), "=&v"(r1)
      : "v"(x0), "v"(x1));
}

// (x0, x1) * s -> packed fp16 pairs hi = fp16(x * s), lo = fp16(x * s - hi), s a power of two (x * s exact),
// element 0 in the low half.  v_fma_mix_f32 computes fma(a, b, c) in fp32 from fp32 or fp16 sources (op_sel_hi: which
// sources are fp16, op_sel: their half): the residual against a half of the packed hi without converting it back.
// (Rounds 3-4 formed both planes with v_fma_mix{lo,hi}_f16, four instructions per pair; see the note above f16_pair.)
__device__ __forceinline__ void f16_pair_scaled(float x0, float x1, float s, uint32_t &hi, uint32_t &lo) {
  float t0, t1;
  asm("v_mul_f32 %2, %4, %6\n\t"
      "v_mul_f32 %3, %5, %6\n\t"
      "v_cvt_pk_f16_f32 %0, %2, %3\n\t"
      "v_fma_mix_f32 %2, %4, %6, -%0 op_sel_hi:[0,0,1]\n\t"
      "v_fma_mix_f32 %3, %5, %6, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
      "v_cvt_pk_f16_f32 %1, %2, %3"
      : "=&v"(hi), "=&v"(lo), "=&v"(t0), "=&v"(t1)
      : "v"(x0), "v"(x1), "v"(s));
}

// The product t0 * d0, t1 * d1 (t in vector registers, d wave-uniform in scalar registers) as a packed fp16 pair
// hi = fp16(t * d) -- ONE rounding of the exact product -- and a WIDE low plane lo = fp16(2^11 (t * d - hi)): the residual
// (exact in fp32: v_fma_mix_f32 subtracts the fp16 value it just formed from the exact product) is scaled up before it
// is rounded, so that it stays a normal fp16 number for |t * d| down to 2^-13 instead of 2^-3 -- 27 binades below the
// top (2^14) with all 22 bits, not 17 -- and its absolute resolution is 2^-35, not 2^-24.  The matrix instruction undoes
// the factor: its other operand is the ReLU gate, 1.0 for the hi product and 2^-11 = fp16 0x1000 for this one (one
// v_and on the gate fragment: 0x3c00 & 0x1000).  k2048: a scalar register holding 2048.0f.  Eight instructions per pair, 25 cycles.
__device__ __forceinline__ void f16_pair_product_wide(float t0, float t1, float d0, float d1, float k2048, uint32_t &hi,
                                                      uint32_t &lo) {
  // (hi = fp16 of the fp32-rounded product -- a double rounding where the mix instruction had one -- and the residual
  // against THAT hi from the exact product: hi + 2^-11 lo is as close to t d as before)
  float r0, r1;
  asm("v_mul_f32 %2, %6, %4\n\t"
      "v_mul_f32 %3, %7, %5\n\t"
      "v_cvt_pk_f16_f32 %0, %2, %3\n\t"
      "v_fma_mix_f32 %2, %4, %6, -%0 op_sel_hi:[0,0,1]\n\t"
      "v_fma_mix_f32 %3, %5, %7, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
      "v_mul_f32 %2, %8, %2\n\t"
      "v_mul_f32 %3, %8, %3\n\t"
      "v_cvt_pk_f16_f32 %1, %2, %3"
      : "=&v"(hi), "=&v"(lo), "=&v"(r0), "=&v"(r1)
      : "v"(t0), "v"(t1), "s"(d0), "s"(d1), "s"(k2048));
}
constexpr uint32_t kF16GateLowMask = 0x10001000u;  // fp16 1.0 (0x3c00) -> 2^-11 (0x1000), both halves

// The same wide low plane for an operand that is a value times a power of two (the general weight gradient's dZ2):
// hi = fp16(x s), lo = fp16(2^11 (x s - hi)).  Its partner in the hi x lo product must carry the 2^-11: the OTHER
// operand's hi plane times 2^-11, formed on the fragment registers (f16_pair_times) once their own product is issued.
__device__ __forceinline__ void f16_pair_scaled_wide(float x0, float x1, float s, float k2048, uint32_t &hi, uint32_t &lo) {
  float r0, r1;
  asm("v_mul_f32 %2, %4, %6\n\t"
      "v_mul_f32 %3, %5, %6\n\t"
      "v_cvt_pk_f16_f32 %0, %2, %3\n\t"
      "v_fma_mix_f32 %2, %4, %6, -%0 op_sel_hi:[0,0,1]\n\t"
      "v_fma_mix_f32 %3, %5, %6, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
      "v_mul_f32 %2, %7, %2\n\t"
      "v_mul_f32 %3, %7, %3\n\t"
      "v_cvt_pk_f16_f32 %1, %2, %3"
      : "=&v"(hi), "=&v"(lo), "=&v"(r0), "=&v"(r1)
      : "v"(x0), "v"(x1), "v"(s), "s"(k2048));
}
// ... with a factor per element (not powers of two: hi = fp16(x s) is ONE rounding of the exact product, the residual
// fma(x, s, -hi) one fp32 rounding of the exact difference): the rows-shape data gradients' class 8, whose two elements of
// a pair are two ROWS with their own factors.
__device__ __forceinline__ void f16_pair_scaled2_wide(float x0, float x1, float s0, float s1, float k2048, uint32_t &hi, uint32_t &lo) {
  float r0, r1;
  asm("v_mul_f32 %2, %4, %6\n\t"
      "v_mul_f32 %3, %5, %7\n\t"
      "v_cvt_pk_f16_f32 %0, %2, %3\n\t"
      "v_fma_mix_f32 %2, %4, %6, -%0 op_sel_hi:[0,0,1]\n\t"
      "v_fma_mix_f32 %3, %5, %7, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
      "v_mul_f32 %2, %8, %2\n\t"
      "v_mul_f32 %3, %8, %3\n\t"
      "v_cvt_pk_f16_f32 %1, %2, %3"
      : "=&v"(hi), "=&v"(lo), "=&v"(r0), "=&v"(r1)
      : "v"(x0), "v"(x1), "v"(s0), "v"(s1), "s"(k2048));
}
// packed fp16 pair times a packed fp16 constant in a scalar register (0x10001000: both halves times 2^-11; exact for
// halves of 2^-3 and more, the fp16 subnormal quantum below)
__device__ __forceinline__ uint32_t f16_pair_times(uint32_t v, uint32_t packed_constant) {
  uint32_t out;
  asm("v_pk_mul_f16 %0, %1, %2" : "=v"(out) : "v"(v), "s"(packed_constant));
  return out;
}

// bound < 2^e for the power of two that scales an operand (2^(14 - e)); bounds below 2^-80
// (and zero) keep a finite factor: such operands are far below fp16's top anyway.
__device__ __forceinline__ int f16_bound_exponent(float bound) {
  const int e = __builtin_amdgcn_frexp_expf(bound);
  return e < -80 ? -80 : e;
}

// One plane product of the wave's 64 x 128 tile, operand roles exchanged (transposed
// accumulators: lane = row; see split_mma_t).
template <bool FIRST>
__device__ __forceinline__ void f16_mma_t(const u32x4 (&a)[2], const u32x4 (&b)[4], f32x16 (&acc)[2][4]) {
  if constexpr ((kSplitDiagSkip & 2048) != 0) {  // tuning builds: no matrix work (one MFMA per group keeps the data flow)
    const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, b[0]), __builtin_bit_cast(half8, a[0]),
                                                       FIRST ? zero : acc[0][0], 0, 0, 0);
    if constexpr (FIRST) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
          if (mt + nt) acc[mt][nt] = acc[0][0];
    }
    return;
  }
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
      acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, b[nt]), __builtin_bit_cast(half8, a[mt]),
                                                           FIRST ? zero : acc[mt][nt], 0, 0, 0);
    }
}

template <bool FIRST>
__device__ __forceinline__ void f16_mma(const u32x4 (&a)[2], const u32x4 (&b)[4], f32x16 (&acc)[2][4]) {
  if constexpr ((kSplitDiagSkip & 2048) != 0) {  // tuning builds: no matrix work (one MFMA per group keeps the data flow)
    const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, a[0]), __builtin_bit_cast(half8, b[0]),
                                                       FIRST ? zero : acc[0][0], 0, 0, 0);
    if constexpr (FIRST) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
          if (mt + nt) acc[mt][nt] = acc[0][0];
    }
    return;
  }
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
      acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, a[mt]), __builtin_bit_cast(half8, b[nt]),
                                                           FIRST ? zero : acc[mt][nt], 0, 0, 0);
    }
}


}  // namespace rl8
