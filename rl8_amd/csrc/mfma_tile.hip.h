// Shared by the MFMA kernels (mlp_kernels.hip: the 256-wide towers; lstm_kernels.hip:
// the 256-wide LSTM): vector types, buffer-descriptor accessors, the fragment
// layout of a packed 256x256 weight matrix and the software-pipelined tile product
// over it.
#pragma once
#include "common.hip.h"

namespace rl8 {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kHidden = 256;
constexpr int kTileRows = 64;         // samples per workgroup tile
constexpr int kLdsStride = kHidden + 1;
constexpr int kMaxIn = 16;
constexpr int kMaxOut = 8;
constexpr int kGroups = kHidden / 8;  // k-groups of 8


typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// Raw buffer descriptors (gfx9 dword3 = 32-bit data format): loads / stores
// through them take their row offset from an SGPR, so the address arithmetic
// runs on the scalar unit, and anything past `bytes` is dropped by the hardware
// -- which is the row guard of a partial last tile.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t buffer_rsrc(const void *base, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, bytes, 0x00020000);
}
__device__ __forceinline__ void buffer_store_f32(float v, __amdgpu_buffer_rsrc_t r, int voffset,
                                                 int soffset) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, voffset, soffset, 0);
}
// ... for data written once and read by a LATER kernel behind gigabytes of other traffic
// (aux = sc1 | nt: streaming, no reuse expected): 0.8 % on the training forward and,
// through what stays in L2, 1 % on the weight-gradient kernel that reads h2 back.
__device__ __forceinline__ void buffer_store_f32_streaming(float v, __amdgpu_buffer_rsrc_t r, int voffset,
                                                           int soffset) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, voffset, soffset, 16 | 2);
}
__device__ __forceinline__ float buffer_load_f32(__amdgpu_buffer_rsrc_t r, int voffset, int soffset) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voffset, soffset, 0));
}
// (Only the 32-bit forms: this toolchain lowers the b64 / b128 load builtins to
// a single dword.)

// COST MODEL (measured, tools/probes/mfma_valu_probe.hip): on gfx950 an fp32 MFMA
// and ordinary VALU instructions do NOT overlap -- not from the same wave and not
// from another wave of the same SIMD.  A stream of v_mfma_f32_32x32x2_f32 runs at
// 155 TFLOP/s (64 cycles each at 2.37 GHz); every VALU instruction in between
// adds ~2.2 ns (~5 cycles) plus ~5 ns per MFMA->VALU->MFMA switch.  LDS and
// memory instructions do overlap.  So a kernel's time is  MFMA + sum(VALU), and
// these kernels are written to minimise the VALU instruction count per tile:
// addresses come from SGPRs / immediates (buffer loads, fully unrolled LDS
// offsets), pairs go through v_pk_fma_f32, and nothing is recomputed on the VALU
// that a memory instruction can fetch.
//
// One 64-row x 256-col GEMM tile on the matrix cores:
//   acc[m][n] = A_tile[32m + i][k] * Bp[(N-tile 2*wave + n)][k]
// A_tile: LDS [64][257]; Bp: fragment-packed [8][32][4][64] floats behind a buffer
// descriptor.  Wave `wave` produces output columns [64*wave, 64*wave + 64).
// Software-pipelined by hand (struct TileGemm below).
//
// Operand fragments of one k-group: B (weights, through L2) and A (activations,
// LDS) are prefetched at different distances, so they are separate sets.
struct BFrag {
  float b0[4], b1[4];
};
template <int KG>  // k-groups per N-tile in the packed weights
__device__ __forceinline__ void load_b(BFrag &f, __amdgpu_buffer_rsrc_t bp, int bvoff, int g) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    f.b0[e] = buffer_load_f32(bp, bvoff + e * (kWave * 4), g * (kWave * 16));
    f.b1[e] = buffer_load_f32(bp, bvoff + KG * kWave * 16 + e * (kWave * 4), g * (kWave * 16));
  }
}

template <int MT>  // 32-row M-tiles of the A operand
struct AFragT {
  float a[MT][4];
};

template <int MT, int STRIDE>
__device__ __forceinline__ void load_a(AFragT<MT> &f, const float *__restrict__ a0p, int g) {
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int m = 0; m < MT; ++m) f.a[m][e] = a0p[32 * m * STRIDE + 8 * g + e];
}

__device__ __forceinline__ void wait_vmcnt0() { __builtin_amdgcn_s_waitcnt(0x0f70); }


// ReLU gates of the backward pass, `gate > 0 ? value : 0`, as a compare into its
// OWN SGPR pair and a select on it.  The compiler funnels every compare through
// vcc, which chains compare -> (2 wait states) -> select -> compare ...; each
// link of such a chain is a gap in which the SIMD's other wave starts an MFMA.
// Callers issue a batch of compares, then the batch of selects (>= 2
// instructions apart, the gfx940-family VALU-SGPR read hazard).
__device__ __forceinline__ unsigned long long positive_mask(float gate) {
  unsigned long long m;
  asm("v_cmp_lt_f32_e64 %0, 0, %1" : "=s"(m) : "v"(gate));
  return m;
}
__device__ __forceinline__ float select_or_zero(unsigned long long mask, float value) {
  float r;
  asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(r) : "v"(value), "s"(mask));
  return r;
}

// max(v, 0) as exactly one v_max_f32 (fmaxf() costs a second, canonicalising,
// v_max on values the compiler cannot prove quiet; NaN -> 0 either way).
__device__ __forceinline__ float relu1(float v) {
  float r;
  asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(v));
  return r;
}

// FIRST: the accumulators start from the inline constant 0 (no v_mov per
// accumulator register).
template <bool FIRST, int MT>
__device__ __forceinline__ void mma_frag(const AFragT<MT> &fa, const BFrag &fb, f32x16 (&acc)[MT][2]) {
  if constexpr (FIRST) {
    const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      acc[m][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.a[m][0], fb.b0[0], zero, 0, 0, 0);
      acc[m][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.a[m][0], fb.b1[0], zero, 0, 0, 0);
    }
  }
#pragma unroll
  for (int e = FIRST ? 1 : 0; e < 4; ++e)
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      acc[m][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.a[m][e], fb.b0[e], acc[m][0], 0, 0, 0);
      acc[m][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.a[m][e], fb.b1[e], acc[m][1], 0, 0, 0);
    }
}

// The 64x256x256 tile product, in two parts so that the first weight fragments
// can be requested BEFORE the VALU phase that precedes the matrix loop:
//   TileGemm gemm(w2 descriptor, wave, lane);      (TileGemmT<MT>: MT x 32 rows)
//   gemm.prefetch();                 // B fragments of k-groups 0..2 -> registers
//   ... VALU phase writing the A tile to LDS (and storing to HBM), barrier ...
//   gemm.run(a_tile, acc);
// Why: loads and stores retire through one in-order counter (vmcnt), so a
// weight load issued after the phase's 64 stores cannot be waited for until
// those stores have drained to L2 (~1 us); requested ahead of them it is ready
// when the loop starts.  Inside the loop B fragments run three sets deep (two
// groups = ~0.9 us ahead), A fragments (LDS) two sets.
constexpr int kValuPhasePriority = 2;

// MT: 32-row M-tiles; KG: k-groups of 8 (K = 8*KG; 33 = the LSTM's 256 hidden +
// up to 7 inputs + the bias column); STRIDE: row pitch of the LDS tile in floats.
template <int MT, int KG = kGroups, int STRIDE = kLdsStride>
struct TileGemmT {
  static_assert(KG == 32 || KG == 33, "the tail of run() is written out for these depths");
  __amdgpu_buffer_rsrc_t bp;
  int bvoff, a_off;
  BFrag b[3];

  __device__ __forceinline__ TileGemmT(__amdgpu_buffer_rsrc_t w, int wave, int lane)
      : bp(w), bvoff((2 * wave) * KG * kWave * 16 + lane * 4),
        a_off((lane & 31) * STRIDE + 4 * (lane >> 5)) {}

  __device__ __forceinline__ void prefetch() {
    load_b<KG>(b[0], bp, bvoff, 0);
    load_b<KG>(b[1], bp, bvoff, 1);
    load_b<KG>(b[2], bp, bvoff, 2);
  }

  template <bool FIRST>
  __device__ __forceinline__ void step(AFragT<MT> &fa, BFrag &fb, const float *a0p, int g,
                                       f32x16 (&acc)[MT][2]) {
    mma_frag<FIRST, MT>(fa, fb, acc);
    load_a<MT, STRIDE>(fa, a0p, g + 2 < KG ? g + 2 : KG - 1);  // (the last reloads are unused)
    load_b<KG>(fb, bp, bvoff, g + 3 < KG ? g + 3 : KG - 1);
  }

  // ACCUMULATE: add to what `acc` already holds instead of starting from zero.
  template <bool ACCUMULATE = false>
  __device__ __forceinline__ void run(const float *__restrict__ a_tile, f32x16 (&acc)[MT][2]) {
    // Wave priority: the matrix loop runs at the lowest priority and every VALU
    // phase at a raised one.  Two workgroups share each SIMD; at equal priority
    // the arbiter alternates one workgroup's VALU instructions with the other's
    // MFMAs one for one, and each such switch costs ~17 cycles of the matrix pipe
    // (measured with phase timestamps: a 140-instruction VALU phase took 9 000
    // cycles).  Raised, the VALU phase issues as a burst (~5 cycles per
    // instruction) and the matrix wave simply resumes behind it.
    __builtin_amdgcn_s_setprio(0);
    const float *a0p = a_tile + a_off;
    AFragT<MT> a[2];
    load_a<MT, STRIDE>(a[0], a0p, 0);
    load_a<MT, STRIDE>(a[1], a0p, 1);
    // A set = g mod 2, B set = g mod 3: six groups per trip.
    step<!ACCUMULATE>(a[0], b[0], a0p, 0, acc);
    step<false>(a[1], b[1], a0p, 1, acc);
    step<false>(a[0], b[2], a0p, 2, acc);
    step<false>(a[1], b[0], a0p, 3, acc);
    step<false>(a[0], b[1], a0p, 4, acc);
    step<false>(a[1], b[2], a0p, 5, acc);
#pragma unroll 1
    for (int g = 6; g < 30; g += 6) {
      step<false>(a[0], b[0], a0p, g, acc);
      step<false>(a[1], b[1], a0p, g + 1, acc);
      step<false>(a[0], b[2], a0p, g + 2, acc);
      step<false>(a[1], b[0], a0p, g + 3, acc);
      step<false>(a[0], b[1], a0p, g + 4, acc);
      step<false>(a[1], b[2], a0p, g + 5, acc);
    }
    // groups 30.. : A sets hold 30, 31; B sets hold 30, 31, 32
    if constexpr (KG == 32) {
      mma_frag<false, MT>(a[0], b[0], acc);
      mma_frag<false, MT>(a[1], b[1], acc);
    } else {
      step<false>(a[0], b[0], a0p, 30, acc);  // also fetches A of group 32
      mma_frag<false, MT>(a[1], b[1], acc);
      mma_frag<false, MT>(a[0], b[2], acc);
    }
    __builtin_amdgcn_s_setprio(kValuPhasePriority);
  }
};

using TileGemm = TileGemmT<2>;  // 64-row tiles (the towers)

constexpr int pad_out(int n) { return n <= 1 ? 1 : n <= 2 ? 2 : n <= 4 ? 4 : 8; }

}  // namespace rl8
