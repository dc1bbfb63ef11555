// N1, fourth generation of the tower forward (round 3): the fp16 two-plane
// scheme of mlp_f16_kernels.hip (three v_mfma_f32_32x32x16_f16 per fp32 product, operands scaled by powers
// of two) with the work laid out so that EVERY WAVE OWNS COMPLETE ROWS:
//
//   * a wave computes 32 MT sample rows x all 256 output columns (MT x 8 accumulator blocks);
//   * its A operand -- h1 = relu(b1 + x . w1) of those rows, scaled and split into two fp16 planes -- is
//     produced by the wave itself directly in MFMA fragment layout (lane = row, k-half = lane / 32: the lane
//     computes the eight k of its own fragment) and never leaves its registers: no LDS stage for A, no
//     ds_write, no producer -> consumer barrier, no LDS exchange of the row factors (the lane that scaled a
//     row is the lane that holds its accumulators);
//   * only W2's planes go through LDS, as a RING of 16-KiB k-step chunks filled by direct-to-LDS loads RING-1
//     steps ahead of their use; the one barrier per k-step only publishes chunks whose loads were issued
//     steps ago (counted vmcnt), so nobody waits at it for data;
//   * the epilogue needs no workgroup barrier at all: the head's dot product, the gate bits and the h2
//     transpose are per wave (a wave has whole rows).
//
// What the producer costs per element (DIN = 1): fma, max and TWO conversions -- v_fma_mixlo/mixhi_f16 fold
// the row's power of two into the fp16 rounding (hi = fp16(h * s)) and the subtraction into the second one
// (lo = fp16(h * s - hi)): 4 VALU instructions instead of the 8 of the previous generation (scalar w1 / b1
// operands there cost a v_mov per fma: one SGPR per VALU instruction), w1 / b1 here are 16-byte broadcast
// reads of an LDS copy.
//
// Layouts: W2 planes as rl8_mlp_pack_w2_f16 writes them (chunk s = 16 one-KiB pieces [column tile][plane]);
// accumulators transposed as in mlp_f16_kernels.hip (lane = sample row, register r of block nt = column
// 32 nt + (r & 3) + 8 (r >> 2) + 4 (lane >> 5)).
#include "split_tile.hip.h"

namespace rl8 {

constexpr int kRowsChunk = 16 * 1024;       // one k-step of W2: [column tile 0..7][plane hi | lo] x 1 KiB
constexpr int kRowsPacked = kSplitSteps * kRowsChunk;  // (= kF16PackedBytes; two floats behind: scale, 1 / scale)
constexpr int kRowsPitch = 128 + 16;        // h2 transpose scratch: row pitch (conflict-free 16-byte accesses both ways)

constexpr int rows_consts_bytes(int k_in, int k_out) { return (2 + k_in + k_out) * kHidden * 4; }
constexpr int rows_lds_bytes(int ring, int k_in, int k_out, int mt, bool store) {
  return ring * kRowsChunk + rows_consts_bytes(k_in, k_out) + (store ? 4 * 32 * mt * kRowsPitch : 0);
}

// (x0, x1) * s -> packed fp16 pairs hi = fp16(x * s), lo = fp16(x * s - hi), s a power of two (x * s exact),
// element 0 in the low half.  v_fma_mix{lo,hi}_f16 compute fma(a, b, c) in fp32 from fp32 or fp16 sources
// (op_sel_hi: which sources are fp16, op_sel: their half) and round ONCE to fp16 into one half of the
// destination, keeping the other: four instructions per pair where cvt / cvt back / sub / cvt takes eight.
// (The compiler itself pairs mixlo + mixhi on one register like this -- no wait state between them.)
__device__ __forceinline__ void f16_pair_scaled(float x0, float x1, float s, uint32_t &hi, uint32_t &lo) {
  asm("v_fma_mixlo_f16 %0, %2, %4, 0\n\t"
      "v_fma_mixhi_f16 %0, %3, %4, 0\n\t"
      "v_fma_mixlo_f16 %1, %2, %4, -%0 op_sel_hi:[0,0,1]\n\t"
      "v_fma_mixhi_f16 %1, %3, %4, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
      : "=&v"(hi), "=&v"(lo)
      : "v"(x0), "v"(x1), "v"(s));
}

// SAVE: 0 inference; 1 training with h2 and its gate bits stored; 2 training with the gate bits alone.
// MT: 32-row blocks per wave (1: 128-row tiles, two workgroups per CU; 2: 256-row tiles, accumulators take
// 256 registers, one workgroup per CU).  RING: chunks of W2 in LDS (loads run RING - 1 steps ahead).
// DIAG (tuning builds only, -DRL8_ROWS_STAMP): 1 no production of the next fragments, 2 no epilogue arithmetic, 4 one
// product per column block instead of 3 MT (no matrix work), 8 the column blocks' LDS reads once per step only.
template <int DIN, int NOUT, int SAVE, int MT, int RING, int DIAG = 0>
__global__ __launch_bounds__(kBlock, MT == 1 ? 2 : 1) void mlp_rows_forward_kernel(
    const float *__restrict__ x, int64_t m, const float *__restrict__ w1, const float *__restrict__ b1,
    const void *__restrict__ w2s, const float *__restrict__ b2, const float *__restrict__ w3,
    const float *__restrict__ b3, float *__restrict__ out, float *__restrict__ save_h2,
    uint32_t *__restrict__ save_gate2
#ifdef RL8_ROWS_STAMP  // tuning builds (tools/diag/rows_clock.py): shader-clock / real-time stamps around the kernel
    , unsigned long long *__restrict__ stamps
#endif
) {
#ifdef RL8_ROWS_STAMP
  const unsigned long long stamp_t0 = __builtin_amdgcn_s_memtime(), stamp_r0 = __builtin_amdgcn_s_memrealtime();
#endif
  constexpr int kIn = DIN, d_in = DIN, n_out = NOUT;
  constexpr int kOut = pad_out(NOUT);
  constexpr int kTile = 128 * MT;  // rows per workgroup tile
  constexpr int kAhead = RING - 1;
  constexpr bool kStore = SAVE == 1;
  static_assert(rows_lds_bytes(RING, kIn, kOut, MT, kStore) <= (MT == 1 ? 80 : 160) * 1024, "LDS budget");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // [ring: RING x 16 KiB][b1 | w1 (transposed: [i][k]) | b2 | w3 rows (zero rows up to kOut)][h2 transpose scratch]
  const unsigned lds0 = lds_offset(smem);
  constexpr int kConstOff = RING * kRowsChunk;
  constexpr int kB2Off = kConstOff + (1 + kIn) * kHidden * 4;
  constexpr int kScratchOff = kConstOff + rows_consts_bytes(kIn, kOut);
  const int tid = threadIdx.x, lane = tid & 63, l32 = lane & 31, hh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const __amdgpu_buffer_rsrc_t w2rsrc = buffer_rsrc(w2s, kRowsPacked);
  const float inv_w2_scale = reinterpret_cast<const float *>(static_cast<const unsigned char *>(w2s) + kRowsPacked)[1];

  const int64_t tiles = (m + kTile - 1) / kTile;
  const int64_t stride = gridDim.x;
  if ((int64_t)blockIdx.x >= tiles) return;

  // ---- constants into LDS; bounds of layer 1 for the row factors ---------------------------------
  float b1max = 0.0f, w1max[kIn];
  {
    float *consts = reinterpret_cast<float *>(smem + kConstOff);
    consts[tid] = b1[tid];
#pragma unroll
    for (int i = 0; i < kIn; ++i) consts[(1 + i) * kHidden + tid] = w1[tid * d_in + i];
    consts[(1 + kIn) * kHidden + tid] = b2[tid];
#pragma unroll
    for (int q = 0; q < kOut; ++q) consts[(2 + kIn + q) * kHidden + tid] = q < n_out ? w3[q * kHidden + tid] : 0.0f;
    __syncthreads();
    for (int k = 0; k < kHidden; ++k) b1max = __builtin_fmaxf(b1max, __builtin_fabsf(consts[k]));
    b1max = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(b1max)));  // (uniform: scalar registers)
#pragma unroll
    for (int i = 0; i < kIn; ++i) {
      float mx = 0.0f;
      for (int k = 0; k < kHidden; ++k) mx = __builtin_fmaxf(mx, __builtin_fabsf(consts[(1 + i) * kHidden + k]));
      w1max[i] = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(mx)));
    }
  }

  // ---- per-lane state -------------------------------------------------------------------------------
  // rows of this lane: tile base + 32 (MT wave + mt) + l32 (both half-waves hold the same rows)
  auto load_x = [&](float (&dst)[MT][kIn], int64_t tile) {
    const int64_t r0 = tile * kTile;
    const float *base = x + r0 * d_in;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int row = 32 * (MT * wave + mt) + l32;
#pragma unroll
      for (int i = 0; i < kIn; ++i) dst[mt][i] = (tile < tiles && r0 + row < m) ? base[(unsigned)(row * d_in + i)] : 0.0f;
    }
  };
  // Row factor: |h1[row][k]| <= max|b1| + sum_i |x_i| max_k |w1[k][i]| < 2^e  =>  planes of h1 * 2^(14 - e)
  // (exact); the accumulators are multiplied back by 2^(e - 14) / (W2's power of two).
  auto row_scales = [&](const float (&xs)[MT][kIn], float (&scale)[MT], float (&inv)[MT]) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      float bound = b1max;
#pragma unroll
      for (int i = 0; i < kIn; ++i) bound = __builtin_fmaf(__builtin_fabsf(xs[mt][i]), w1max[i], bound);
      const int e = f16_bound_exponent(bound);
      scale[mt] = __builtin_amdgcn_ldexpf(1.0f, kF16Top - e);
      inv[mt] = __builtin_amdgcn_ldexpf(inv_w2_scale, e - kF16Top);
    }
  };
  float xc[MT][kIn], xn[MT][kIn];          // observations of this tile / of the workgroup's next tile
  float sc[MT], inv_c[MT];                 // this tile's row factors (planes) and their inverses (accumulators)
  load_x(xc, blockIdx.x);
  load_x(xn, blockIdx.x + stride);
  row_scales(xc, sc, inv_c);

  // ---- W2 ring ------------------------------------------------------------------------------------------
  // chunk `ks` -> stage `stage`: sixteen one-KiB pieces, four per wave
  auto request_b = [&](int ks, int stage) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int piece = wave * 4 + u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(w2rsrc, smem + stage * kRowsChunk + piece * 1024, 16, lane * 16,
                                               ks * kRowsChunk + piece * 1024, 0, 0);
    }
  };
  const unsigned b_lane = lds0 + lane * 16;
  const unsigned c_lane = lds0 + kConstOff + hh * 32;  // this lane's eight k of a step: + 64 ks

  // The A fragments of chunk `ks` for rows xs: planes hi / lo, eight k each (k = 16 ks + 8 hh + e).
  struct Consts {
    u32x4 b1q[2], w1q[kIn][2];
  };
  static_assert(kIn <= 5, "the constants' reads are written out for five inputs");
  // element e (0..7) of the fragment of m-tile mt
  auto h1_element = [&](const Consts &c, const float (&xs)[MT][kIn], int mt, int e) {
    float v = __uint_as_float(c.b1q[e >> 2][e & 3]);
#pragma unroll
    for (int i = 0; i < kIn; ++i) v = __builtin_fmaf(xs[mt][i], __uint_as_float(c.w1q[i][e >> 2][e & 3]), v);
    return relu1(v);
  };

  f32x16 acc[MT][8];
  u32x4 a_hi[2][MT], a_lo[2][MT];  // A fragments: [set], current / next step
  u32x4 bh[4], bl[4];              // B fragments of column block nt live in set nt % 4, requested two blocks ahead
  Consts c;                        // layer-1 constants of the chunk being produced: elements 0..3 | 4..7 re-requested in turn
  constexpr int kHalf = 1 + kIn;   // LDS reads per half of the constants

  auto request_consts_half = [&](int ks, auto half_tag) {
    constexpr int Q = decltype(half_tag)::value;
    const unsigned a = c_lane + ks * 64;
    c.b1q[Q] = lds_read_b128<16 * Q>(a);
#pragma unroll
    for (int i = 0; i < kIn; ++i)
      c.w1q[i][Q] = i == 0   ? lds_read_b128<1 * 1024 + 16 * Q>(a)
                    : i == 1 ? lds_read_b128<2 * 1024 + 16 * Q>(a)
                    : i == 2 ? lds_read_b128<3 * 1024 + 16 * Q>(a)
                    : i == 3 ? lds_read_b128<4 * 1024 + 16 * Q>(a)
                             : lds_read_b128<5 * 1024 + 16 * Q>(a);
  };
  auto tie_consts_half = [&](auto half_tag) {  // (behind a counted wait that covers them)
    constexpr int Q = decltype(half_tag)::value;
#pragma unroll
    for (int i = 0; i < kIn; ++i) {
      u32x4 &w = c.w1q[i][Q];  // (named outside the asm: operands alone do not capture in a generic lambda)
      asm volatile("" : "+v"(w));
    }
    u32x4 &b = c.b1q[Q];
    asm volatile("" : "+v"(b));
  };
  using H0 = std::integral_constant<int, 0>;
  using H1 = std::integral_constant<int, 1>;
  // planes of column block nt (0..7) of the chunk in the stage at `br`
  auto request_block = [&](unsigned br, int nt, int set) {
    bh[set] = nt == 0   ? lds_read_b128<0 * 1024>(br)
              : nt == 1 ? lds_read_b128<2 * 1024>(br)
              : nt == 2 ? lds_read_b128<4 * 1024>(br)
              : nt == 3 ? lds_read_b128<6 * 1024>(br)
              : nt == 4 ? lds_read_b128<8 * 1024>(br)
              : nt == 5 ? lds_read_b128<10 * 1024>(br)
              : nt == 6 ? lds_read_b128<12 * 1024>(br)
                        : lds_read_b128<14 * 1024>(br);
    bl[set] = nt == 0   ? lds_read_b128<1 * 1024>(br)
              : nt == 1 ? lds_read_b128<3 * 1024>(br)
              : nt == 2 ? lds_read_b128<5 * 1024>(br)
              : nt == 3 ? lds_read_b128<7 * 1024>(br)
              : nt == 4 ? lds_read_b128<9 * 1024>(br)
              : nt == 5 ? lds_read_b128<11 * 1024>(br)
              : nt == 6 ? lds_read_b128<13 * 1024>(br)
                        : lds_read_b128<15 * 1024>(br);
  };

  // One k-step = eight blocks (column block nt: 3 MT products + one element of the next step's fragments).
  // The stream of LDS reads never stops at a step boundary: blocks 6 / 7 request blocks 0 / 1 of the NEXT chunk,
  // which is why the step's barrier sits in the MIDDLE of the step before (behind block 3): it publishes chunk
  // s + 1 (loads issued two steps ago: counted vmcnt) and frees the stage of chunk s - 1, which blocks 4..7 then
  // refill, one direct-to-LDS piece each, with chunk s + RING - 1.
  // CUR: the A set consumed; the other set receives chunk s + 1.  LAST (step 15): nothing is produced or requested
  // for the next step -- the epilogue lies in between and needs the registers; open_tile() restarts the stream.
  auto do_step = [&](auto first_tag, auto cur_tag, auto last_tag, int s, int stage) {
    constexpr bool FIRST = decltype(first_tag)::value;
    constexpr int CUR = decltype(cur_tag)::value;
    constexpr bool LAST = decltype(last_tag)::value;
    const int stage_next = stage + 1 == RING ? 0 : stage + 1, stage_free = stage == 0 ? RING - 1 : stage - 1;
    const unsigned br = b_lane + stage * kRowsChunk, br_next = b_lane + stage_next * kRowsChunk;
    [[maybe_unused]] float h[MT][8];
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) {
      const int set = nt % 4, ahead = (nt + 2) % 4;
      // requests: constants first (they are older than the block pair, hence covered by later waits), then the
      // block two ahead
      if (!LAST && nt == 1) request_consts_half((s + 1) & (kSplitSteps - 1), H1{});   // elements 4..7 of THIS step's production
      if (!LAST && nt == 5) request_consts_half((s + 2) & (kSplitSteps - 1), H0{});   // elements 0..3 of the NEXT step's
      if constexpr ((DIAG & 8) != 0) {
        if (nt == 0) { request_block(br, 2, 2); request_block(br, 3, 3); }
        if (nt == 6 && !LAST) { request_block(br_next, 0, 0); request_block(br_next, 1, 1); }
      } else {
        if (nt < 6) request_block(br, nt + 2, ahead);
        else if (!LAST) request_block(br_next, nt - 6, ahead);
      }
      // in-order LDS returns: everything but the younger block pairs (and, in the block that requested them and
      // the one behind it, the constants) has landed
      if constexpr ((DIAG & 8) != 0) {
        wait_lds<0>(bh[set], bl[set]);
        if (!LAST && (nt == 0 || nt == 7)) tie_consts_half(H0{});
        if (!LAST && nt == 3) tie_consts_half(H1{});
      } else if constexpr (LAST) {
        if (nt < 6) wait_lds<4>(bh[set], bl[set]);
        else if (nt == 6) wait_lds<2>(bh[set], bl[set]);
        else wait_lds<0>(bh[set], bl[set]);
      } else {
        if (nt == 1 || nt == 2 || nt == 5 || nt == 6) wait_lds<4 + kHalf>(bh[set], bl[set]);
        else wait_lds<4>(bh[set], bl[set]);
        if (nt == 0 || nt == 7) tie_consts_half(H0{});  // (landed: requested in block 5 of the step before / by open_tile)
        if (nt == 3) tie_consts_half(H1{});
      }
      const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, bh[set]),
                                                             __builtin_bit_cast(half8, a_lo[CUR][mt]),
                                                             FIRST ? zero : acc[mt][nt], 0, 0, 0);
        if constexpr ((DIAG & 4) != 0) {  // (the other operands stay live)
          asm volatile("" ::"v"(bl[set]), "v"(a_hi[CUR][mt]));
          continue;
        }
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, bl[set]),
                                                             __builtin_bit_cast(half8, a_hi[CUR][mt]), acc[mt][nt], 0, 0, 0);
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, bh[set]),
                                                             __builtin_bit_cast(half8, a_hi[CUR][mt]), acc[mt][nt], 0, 0, 0);
      }
      // one element of the next step's fragments per block: they issue beside the block's products
      if constexpr (!LAST && !(DIAG & 1)) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          h[mt][nt] = h1_element(c, xc, mt, nt);
          if (nt & 1) {
            uint32_t hi, lo;
            f16_pair_scaled(h[mt][nt - 1], h[mt][nt], sc[mt], hi, lo);
            a_hi[CUR ^ 1][mt][nt >> 1] = hi;
            a_lo[CUR ^ 1][mt][nt >> 1] = lo;
          }
        }
      }
      if constexpr (!LAST && (DIAG & 1) != 0) {
        if (nt == 7) {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            a_hi[CUR ^ 1][mt] = a_hi[CUR][mt];
            a_lo[CUR ^ 1][mt] = a_lo[CUR][mt];
          }
        }
      }
      if (nt == 3) {
        // chunk s + 1 has landed for every wave (vector memory returns in order: all but the pieces of the
        // kAhead - 2 chunks requested since -- anything else in flight only makes this wait longer), and every
        // wave is past step s - 1
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(4 * (kAhead - 2)) : "memory");
      }
      if (nt >= 4) {
        const int piece = wave * 4 + (nt - 4);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(w2rsrc, smem + stage_free * kRowsChunk + piece * 1024, 16, lane * 16,
                                                 ((s + kAhead) & (kSplitSteps - 1)) * kRowsChunk + piece * 1024, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // Start of a tile's stream: the fragments of chunk 0 from xc / sc into set 0, then what step 0 expects to be on
  // its way -- constants 0..3 of chunk 1, column blocks 0 and 1 of chunk 0 (in the stage `stage`, published by the
  // barrier of the step before, or the prologue's).
  auto open_tile = [&](int stage) {
    request_consts_half(0, H0{});
    request_consts_half(0, H1{});
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    tie_consts_half(H0{});
    tie_consts_half(H1{});
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int e = 0; e < 8; e += 2) {
        uint32_t hi, lo;
        f16_pair_scaled(h1_element(c, xc, mt, e), h1_element(c, xc, mt, e + 1), sc[mt], hi, lo);
        a_hi[0][mt][e >> 1] = hi;
        a_lo[0][mt][e >> 1] = lo;
      }
    request_consts_half(1, H0{});
    const unsigned br = b_lane + stage * kRowsChunk;
    request_block(br, 0, 0);
    request_block(br, 1, 1);
  };

  using T = std::true_type;
  using F = std::false_type;
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  static_assert(kAhead >= 2, "the mid-step barrier publishes a chunk requested at least a step earlier");

  // ---- prologue: the first kAhead chunks on their way; chunk 0 readable by every wave ------------------------------
  __syncthreads();  // (the constants are in LDS for every wave; also orders the reads of the bound loops above)
#pragma unroll
  for (int d = 0; d < kAhead; ++d) request_b(d, d);
  asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(4 * (kAhead - 1)) : "memory");

  int stage = 0;
  auto next_stage = [&]() { stage = stage + 1 == RING ? 0 : stage + 1; };
  for (int64_t tile = blockIdx.x; tile < tiles; tile += stride) {
    const int64_t r0 = tile * kTile;
    open_tile(stage);
    do_step(T{}, S0{}, F{}, 0, stage);
    next_stage();
    do_step(F{}, S1{}, F{}, 1, stage);
    next_stage();
#pragma unroll 1
    for (int s = 2; s < kSplitSteps - 2; s += 2) {
      do_step(F{}, S0{}, F{}, s, stage);
      next_stage();
      do_step(F{}, S1{}, F{}, s + 1, stage);
      next_stage();
    }
    do_step(F{}, S0{}, F{}, kSplitSteps - 2, stage);
    next_stage();
    do_step(F{}, S1{}, T{}, kSplitSteps - 1, stage);
    next_stage();

    // ---- epilogue, per wave: this lane holds rows 32 (MT wave + mt) + l32 and, of column block nt, the
    // sixteen columns 32 nt + 8 g + 4 hh + e (register 4 g + e) ------------------------------------------------
    const int wrow0 = 32 * MT * wave;  // first row of this wave in the tile
    const unsigned constp = lds0 + kB2Off + (4 * hh) * 4;
    constexpr int kChains = kOut <= 2 ? 4 : 2;
    float part[MT][kOut][kChains];
    [[maybe_unused]] uint32_t gate_words[MT][8];
    [[maybe_unused]] const unsigned t_base = lds0 + kScratchOff + wave * (32 * MT * kRowsPitch);
    [[maybe_unused]] const unsigned t_write = t_base + l32 * kRowsPitch + 16 * hh;
    [[maybe_unused]] const unsigned t_read = t_base + (lane >> 3) * kRowsPitch + (lane & 7) * 16;
    const int64_t rows_left = m - r0 - wrow0;  // rows of this wave that exist
    const int wrows = rows_left <= 0 ? 0 : rows_left < 32 * MT ? (int)rows_left : 32 * MT;
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t h2rsrc =
        buffer_rsrc(kStore ? save_h2 + (r0 + wrow0) * kHidden : nullptr, wrows * kHidden * 4);
    [[maybe_unused]] const int h2_voff = ((lane >> 3) * kHidden + 4 * (lane & 7)) * 4;
    constexpr int kGroup = kOut >= 4 ? 1 : kOut >= 2 ? 2 : 4;  // quads (g) per pipeline stage
    constexpr int kStages = 8 * (4 / kGroup);
    u32x4 bq[kGroup], wq[kOut][kGroup];
    auto request_b2 = [&](int st) {
      const int nt = st / (4 / kGroup), g0 = (st % (4 / kGroup)) * kGroup;
#pragma unroll
      for (int gi = 0; gi < kGroup; ++gi) bq[gi] = lds_read_b128<0>(constp + (32 * nt + 8 * (g0 + gi)) * 4);
    };
    auto request_w3 = [&](int st) {
      const int nt = st / (4 / kGroup), g0 = (st % (4 / kGroup)) * kGroup;
#pragma unroll
      for (int gi = 0; gi < kGroup; ++gi) {
        const unsigned a = constp + (32 * nt + 8 * (g0 + gi)) * 4;
#pragma unroll
        for (int q = 0; q < kOut; ++q)
          wq[q][gi] = q == 0   ? lds_read_b128<1 * kHidden * 4>(a)
                      : q == 1 ? lds_read_b128<2 * kHidden * 4>(a)
                      : q == 2 ? lds_read_b128<3 * kHidden * 4>(a)
                               : lds_read_b128<4 * kHidden * 4>(a);
      }
    };
    static_assert(kOut <= 4, "request_w3 is written out for four outputs");
    if constexpr ((DIAG & 2) != 0) {  // tuning builds: the accumulators are consumed, nothing else
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int q = 0; q < kOut; ++q)
#pragma unroll
          for (int j = 0; j < kChains; ++j) part[mt][q][j] = 0.0f;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) {
          part[mt][0][0] += acc[mt][nt][0] + acc[mt][nt][15];  // (every product stays live)
          if constexpr (SAVE != 0) gate_words[mt][nt] = 0;
        }
    }
    if constexpr ((DIAG & 2) == 0) {
    request_b2(0);
    request_w3(0);
    }
    [[maybe_unused]] u32x4 t_rows[4 * MT];
    constexpr int kRunStages = (DIAG & 2) ? 0 : kStages;
#pragma unroll
    for (int st = 0; st < kRunStages; ++st) {
      const int nt = st / (4 / kGroup), g0 = (st % (4 / kGroup)) * kGroup;
      const bool last_of_block = g0 + kGroup == 4;
#pragma unroll
      for (int gi = 0; gi < kGroup; ++gi) wait_lds<0>(bq[gi]);
#pragma unroll
      for (int q = 0; q < kOut; ++q)
#pragma unroll
        for (int gi = 0; gi < kGroup; ++gi) wait_lds<0>(wq[q][gi]);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int gi = 0; gi < kGroup; ++gi) {
          const int g = g0 + gi;
#pragma unroll
          for (int e = 0; e < 4; ++e)
            acc[mt][nt][4 * g + e] = relu1(__builtin_fmaf(acc[mt][nt][4 * g + e], inv_c[mt], __uint_as_float(bq[gi][e])));
        }
      if (st + 1 < kStages) request_b2(st + 1);
      if constexpr (kStore) {
        // block [32 MT rows][32 columns] -> the wave's transpose scratch, back as eight lanes per row: a
        // store instruction is then eight full 128-byte lines
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int gi = 0; gi < kGroup; ++gi) {
            const int g = g0 + gi;
            const f32x4 v = {acc[mt][nt][4 * g], acc[mt][nt][4 * g + 1], acc[mt][nt][4 * g + 2], acc[mt][nt][4 * g + 3]};
            const u32x4 u = __builtin_bit_cast(u32x4, v);
            if (mt == 0) {
              g == 0 ? lds_write_b128<0>(t_write, u) : g == 1 ? lds_write_b128<32>(t_write, u)
              : g == 2 ? lds_write_b128<64>(t_write, u) : lds_write_b128<96>(t_write, u);
            } else {
              g == 0 ? lds_write_b128<32 * kRowsPitch>(t_write, u) : g == 1 ? lds_write_b128<32 * kRowsPitch + 32>(t_write, u)
              : g == 2 ? lds_write_b128<32 * kRowsPitch + 64>(t_write, u) : lds_write_b128<32 * kRowsPitch + 96>(t_write, u);
            }
          }
        if (last_of_block) {
          t_rows[0] = lds_read_b128<0 * 8 * kRowsPitch>(t_read);
          t_rows[1] = lds_read_b128<1 * 8 * kRowsPitch>(t_read);
          t_rows[2] = lds_read_b128<2 * 8 * kRowsPitch>(t_read);
          t_rows[3] = lds_read_b128<3 * 8 * kRowsPitch>(t_read);
          if constexpr (MT == 2) {
            t_rows[4 * (MT - 1) + 0] = lds_read_b128<4 * 8 * kRowsPitch>(t_read);
            t_rows[4 * (MT - 1) + 1] = lds_read_b128<5 * 8 * kRowsPitch>(t_read);
            t_rows[4 * (MT - 1) + 2] = lds_read_b128<6 * 8 * kRowsPitch>(t_read);
            t_rows[4 * (MT - 1) + 3] = lds_read_b128<7 * 8 * kRowsPitch>(t_read);
          }
        }
      }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int gi = 0; gi < kGroup; ++gi) {
          const int g = g0 + gi;
          if constexpr (SAVE != 0) {
            // gate of h2: bit c of word nt of a row <=> column 32 nt + c > 0; h2 >= +0, so "h2 > 0" is bit 31 of
            // (bits + 0x7fffffff); this lane's nibble for quad g sits at bit 8 g + 4 hh
            uint32_t nib = 0;
#pragma unroll
            for (int e = 3; e >= 0; --e)
              nib = __builtin_amdgcn_alignbit(nib, __float_as_uint(acc[mt][nt][4 * g + e]) + 0x7fffffffu, 31);
            gate_words[mt][nt] = (g == 0 ? 0u : gate_words[mt][nt]) | (nib << (8 * g + 4 * hh));
          }
#pragma unroll
          for (int q = 0; q < kOut; ++q) {
            float p = (nt == 0 && g < kChains) ? 0.0f : part[mt][q][g % kChains];
#pragma unroll
            for (int e = 0; e < 4; ++e) p = __builtin_fmaf(acc[mt][nt][4 * g + e], __uint_as_float(wq[q][gi][e]), p);
            part[mt][q][g % kChains] = p;
          }
        }
      if (st + 1 < kStages) request_w3(st + 1);
      if constexpr (kStore) {
        if (last_of_block) {
          constexpr int kNewer = kGroup * kOut;  // the next stage's W3 quads, requested behind the block's reads
#pragma unroll
          for (int i = 0; i < 4 * MT; i += 4) {
            if (st + 1 < kStages) wait_lds<kNewer>(t_rows[i], t_rows[i + 1], t_rows[i + 2], t_rows[i + 3]);
            else wait_lds<0>(t_rows[i], t_rows[i + 1], t_rows[i + 2], t_rows[i + 3]);
          }
#pragma unroll
          for (int i = 0; i < 4 * MT; ++i)  // rows 8 i + (lane >> 3), columns 32 nt + 4 (lane & 7) .. + 3
            __builtin_amdgcn_raw_buffer_store_b128(t_rows[i], h2rsrc, h2_voff + (8 * i * kHidden + 32 * nt) * 4, 0, RL8_H2_STORE_AUX);
        }
      }
    }
    // the two half-waves hold the two halves of every row's columns: one exchange completes a row
    float total[MT][kOut];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int q = 0; q < kOut; ++q) {
        float p = part[mt][q][0] + part[mt][q][1];
        if constexpr (kChains == 4) p += part[mt][q][2] + part[mt][q][3];
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(p), __float_as_uint(p), false, false);
        total[mt][q] = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
      }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int row = 32 * mt + l32;  // within the wave
      if constexpr (SAVE != 0) {
        if (save_gate2 != nullptr) {
          // a row's eight words: own nibbles | the other half-wave's; half-wave hh stores words 4 hh .. 4 hh + 3
          u32x4 words;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const auto lo = __builtin_amdgcn_permlane32_swap(gate_words[mt][j], gate_words[mt][j], false, false);
            const auto hi = __builtin_amdgcn_permlane32_swap(gate_words[mt][4 + j], gate_words[mt][4 + j], false, false);
            const uint32_t wl = lo[0] | lo[1], wh = hi[0] | hi[1];
            words[j] = hh ? wh : wl;
          }
          if (row < wrows) {
            uint32_t *dst = save_gate2 + (r0 + wrow0) * 8;
            *reinterpret_cast<u32x4 *>(dst + (unsigned)(row * 8 + 4 * hh)) = words;
          }
        }
      }
      if (hh == 0 && row < wrows) {
        float *dst = out + (r0 + wrow0) * n_out;
#pragma unroll
        for (int q = 0; q < kOut; ++q)
          if (q < n_out) dst[(unsigned)(row * n_out + q)] = total[mt][q] + b3[q];
      }
    }
    // the workgroup's next tile
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int i = 0; i < kIn; ++i) xc[mt][i] = xn[mt][i];
    row_scales(xc, sc, inv_c);
    load_x(xn, tile + 2 * stride);
  }
  // (the ring's last requests and the reads for a next step ran past the last tile: nothing may land in LDS or in
  // registers after the wave is gone)
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#ifdef RL8_ROWS_STAMP
  if (stamps != nullptr && lane == 0) {  // (a buffer of their own: no output depends on the stamps)
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    unsigned long long *dst = stamps + ((size_t)blockIdx.x * 4 + wave) * 4;
    dst[0] = t1 - stamp_t0;
    dst[1] = r1 - stamp_r0;
    dst[2] = (unsigned long long)((tiles - blockIdx.x + stride - 1) / stride);  // tiles this workgroup ran
    dst[3] = __builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20);  // HW_REG_XCC_ID
  }
#endif
}

// ---- the same kernel on v_mfma_f32_16x16x32_f16 ----------------------------------------------------------------
// Why a second shape: these kernels run against the chip's power limit, and at equal work the 16x16x32 form holds
// a higher clock (tools/probes/mfma_shape_probe.hip on random data, every operand re-read from LDS: 1 677-1 687
// TFLOP/s at 1 824 MHz against 1 483-1 520 at 1 714-1 726 for 32x32x16, same cycles per flop).
// What changes: a fragment is 16 rows x 32 k (lane = row mod 16, k-block = lane / 16), so
//   * K runs in eight blocks of 32; a block's W2 planes are two ring chunks of 16 KiB (column tiles 0..7 | 8..15:
//     rl8_mlp_pack_w2_f16 layout 1), i.e. sixteen HALF-steps per tile with exactly the ring, barrier and request
//     pattern of the kernel above;
//   * a wave's 32 rows are two row tiles; their A fragments (hi / lo planes) serve both half-steps of a block, and
//     half-step c produces row tile c of the NEXT block (one element per column-tile slot, as before);
//   * accumulators: 2 x 16 tiles of 4 registers; transposed product (first operand = W2 fragment), so a lane holds,
//     for row (lane & 15) + 16 rt and column tile ct, the four consecutive columns 16 ct + 4 (lane >> 4) + r: quads
//     again, a row being shared by the four lanes of equal lane & 15 (two swaps finish a row's sums).
template <int DIN, int NOUT, int SAVE, int RING>
__global__ __launch_bounds__(kBlock, 2) void mlp_rows16_forward_kernel(
    const float *__restrict__ x, int64_t m, const float *__restrict__ w1, const float *__restrict__ b1,
    const void *__restrict__ w2s, const float *__restrict__ b2, const float *__restrict__ w3,
    const float *__restrict__ b3, float *__restrict__ out, float *__restrict__ save_h2,
    uint32_t *__restrict__ save_gate2) {
  constexpr int kIn = DIN, d_in = DIN, n_out = NOUT;
  constexpr int kOut = pad_out(NOUT);
  constexpr int kTile = 128;
  constexpr int kAhead = RING - 1;
  constexpr bool kStore = SAVE == 1;
  constexpr int kHalfSteps = 16;  // (k block S = hs / 2, column half c = hs % 2)
  static_assert(rows_lds_bytes(RING, kIn, kOut, 1, kStore) <= 80 * 1024, "two workgroups per CU");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const unsigned lds0 = lds_offset(smem);
  constexpr int kConstOff = RING * kRowsChunk;
  constexpr int kB2Off = kConstOff + (1 + kIn) * kHidden * 4;
  constexpr int kScratchOff = kConstOff + rows_consts_bytes(kIn, kOut);
  const int tid = threadIdx.x, lane = tid & 63, l16 = lane & 15, kq = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const __amdgpu_buffer_rsrc_t w2rsrc = buffer_rsrc(w2s, kRowsPacked);
  const float inv_w2_scale = reinterpret_cast<const float *>(static_cast<const unsigned char *>(w2s) + kRowsPacked)[1];

  const int64_t tiles = (m + kTile - 1) / kTile;
  const int64_t stride = gridDim.x;
  if ((int64_t)blockIdx.x >= tiles) return;

  float b1max = 0.0f, w1max[kIn];
  {
    float *consts = reinterpret_cast<float *>(smem + kConstOff);
    consts[tid] = b1[tid];
#pragma unroll
    for (int i = 0; i < kIn; ++i) consts[(1 + i) * kHidden + tid] = w1[tid * d_in + i];
    consts[(1 + kIn) * kHidden + tid] = b2[tid];
#pragma unroll
    for (int q = 0; q < kOut; ++q) consts[(2 + kIn + q) * kHidden + tid] = q < n_out ? w3[q * kHidden + tid] : 0.0f;
    __syncthreads();
    for (int k = 0; k < kHidden; ++k) b1max = __builtin_fmaxf(b1max, __builtin_fabsf(consts[k]));
    b1max = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(b1max)));
#pragma unroll
    for (int i = 0; i < kIn; ++i) {
      float mx = 0.0f;
      for (int k = 0; k < kHidden; ++k) mx = __builtin_fmaxf(mx, __builtin_fabsf(consts[(1 + i) * kHidden + k]));
      w1max[i] = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(mx)));
    }
  }

  // rows of this lane: tile base + 32 wave + 16 rt + l16 (the four lanes of equal l16 hold the same rows)
  auto load_x = [&](float (&dst)[2][kIn], int64_t tile) {
    const int64_t r0 = tile * kTile;
    const float *base = x + r0 * d_in;
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      const int row = 32 * wave + 16 * rt + l16;
#pragma unroll
      for (int i = 0; i < kIn; ++i) dst[rt][i] = (tile < tiles && r0 + row < m) ? base[(unsigned)(row * d_in + i)] : 0.0f;
    }
  };
  auto row_scales = [&](const float (&xs)[2][kIn], float (&scale)[2], float (&inv)[2]) {
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      float bound = b1max;
#pragma unroll
      for (int i = 0; i < kIn; ++i) bound = __builtin_fmaf(__builtin_fabsf(xs[rt][i]), w1max[i], bound);
      const int e = f16_bound_exponent(bound);
      scale[rt] = __builtin_amdgcn_ldexpf(1.0f, kF16Top - e);
      inv[rt] = __builtin_amdgcn_ldexpf(inv_w2_scale, e - kF16Top);
    }
  };
  float xc[2][kIn], xn[2][kIn];
  float sc[2], inv_c[2];
  load_x(xc, blockIdx.x);
  load_x(xn, blockIdx.x + stride);
  row_scales(xc, sc, inv_c);

  auto request_b = [&](int hs, int stage) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int piece = wave * 4 + u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(w2rsrc, smem + stage * kRowsChunk + piece * 1024, 16, lane * 16,
                                               hs * kRowsChunk + piece * 1024, 0, 0);
    }
  };
  const unsigned b_lane = lds0 + lane * 16;
  const unsigned c_lane = lds0 + kConstOff + kq * 32;  // this lane's eight k of a block: + 128 S

  struct Consts {
    u32x4 b1q[2], w1q[kIn][2];
  };
  static_assert(kIn <= 5, "the constants' reads are written out for five inputs");
  auto h1_element = [&](const Consts &cc, const float (&xs)[2][kIn], int rt, int e) {
    float v = __uint_as_float(cc.b1q[e >> 2][e & 3]);
#pragma unroll
    for (int i = 0; i < kIn; ++i) v = __builtin_fmaf(xs[rt][i], __uint_as_float(cc.w1q[i][e >> 2][e & 3]), v);
    return relu1(v);
  };

  f32x4 acc[2][16];
  u32x4 a_hi[2][2], a_lo[2][2];    // A fragments: [set = k block parity][row tile]
  u32x4 bh[4], bl[4];
  Consts c;
  constexpr int kHalf = 1 + kIn;

  auto request_consts_half = [&](int S, auto half_tag) {  // constants of k block S, elements 4 Q .. 4 Q + 3
    constexpr int Q = decltype(half_tag)::value;
    const unsigned a = c_lane + S * 128;
    c.b1q[Q] = lds_read_b128<16 * Q>(a);
#pragma unroll
    for (int i = 0; i < kIn; ++i)
      c.w1q[i][Q] = i == 0   ? lds_read_b128<1 * 1024 + 16 * Q>(a)
                    : i == 1 ? lds_read_b128<2 * 1024 + 16 * Q>(a)
                    : i == 2 ? lds_read_b128<3 * 1024 + 16 * Q>(a)
                    : i == 3 ? lds_read_b128<4 * 1024 + 16 * Q>(a)
                             : lds_read_b128<5 * 1024 + 16 * Q>(a);
  };
  auto tie_consts_half = [&](auto half_tag) {
    constexpr int Q = decltype(half_tag)::value;
#pragma unroll
    for (int i = 0; i < kIn; ++i) {
      u32x4 &w = c.w1q[i][Q];
      asm volatile("" : "+v"(w));
    }
    u32x4 &b = c.b1q[Q];
    asm volatile("" : "+v"(b));
  };
  using H0 = std::integral_constant<int, 0>;
  using H1 = std::integral_constant<int, 1>;
  auto request_block = [&](unsigned br, int ctl, int set) {  // planes of local column tile ctl (0..7) of the chunk at `br`
    bh[set] = ctl == 0   ? lds_read_b128<0 * 1024>(br)
              : ctl == 1 ? lds_read_b128<2 * 1024>(br)
              : ctl == 2 ? lds_read_b128<4 * 1024>(br)
              : ctl == 3 ? lds_read_b128<6 * 1024>(br)
              : ctl == 4 ? lds_read_b128<8 * 1024>(br)
              : ctl == 5 ? lds_read_b128<10 * 1024>(br)
              : ctl == 6 ? lds_read_b128<12 * 1024>(br)
                         : lds_read_b128<14 * 1024>(br);
    bl[set] = ctl == 0   ? lds_read_b128<1 * 1024>(br)
              : ctl == 1 ? lds_read_b128<3 * 1024>(br)
              : ctl == 2 ? lds_read_b128<5 * 1024>(br)
              : ctl == 3 ? lds_read_b128<7 * 1024>(br)
              : ctl == 4 ? lds_read_b128<9 * 1024>(br)
              : ctl == 5 ? lds_read_b128<11 * 1024>(br)
              : ctl == 6 ? lds_read_b128<13 * 1024>(br)
                         : lds_read_b128<15 * 1024>(br);
  };

  // One half-step: column tiles 8 C .. 8 C + 7 of k block S (chunk hs = 2 S + C).  CUR: the A set of block S; half-step
  // C produces row tile C of block S + 1 into the other set.  KIND 0: any half-step up to (6, 0); 1: (6, 1) -- the
  // half-step behind it produces nothing, so no constants are requested for it; 2: (7, 0), nothing produced; 3: (7, 1),
  // the tile's last: nothing produced and nothing requested for a next half-step (the epilogue lies in between).
  // Everything else -- requests two slots ahead, the barrier behind slot 3, the ring -- as in the 32x32x16 kernel.
  auto do_half = [&](auto first_tag, auto cur_tag, auto c_tag, auto kind_tag, int hs, int stage) {
    constexpr bool FIRST = decltype(first_tag)::value;
    constexpr int CUR = decltype(cur_tag)::value;
    constexpr int C = decltype(c_tag)::value;
    constexpr int KIND = decltype(kind_tag)::value;
    constexpr bool kProduce = KIND <= 1, kNextProduces = KIND == 0, kEnd = KIND == 3;
    const int stage_next = stage + 1 == RING ? 0 : stage + 1, stage_free = stage == 0 ? RING - 1 : stage - 1;
    const unsigned br = b_lane + stage * kRowsChunk, br_next = b_lane + stage_next * kRowsChunk;
    const int S = hs >> 1;
    [[maybe_unused]] float h[8];
#pragma unroll
    for (int sl = 0; sl < 8; ++sl) {
      const int set = sl % 4, ahead = (sl + 2) % 4, ct = 8 * C + sl;
      // constants of the block being produced (S + 1): elements 4..7 for this half-step; elements 0..3 for the
      // half-step behind this one (which produces block S + 1 again after C == 0, block S + 2 after C == 1)
      if (kProduce && sl == 1) request_consts_half(S + 1, H1{});
      if (kNextProduces && sl == 5) request_consts_half(S + 1 + C, H0{});
      if (sl < 6) request_block(br, sl + 2, ahead);
      else if (!kEnd) request_block(br_next, sl - 6, ahead);
      if constexpr (kEnd) {
        if (sl < 6) wait_lds<4>(bh[set], bl[set]);
        else if (sl == 6) wait_lds<2>(bh[set], bl[set]);
        else wait_lds<0>(bh[set], bl[set]);
      } else {
        if ((kProduce && (sl == 1 || sl == 2)) || (kNextProduces && (sl == 5 || sl == 6))) wait_lds<4 + kHalf>(bh[set], bl[set]);
        else wait_lds<4>(bh[set], bl[set]);
      }
      if ((kProduce && sl == 0) || (kNextProduces && sl == 7)) tie_consts_half(H0{});
      if (kProduce && sl == 3) tie_consts_half(H1{});
      const f32x4 zero = {0, 0, 0, 0};
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, bh[set]),
                                                             __builtin_bit_cast(half8, a_lo[CUR][rt]),
                                                             FIRST ? zero : acc[rt][ct], 0, 0, 0);
        acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, bl[set]),
                                                             __builtin_bit_cast(half8, a_hi[CUR][rt]), acc[rt][ct], 0, 0, 0);
        acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, bh[set]),
                                                             __builtin_bit_cast(half8, a_hi[CUR][rt]), acc[rt][ct], 0, 0, 0);
      }
      if constexpr (kProduce) {  // element sl of row tile C of block S + 1
        h[sl] = h1_element(c, xc, C, sl);
        if (sl & 1) {
          uint32_t hi, lo;
          f16_pair_scaled(h[sl - 1], h[sl], sc[C], hi, lo);
          a_hi[CUR ^ 1][C][sl >> 1] = hi;
          a_lo[CUR ^ 1][C][sl >> 1] = lo;
        }
      }
      if (sl == 3) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(4 * (kAhead - 2)) : "memory");
      if (sl >= 4) {
        const int piece = wave * 4 + (sl - 4);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(w2rsrc, smem + stage_free * kRowsChunk + piece * 1024, 16, lane * 16,
                                                 ((hs + kAhead) & (kHalfSteps - 1)) * kRowsChunk + piece * 1024, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // Start of a tile's stream: both row tiles' fragments of block 0 into set 0; on their way: constants 0..3 of block 1,
  // column tiles 0 and 1 of chunk 0.
  auto open_tile = [&](int stage) {
    request_consts_half(0, H0{});
    request_consts_half(0, H1{});
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    tie_consts_half(H0{});
    tie_consts_half(H1{});
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int e = 0; e < 8; e += 2) {
        uint32_t hi, lo;
        f16_pair_scaled(h1_element(c, xc, rt, e), h1_element(c, xc, rt, e + 1), sc[rt], hi, lo);
        a_hi[0][rt][e >> 1] = hi;
        a_lo[0][rt][e >> 1] = lo;
      }
    request_consts_half(1, H0{});
    const unsigned br = b_lane + stage * kRowsChunk;
    request_block(br, 0, 0);
    request_block(br, 1, 1);
  };

  using T = std::true_type;
  using F = std::false_type;
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using K0 = I0;
  using K1 = I1;
  using K2 = std::integral_constant<int, 2>;
  using K3 = std::integral_constant<int, 3>;
  static_assert(kAhead >= 2, "the mid-step barrier publishes a chunk requested at least a step earlier");

  __syncthreads();
#pragma unroll
  for (int d = 0; d < kAhead; ++d) request_b(d, d);
  asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(4 * (kAhead - 1)) : "memory");

  int stage = 0;
  auto next_stage = [&]() { stage = stage + 1 == RING ? 0 : stage + 1; };
  for (int64_t tile = blockIdx.x; tile < tiles; tile += stride) {
    const int64_t r0 = tile * kTile;
    open_tile(stage);
    do_half(T{}, I0{}, I0{}, K0{}, 0, stage);  next_stage();
    do_half(T{}, I0{}, I1{}, K0{}, 1, stage);  next_stage();
    do_half(F{}, I1{}, I0{}, K0{}, 2, stage);  next_stage();
    do_half(F{}, I1{}, I1{}, K0{}, 3, stage);  next_stage();
#pragma unroll 1
    for (int hs = 4; hs < kHalfSteps - 4; hs += 4) {
      do_half(F{}, I0{}, I0{}, K0{}, hs, stage);      next_stage();
      do_half(F{}, I0{}, I1{}, K0{}, hs + 1, stage);  next_stage();
      do_half(F{}, I1{}, I0{}, K0{}, hs + 2, stage);  next_stage();
      do_half(F{}, I1{}, I1{}, K0{}, hs + 3, stage);  next_stage();
    }
    do_half(F{}, I0{}, I0{}, K0{}, kHalfSteps - 4, stage);  next_stage();
    do_half(F{}, I0{}, I1{}, K1{}, kHalfSteps - 3, stage);  next_stage();
    do_half(F{}, I1{}, I0{}, K2{}, kHalfSteps - 2, stage);  next_stage();
    do_half(F{}, I1{}, I1{}, K3{}, kHalfSteps - 1, stage);  next_stage();

    // ---- epilogue, per wave: this lane holds rows 16 rt + l16 and, of column tile ct, columns 16 ct + 4 kq + r ---------
    const int wrow0 = 32 * wave;
    const unsigned constp = lds0 + kB2Off + (4 * kq) * 4;
    constexpr int kChains = kOut <= 2 ? 4 : 2;
    float part[2][kOut][kChains];
    [[maybe_unused]] uint32_t gate_words[2][8];
    [[maybe_unused]] const unsigned t_base = lds0 + kScratchOff + wave * (32 * kRowsPitch);
    [[maybe_unused]] const unsigned t_write = t_base + l16 * kRowsPitch + 16 * kq;
    [[maybe_unused]] const unsigned t_read = t_base + (lane >> 3) * kRowsPitch + (lane & 7) * 16;
    const int64_t rows_left = m - r0 - wrow0;
    const int wrows = rows_left <= 0 ? 0 : rows_left < 32 ? (int)rows_left : 32;
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t h2rsrc =
        buffer_rsrc(kStore ? save_h2 + (r0 + wrow0) * kHidden : nullptr, wrows * kHidden * 4);
    [[maybe_unused]] const int h2_voff = ((lane >> 3) * kHidden + 4 * (lane & 7)) * 4;
    constexpr int kGroup = kOut >= 4 ? 1 : kOut >= 2 ? 2 : 4;  // column tiles (one quad each) per pipeline stage
    constexpr int kStages = 16 / kGroup;
    u32x4 bq[kGroup], wq[kOut][kGroup];
    auto request_b2 = [&](int st) {
#pragma unroll
      for (int gi = 0; gi < kGroup; ++gi) bq[gi] = lds_read_b128<0>(constp + 16 * (st * kGroup + gi) * 4);
    };
    auto request_w3 = [&](int st) {
#pragma unroll
      for (int gi = 0; gi < kGroup; ++gi) {
        const unsigned a = constp + 16 * (st * kGroup + gi) * 4;
#pragma unroll
        for (int q = 0; q < kOut; ++q)
          wq[q][gi] = q == 0   ? lds_read_b128<1 * kHidden * 4>(a)
                      : q == 1 ? lds_read_b128<2 * kHidden * 4>(a)
                      : q == 2 ? lds_read_b128<3 * kHidden * 4>(a)
                               : lds_read_b128<4 * kHidden * 4>(a);
      }
    };
    static_assert(kOut <= 4, "request_w3 is written out for four outputs");
    request_b2(0);
    request_w3(0);
    [[maybe_unused]] u32x4 t_rows[4];
#pragma unroll
    for (int st = 0; st < kStages; ++st) {
      // the h2 transpose works on blocks of TWO column tiles (32 columns = one 128-byte line per row)
      const bool last_of_block = ((st + 1) * kGroup) % 2 == 0;
#pragma unroll
      for (int gi = 0; gi < kGroup; ++gi) wait_lds<0>(bq[gi]);
#pragma unroll
      for (int q = 0; q < kOut; ++q)
#pragma unroll
        for (int gi = 0; gi < kGroup; ++gi) wait_lds<0>(wq[q][gi]);
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int gi = 0; gi < kGroup; ++gi) {
          const int ct = st * kGroup + gi;
#pragma unroll
          for (int e = 0; e < 4; ++e)
            acc[rt][ct][e] = relu1(__builtin_fmaf(acc[rt][ct][e], inv_c[rt], __uint_as_float(bq[gi][e])));
        }
      if (st + 1 < kStages) request_b2(st + 1);
      if constexpr (kStore) {
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
          for (int gi = 0; gi < kGroup; ++gi) {
            const int ct = st * kGroup + gi;
            const u32x4 u = __builtin_bit_cast(u32x4, acc[rt][ct]);
            if (rt == 0) {
              (ct & 1) == 0 ? lds_write_b128<0>(t_write, u) : lds_write_b128<64>(t_write, u);
            } else {
              (ct & 1) == 0 ? lds_write_b128<16 * kRowsPitch>(t_write, u) : lds_write_b128<16 * kRowsPitch + 64>(t_write, u);
            }
          }
        if (last_of_block) {
          t_rows[0] = lds_read_b128<0 * 8 * kRowsPitch>(t_read);
          t_rows[1] = lds_read_b128<1 * 8 * kRowsPitch>(t_read);
          t_rows[2] = lds_read_b128<2 * 8 * kRowsPitch>(t_read);
          t_rows[3] = lds_read_b128<3 * 8 * kRowsPitch>(t_read);
        }
      }
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int gi = 0; gi < kGroup; ++gi) {
          const int ct = st * kGroup + gi;
          if constexpr (SAVE != 0) {
            // gate of h2: bit b of word w of a row <=> column 32 w + b > 0; this lane's nibble of column tile ct sits
            // in word ct / 2 at bit 16 (ct & 1) + 4 kq
            uint32_t nib = 0;
#pragma unroll
            for (int e = 3; e >= 0; --e) nib = __builtin_amdgcn_alignbit(nib, __float_as_uint(acc[rt][ct][e]) + 0x7fffffffu, 31);
            gate_words[rt][ct >> 1] = ((ct & 1) == 0 ? 0u : gate_words[rt][ct >> 1]) | (nib << (16 * (ct & 1) + 4 * kq));
          }
#pragma unroll
          for (int q = 0; q < kOut; ++q) {
            float p = ct < kChains ? 0.0f : part[rt][q][ct % kChains];
#pragma unroll
            for (int e = 0; e < 4; ++e) p = __builtin_fmaf(acc[rt][ct][e], __uint_as_float(wq[q][gi][e]), p);
            part[rt][q][ct % kChains] = p;
          }
        }
      if (st + 1 < kStages) request_w3(st + 1);
      if constexpr (kStore) {
        if (last_of_block) {
          constexpr int kNewer = kGroup * kOut;
          if (st + 1 < kStages) wait_lds<kNewer>(t_rows[0], t_rows[1], t_rows[2], t_rows[3]);
          else wait_lds<0>(t_rows[0], t_rows[1], t_rows[2], t_rows[3]);
          const int blk = (st * kGroup + kGroup - 1) >> 1;  // columns 32 blk ..
#pragma unroll
          for (int i = 0; i < 4; ++i)  // rows 8 i + (lane >> 3), columns 32 blk + 4 (lane & 7) .. + 3
            __builtin_amdgcn_raw_buffer_store_b128(t_rows[i], h2rsrc, h2_voff + (8 * i * kHidden + 32 * blk) * 4, 0, RL8_H2_STORE_AUX);
        }
      }
    }
    // a row's columns are spread over the four lanes of equal l16: two swaps complete its sums
    auto across_row = [&](uint32_t v, auto combine) {
      const auto s16 = __builtin_amdgcn_permlane16_swap(v, v, false, false);
      const uint32_t u = combine(s16[0], s16[1]);
      const auto s32 = __builtin_amdgcn_permlane32_swap(u, u, false, false);
      return combine(s32[0], s32[1]);
    };
    float total[2][kOut];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int q = 0; q < kOut; ++q) {
        float p = part[rt][q][0] + part[rt][q][1];
        if constexpr (kChains == 4) p += part[rt][q][2] + part[rt][q][3];
        total[rt][q] = __uint_as_float(across_row(__float_as_uint(p), [](uint32_t a, uint32_t b) {
          return __float_as_uint(__uint_as_float(a) + __uint_as_float(b));
        }));
      }
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      const int row = 16 * rt + l16;  // within the wave
      if constexpr (SAVE != 0) {
        if (save_gate2 != nullptr) {
          // full words = the four lanes' nibbles; lane kq stores words 2 kq, 2 kq + 1 of its row
          uint32_t full[8];
#pragma unroll
          for (int w = 0; w < 8; ++w) full[w] = across_row(gate_words[rt][w], [](uint32_t a, uint32_t b) { return a | b; });
          const uint32_t w0 = kq == 0 ? full[0] : kq == 1 ? full[2] : kq == 2 ? full[4] : full[6];
          const uint32_t w1v = kq == 0 ? full[1] : kq == 1 ? full[3] : kq == 2 ? full[5] : full[7];
          if (row < wrows) {
            uint32_t *dst = save_gate2 + (r0 + wrow0) * 8;
            *reinterpret_cast<u32x2 *>(dst + (unsigned)(row * 8 + 2 * kq)) = u32x2{w0, w1v};
          }
        }
      }
      if (kq == 0 && row < wrows) {
        float *dst = out + (r0 + wrow0) * n_out;
#pragma unroll
        for (int q = 0; q < kOut; ++q)
          if (q < n_out) dst[(unsigned)(row * n_out + q)] = total[rt][q] + b3[q];
      }
    }
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int i = 0; i < kIn; ++i) xc[rt][i] = xn[rt][i];
    row_scales(xc, sc, inv_c);
    load_x(xn, tile + 2 * stride);
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
}

template <int DIN, int NOUT, int SAVE>
static int launch_rows16_forward(hipStream_t s, const float *x, int64_t m, const float *w1, const float *b1, const void *w2s,
                                 const float *b2, const float *w3, const float *b3, float *out, float *h2, uint32_t *gate) {
  constexpr int kOut = pad_out(NOUT);
  constexpr int kRing = SAVE == 1 ? 3 : 4;
  constexpr int kLds = rows_lds_bytes(kRing, DIN, kOut, 1, SAVE == 1);
  auto kernel = &mlp_rows16_forward_kernel<DIN, NOUT, SAVE, kRing>;
  static bool attr_set = false;
  if (!attr_set) {
    attr_set = true;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipGetLastError();
  }
  const int64_t tiles = (m + 127) / 128;
  static const int cap = env_int("RL8_MLP_GRID_CAP");
  const int max_grid = cap > 0 ? cap : 2 * kCUs;
  const int grid = (int)(tiles < max_grid ? tiles : max_grid);
  kernel<<<grid, kBlock, kLds, s>>>(x, m, w1, b1, w2s, b2, w3, b3, out, h2, gate);
  return launch_status();
}

template <int DIN, int NOUT>
static int launch_rows16_forward_save(hipStream_t s, const float *x, int64_t m, const float *w1, const float *b1, const void *w2s,
                                      const float *b2, const float *w3, const float *b3, float *out, float *h2, uint32_t *gate) {
  return h2     ? launch_rows16_forward<DIN, NOUT, 1>(s, x, m, w1, b1, w2s, b2, w3, b3, out, h2, gate)
         : gate ? launch_rows16_forward<DIN, NOUT, 2>(s, x, m, w1, b1, w2s, b2, w3, b3, out, h2, gate)
                : launch_rows16_forward<DIN, NOUT, 0>(s, x, m, w1, b1, w2s, b2, w3, b3, out, h2, gate);
}

template <int DIN, int NOUT, int SAVE, int MT>
static int launch_rows_forward(hipStream_t s, const float *x, int64_t m, const float *w1, const float *b1, const void *w2s,
                               const float *b2, const float *w3, const float *b3, float *out, float *h2, uint32_t *gate) {
  constexpr int kOut = pad_out(NOUT);
  // two workgroups per CU: four chunks (three when the h2 transpose scratch is needed); one per CU: eight / six
  constexpr int kRing = MT == 1 ? (SAVE == 1 ? 3 : 4) : (SAVE == 1 ? 6 : 8);
  constexpr int kLds = rows_lds_bytes(kRing, DIN, kOut, MT, SAVE == 1);
  auto kernel = &mlp_rows_forward_kernel<DIN, NOUT, SAVE, MT, kRing>;
#ifdef RL8_ROWS_STAMP
  if constexpr (DIN == 1 && NOUT == 2 && SAVE == 0) {  // tuning builds: RL8_ROWS_DIAG selects a cut-down variant
    const int diag = env_int("RL8_ROWS_DIAG");
    kernel = diag == 1   ? &mlp_rows_forward_kernel<DIN, NOUT, SAVE, MT, kRing, 1>
             : diag == 2 ? &mlp_rows_forward_kernel<DIN, NOUT, SAVE, MT, kRing, 2>
             : diag == 3 ? &mlp_rows_forward_kernel<DIN, NOUT, SAVE, MT, kRing, 3>
             : diag == 4 ? &mlp_rows_forward_kernel<DIN, NOUT, SAVE, MT, kRing, 4>
             : diag == 7 ? &mlp_rows_forward_kernel<DIN, NOUT, SAVE, MT, kRing, 7>
             : diag == 8 ? &mlp_rows_forward_kernel<DIN, NOUT, SAVE, MT, kRing, 8>
             : diag == 11 ? &mlp_rows_forward_kernel<DIN, NOUT, SAVE, MT, kRing, 11>
             : diag == 15 ? &mlp_rows_forward_kernel<DIN, NOUT, SAVE, MT, kRing, 15>
                         : kernel;
  }
#endif
#ifdef RL8_ROWS_STAMP
  {
#else
  static bool attr_set = false;
  if (!attr_set) {
    attr_set = true;
#endif
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipGetLastError();
  }
  const int64_t tiles = (m + 128 * MT - 1) / (128 * MT);
  static const int cap = env_int("RL8_MLP_GRID_CAP");
  const int max_grid = cap > 0 ? cap : (MT == 1 ? 2 : 1) * kCUs;
  const int grid = (int)(tiles < max_grid ? tiles : max_grid);
#ifdef RL8_ROWS_STAMP
  const char *sp = getenv("RL8_ROWS_STAMP_PTR");
  kernel<<<grid, kBlock, kLds, s>>>(x, m, w1, b1, w2s, b2, w3, b3, out, h2, gate,
                                    sp ? reinterpret_cast<unsigned long long *>(strtoull(sp, nullptr, 0)) : nullptr);
#else
  kernel<<<grid, kBlock, kLds, s>>>(x, m, w1, b1, w2s, b2, w3, b3, out, h2, gate);
#endif
  return launch_status();
}

template <int DIN, int NOUT, int MT>
static int launch_rows_forward_save(hipStream_t s, const float *x, int64_t m, const float *w1, const float *b1, const void *w2s,
                                    const float *b2, const float *w3, const float *b3, float *out, float *h2, uint32_t *gate) {
  return h2     ? launch_rows_forward<DIN, NOUT, 1, MT>(s, x, m, w1, b1, w2s, b2, w3, b3, out, h2, gate)
         : gate ? launch_rows_forward<DIN, NOUT, 2, MT>(s, x, m, w1, b1, w2s, b2, w3, b3, out, h2, gate)
                : launch_rows_forward<DIN, NOUT, 0, MT>(s, x, m, w1, b1, w2s, b2, w3, b3, out, h2, gate);
}

// The rows-per-wave forward behind rl8_mlp_tower_forward_f16_f32 (mlp_f16_kernels.hip dispatches here):
// RL8_E* / launch status, or -1 when (mode, widths) has no compiled variant.
int mlp_rows_forward_dispatch(int mode, hipStream_t s, const float *x, int64_t m, int d_in, const float *w1, const float *b1,
                              const void *w2s, const float *b2, const float *w3, const float *b3, int n_out, float *out,
                              float *h2, uint32_t *gate) {
#define RL8_ROWS(D, N, MTV) \
  if (d_in == D && n_out == N && mode == MTV) return launch_rows_forward_save<D, N, MTV>(s, x, m, w1, b1, w2s, b2, w3, b3, out, h2, gate);
  RL8_ROWS(1, 1, 1) RL8_ROWS(1, 2, 1) RL8_ROWS(1, 3, 1)
  RL8_ROWS(2, 1, 1) RL8_ROWS(2, 2, 1) RL8_ROWS(2, 3, 1)
  RL8_ROWS(3, 1, 1) RL8_ROWS(3, 2, 1) RL8_ROWS(3, 3, 1)
  RL8_ROWS(5, 1, 1) RL8_ROWS(5, 2, 1) RL8_ROWS(5, 3, 1)
  RL8_ROWS(1, 1, 2) RL8_ROWS(1, 2, 2)
#undef RL8_ROWS
#define RL8_ROWS16(D, N) \
  if (d_in == D && n_out == N && mode == 16) return launch_rows16_forward_save<D, N>(s, x, m, w1, b1, w2s, b2, w3, b3, out, h2, gate);
  RL8_ROWS16(1, 1) RL8_ROWS16(1, 2) RL8_ROWS16(1, 3)
#undef RL8_ROWS16
  return -1;
}

}  // namespace rl8
