// N1, fourth generation of the tower forward (round 3): the fp16 two-plane scheme of
// mlp_f16_kernels.hip (three fp16 MFMAs per fp32 product, operands scaled by powers of two, fp32 accumulate) with
// the work laid out so that EVERY WAVE OWNS COMPLETE ROWS, on v_mfma_f32_16x16x32_f16.
//
//   * a wave computes 32 sample rows (two row tiles of 16) x all 256 output columns (16 column tiles);
//   * its A operand -- h1 = relu(b1 + x . w1) of those rows, scaled and split into two fp16 planes -- is produced by
//     the wave itself directly in MFMA fragment layout (lane = row mod 16, k-block = lane / 16: the lane computes the
//     eight k of its own fragment) and never leaves its registers: no LDS stage for A, no ds_write, no producer ->
//     consumer barrier, no LDS exchange of the row factors (the lane that scaled a row holds its accumulators);
//   * only W2's planes go through LDS, as a RING of 16-KiB chunks filled by direct-to-LDS loads two half-steps
//     ahead of their use; the one barrier per half-step only publishes chunks whose loads were issued long ago
//     (counted vmcnt) and sits in the MIDDLE of a half-step, so the stream of fragment reads never stops at a
//     boundary;
//   * the epilogue needs no workgroup barrier at all: head, gate bits and the h2 transpose are per wave.
//
// Why this MFMA shape.  These kernels run against the chip's power limit (tools/diag/rows_clock.py: with the matrix
// pipe 100 % busy on 32x32x16 the chip holds 1 675 MHz), and at equal work the 16x16x32 form holds a higher clock
// (tools/probes/mfma_shape_probe.hip, random data, every operand re-read from LDS: 1 677-1 687 TFLOP/s at 1 824 MHz
// against 1 483-1 520 at 1 714-1 726, same cycles per flop).  Measured on this kernel, per 2^20 rows (inference):
// previous generation 383 us, this structure on 32x32x16 358 us (MT = 2, 256 accumulator registers at one wave per
// SIMD: 361), on 16x16x32 337 us.
//
// K runs in eight blocks of 32; a block's W2 planes are two chunks (column tiles 0..7 | 8..15), i.e. sixteen
// HALF-steps per tile: half-step (S, C) multiplies block S's A fragments with column tiles 8 C .. 8 C + 7 and, one
// element per column-tile slot, produces row tile C of block S + 1.  What the producer costs per element (DIN = 1):
// fma, max and TWO conversions -- v_fma_mixlo/mixhi_f16 fold the row's power of two into the fp16 rounding
// (hi = fp16(h * s)) and the subtraction into the second one (lo = fp16(h * s - hi)); b1 / w1 are per-k records in
// LDS, read as 16-byte broadcast reads one or two slots ahead of their use.
//
// Layouts: W2 planes as rl8_mlp_pack_w2_f16(transposed = 0) writes them (16-byte unit ((hs*8 + ctl)*2 + p)*64 + l =
// plane p of B(col = 16 (8 (hs & 1) + ctl) + (l & 15), k = 32 (hs >> 1) + 8 (l >> 4) + e)); accumulators transposed
// (first MFMA operand = W2 fragment): a lane holds, for row (lane & 15) + 16 rt and column tile ct, the four
// consecutive columns 16 ct + 4 (lane >> 4) + r -- quads, a row being shared by the four lanes of equal lane & 15.
#include "split_tile.hip.h"

namespace rl8 {

constexpr int kRowsChunk = 16 * 1024;                  // one half-step of W2: [column tile 0..7][plane hi | lo] x 1 KiB
constexpr int kRowsHalfSteps = 16;
constexpr int kRowsPacked = kRowsHalfSteps * kRowsChunk;  // (= kF16PackedBytes; two floats behind: scale, 1 / scale)
constexpr int kRowsPitch = 128 + 16;                   // h2 transpose scratch (16 rows per wave): row pitch (conflict-free 16-byte accesses both ways)

// floats per k of the layer-1 record [b1 | w1[k][0..DIN-1] | pad]
// (width CLASSES since round 5: a kernel compiled for class DIN serves every run-time d_in <= DIN -- weights past d_in
// are zero, observations past d_in are not loaded.  Classes 1, 2, 3 form h1 with one fma per input on the vector ALU
// from these records; class 8 -- d_in = 4..8 -- and class 16 -- 9..16 -- form it on the MATRIX pipe, see "layer 1 as a matrix product" in the
// kernel: its "record" is 8 KiB of W1 fragments + 1 KiB of b1, nine floats per k.)
// (class 16 -- d_in = 9..16 -- is class 8 with a second pass of eight inputs: 16 KiB of fragments, two chained MFMAs)
constexpr int rows_record(int d_in) { return d_in == 1 ? 2 : d_in <= 3 ? 4 : d_in <= 8 ? 9 : 17; }
constexpr int rows_record_vecs(int d_in) { return d_in <= 3 ? 1 : d_in <= 8 ? 4 : 6; }  // 16-byte reads per record (classes 8, 16: per k BLOCK)
constexpr int rows_consts_bytes(int k_in, int k_out) { return (rows_record(k_in) + 1 + k_out) * kHidden * 4; }
constexpr int rows_lds_bytes(int ring, int k_in, int k_out, bool store) {
  return ring * kRowsChunk + rows_consts_bytes(k_in, k_out) + (store ? 4 * 16 * kRowsPitch : 0);
}

// (f16_pair_scaled: split_tile.hip.h)

// layer-1 record reads issued in slot s (0..7) of a half-step that produces (`produce`) / whose successor produces
// (`next_produces`): the record of element s + 1 (DIN = 1: of the element pair starting at s + 1, at odd s)
template <int DIN>
constexpr int rows_record_reads(int s, bool produce, bool next_produces) {
  const bool on = s == 7 ? next_produces : produce;
  if (DIN > 3) return s == 7 && on ? rows_record_vecs(DIN) : 0;  // (one request per k block: two W1 fragments per pass, eight b1)
  if (DIN == 1) return (s & 1) && on ? 1 : 0;
  return on ? rows_record_vecs(DIN) : 0;
}

// SAVE: 0 inference; 1 training with h2 and its gate bits (and, on request, h1) stored; 2 training with the gate bits
// alone (rank-one heads).  RING: chunks of W2 in LDS.
// DIAG (tuning builds only, -DRL8_ROWS_STAMP): 1 no production of the next fragments, 2 no epilogue arithmetic, 4 one
// product per column tile and row tile instead of three (no matrix work).
template <int DIN, int NOUT, int SAVE, int RING, int DIAG = 0>
__global__ __launch_bounds__(kBlock, 2) void mlp_rows_forward_kernel(
    const float *__restrict__ x, int64_t m, const float *__restrict__ w1, const float *__restrict__ b1,
    const void *__restrict__ w2s, const float *__restrict__ b2, const float *__restrict__ w3,
    const float *__restrict__ b3, float *__restrict__ out, float *__restrict__ save_h1, float *__restrict__ save_h2,
    uint32_t *__restrict__ save_gate2, int d_in, int n_out  // run-time widths: d_in <= DIN, n_out <= NOUT
#ifdef RL8_ROWS_STAMP  // tuning builds (tools/diag/rows_clock.py): shader-clock / real-time stamps around the kernel
    , unsigned long long *__restrict__ stamps
#endif
) {
#ifdef RL8_ROWS_STAMP
  const unsigned long long stamp_t0 = __builtin_amdgcn_s_memtime(), stamp_r0 = __builtin_amdgcn_s_memrealtime();
#endif
  constexpr int kIn = DIN;
  constexpr int kOut = NOUT;
  constexpr int kTile = 128;
  constexpr int kAhead = RING - 1;
  constexpr bool kStore = SAVE == 1;
  constexpr int kRec = rows_record(DIN);
  // Layer 1 as a matrix product (class 8).  z1 = W1 x is ONE 16x16x32 MFMA per sixteen hidden units and row tile: the
  // 32 k slots hold the FOUR plane products of up to eight inputs -- lanes 0..15 W1hi . xhi, 16..31 W1hi . xlo,
  // 32..47 W1lo . xhi, 48..63 W1lo . xlo -- so z1 carries all 22 + 22 operand bits (nothing dropped, unlike the
  // three-product layers) and costs the same whatever d_in is: 32 MFMAs per 32 rows (4 % of the layer-2 work) where the
  // vector ALU spent d_in fmas per element (d_in = 5: a quarter of the kernel's time, tools/diag/tower_width_sweep.py).
  // The product is transposed like layer 2's (first operand = the W1 fragment), and the fragment's sixteen M slots are
  // the hidden units 32 S + 8 (i >> 2) + 4 hf + (i & 3): lane (row l16, k block kq) then receives, from the two
  // fragments hf = 0, 1 of block S, exactly the eight k of its own layer-2 A fragment.  x is scaled per row, W1 per
  // tensor, by powers of two; bias, ReLU and the plane split of h1 stay on the vector ALU (four instructions per
  // element, what class 1 spends).
  constexpr bool kMma1 = DIN > 3;
  constexpr int kPasses = DIN > 8 ? 2 : 1;          // eight inputs per MFMA: class 16 chains two
  constexpr int kW1Planes = kPasses * 8 * 1024;     // [pass][S][hf][plane hi | lo][M slot] x 16 B (lanes kq and kq ^ 1 read the same plane)
  static_assert(DIN == 1 || DIN == 2 || DIN == 3 || DIN == 8 || DIN == 16, "width classes of layer 1");
  static_assert(NOUT == pad_out(NOUT), "output classes: 1, 2, 4, 8");
  static_assert(rows_lds_bytes(RING, kIn, kOut, kStore) <= 80 * 1024, "two workgroups per CU");
  static_assert(kAhead >= 2, "the mid-step barrier publishes a chunk requested at least a half-step earlier");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // [ring: RING x 16 KiB][layer-1 records [256][kRec] | b2 | w3 rows (zero rows up to kOut)][h2 transpose scratch]
  const unsigned lds0 = lds_offset(smem);
  constexpr int kConstOff = RING * kRowsChunk;
  constexpr int kB2Off = kConstOff + kRec * kHidden * 4;
  constexpr int kScratchOff = kConstOff + rows_consts_bytes(kIn, kOut);
  const int tid = threadIdx.x, lane = tid & 63, l16 = lane & 15, kq = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const __amdgpu_buffer_rsrc_t w2rsrc = buffer_rsrc(w2s, kRowsPacked);
  const float inv_w2_scale = reinterpret_cast<const float *>(static_cast<const unsigned char *>(w2s) + kRowsPacked)[1];

  const int64_t tiles = (m + kTile - 1) / kTile;
  const int64_t stride = gridDim.x;
  if ((int64_t)blockIdx.x >= tiles) return;

  // ---- constants into LDS; bounds of layer 1 for the row factors ---------------------------------
  float b1max = 0.0f, w1max[kIn];
  [[maybe_unused]] float inv_w1 = 1.0f;  // class 8: 1 / (W1's power of two)
  {
    float *consts = reinterpret_cast<float *>(smem + kConstOff);
    [[maybe_unused]] float wv[kIn];  // class 8: this thread's unit's weights, until the tensor's bound is known
    if constexpr (kMma1) {
      consts[kW1Planes / 4 + tid] = b1[tid];
#pragma unroll
      for (int i = 0; i < kIn; ++i) wv[i] = i < d_in ? w1[tid * d_in + i] : 0.0f;
    } else {
      consts[tid * kRec] = b1[tid];
#pragma unroll
      for (int i = 0; i < kRec - 1; ++i) consts[tid * kRec + 1 + i] = i < d_in ? w1[tid * d_in + i] : 0.0f;
    }
    consts[kRec * kHidden + tid] = b2[tid];
#pragma unroll
    for (int q = 0; q < kOut; ++q) consts[(kRec + 1 + q) * kHidden + tid] = q < n_out ? w3[q * kHidden + tid] : 0.0f;
    // max |b1|, max_k |w1[k][i]|: thread = k; a wave's 64 by lane exchange, the four waves' through LDS (the ring's
    // first bytes: nothing is requested into it before the prologue's barrier).  (Round 4: every thread used to walk
    // all 256 records in LDS, (1 + DIN) x 256 dependent reads -- 36 us of every launch at DIN = 1, 86 us at DIN = 5:
    // a fifth of a rollout timestep's launch of 2^20 rows, half of one of 2^18; tools/diag/forward_size_sweep.py.)
    float mine[1 + kIn];
    mine[0] = __builtin_fabsf(kMma1 ? consts[kW1Planes / 4 + tid] : consts[tid * kRec]);
#pragma unroll
    for (int i = 0; i < kIn; ++i) mine[1 + i] = __builtin_fabsf(kMma1 ? wv[i] : consts[tid * kRec + 1 + i]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
#pragma unroll
      for (int j = 0; j < 1 + kIn; ++j) mine[j] = __builtin_fmaxf(mine[j], __shfl_xor(mine[j], off, 64));
    float *red = reinterpret_cast<float *>(smem);  // [wave][1 + kIn]
    if (lane == 0) {
#pragma unroll
      for (int j = 0; j < 1 + kIn; ++j) red[wave * (1 + kIn) + j] = mine[j];
    }
    __syncthreads();
    auto of_all = [&](int j) {
      const float mx = __builtin_fmaxf(__builtin_fmaxf(red[j], red[(1 + kIn) + j]), __builtin_fmaxf(red[2 * (1 + kIn) + j], red[3 * (1 + kIn) + j]));
      return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(__builtin_fmaxf(mx, 0.0f))));  // (uniform: scalar registers)
    };
    b1max = of_all(0);
#pragma unroll
    for (int i = 0; i < kIn; ++i) w1max[i] = of_all(1 + i);
    if constexpr (kMma1) {
      // this unit's two planes into its M slot of fragment (S, hf): |w 2^(14 - e)| < 2^14 for the tensor's bound 2^e
      float all = 0.0f;
#pragma unroll
      for (int i = 0; i < kIn; ++i) all = __builtin_fmaxf(all, w1max[i]);
      const int ew = f16_bound_exponent(all);
      const float sw = __builtin_amdgcn_ldexpf(1.0f, kF16Top - ew);
      inv_w1 = __builtin_amdgcn_ldexpf(1.0f, ew - kF16Top);
      const int slot = 4 * ((tid >> 3) & 3) + (tid & 3), frag = 2 * (tid >> 5) + ((tid >> 2) & 1);
      u32x4 *planes = reinterpret_cast<u32x4 *>(smem + kConstOff);
#pragma unroll
      for (int pass = 0; pass < kPasses; ++pass) {
        u32x4 hi, lo;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          uint32_t h, l;
          f16_pair_scaled(wv[8 * pass + 2 * j], wv[8 * pass + 2 * j + 1], sw, h, l);
          hi[j] = h;
          lo[j] = l;
        }
        planes[pass * 512 + (frag * 2 + 0) * 16 + slot] = hi;
        planes[pass * 512 + (frag * 2 + 1) * 16 + slot] = lo;
      }
    }
  }

  // ---- per-lane state -------------------------------------------------------------------------------
  // rows of this lane: tile base + 32 wave + 16 rt + l16 (the four lanes of equal l16 hold the same rows)
  // (through a descriptor over the tile's rows: rows past the end read as zero, no per-lane 64-bit address or compare)
  auto load_x = [&](float (&dst)[2][kIn], int64_t tile) {
    const int64_t left = tile < tiles ? m - tile * kTile : 0;
    const int rows = left <= 0 ? 0 : left < kTile ? (int)left : kTile;
    const __amdgpu_buffer_rsrc_t xrsrc = buffer_rsrc(rows > 0 ? x + tile * kTile * d_in : x, rows * d_in * 4);
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int i = 0; i < kIn; ++i)  // (d_in is uniform: a scalar branch per i, nothing loaded past the row's end)
        dst[rt][i] = i < d_in ? buffer_load_f32(xrsrc, ((32 * wave + 16 * rt + l16) * d_in + i) * 4, 0) : 0.0f;
  };
  // Row factor: |h1[row][k]| <= max|b1| + sum_i |x_i| max_k |w1[k][i]| < 2^e  =>  planes of h1 * 2^(14 - e)
  // (exact); the accumulators are multiplied back by 2^(e - 14) / (W2's power of two).
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
  // Class 8: a lane loads only the input PAIR 2 kq, 2 kq + 1 of its rows (inputs past d_in: an offset past the
  // descriptor's end, which reads as zero); the four lanes of a row exchange bounds and fp16 words at the tile switch.
  auto row4 = [&](uint32_t v, auto combine) {  // over the four lanes of equal l16
    const auto s16 = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    const uint32_t u = combine(s16[0], s16[1]);
    const auto s32 = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return combine(s32[0], s32[1]);
  };
  auto load_pair = [&](float (&dst)[2][2 * kPasses], int64_t tile) {
    const int64_t left = tile < tiles ? m - tile * kTile : 0;
    const int rows = left <= 0 ? 0 : left < kTile ? (int)left : kTile;
    const __amdgpu_buffer_rsrc_t xrsrc = buffer_rsrc(rows > 0 ? x + tile * kTile * d_in : x, rows * d_in * 4);
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int j = 0; j < 2 * kPasses; ++j) {  // (j >> 1: the pass)
        const int f = 8 * (j >> 1) + 2 * kq + (j & 1);
        dst[rt][j] = buffer_load_f32(xrsrc, f < d_in ? ((32 * wave + 16 * rt + l16) * d_in + f) * 4 : 0x7ffffff0, 0);
      }
  };
  [[maybe_unused]] float wmx[2 * kPasses];  // max_k |w1[k][8 pass + 2 kq + j]|
  if constexpr (kMma1) {
#pragma unroll
    for (int j = 0; j < 2 * kPasses; ++j) {
      const int b = 8 * (j >> 1) + (j & 1);
      wmx[j] = kq == 0 ? w1max[b] : kq == 1 ? w1max[b + 2] : kq == 2 ? w1max[b + 4] : w1max[b + 6];
    }
  }
  [[maybe_unused]] u32x4 xf[2][kPasses];  // classes 8, 16: the rows' layer-1 fragments (plane hi in even k blocks, lo in odd ones)
  [[maybe_unused]] float inv_x[2];  // ... and 1 / (the row's power of two x W1's)
  // from the pair: the factor of h1 (as row_scales), the factor of x (|x_i| < 2^e over the row) and the fragment
  auto row_fragments = [&](const float (&xs)[2][2 * kPasses], float (&scale)[2], float (&inv)[2]) {
    const auto fmax_u = [](uint32_t a, uint32_t b) { return __float_as_uint(__builtin_fmaxf(__uint_as_float(a), __uint_as_float(b))); };
    const auto fadd_u = [](uint32_t a, uint32_t b) { return __float_as_uint(__uint_as_float(a) + __uint_as_float(b)); };
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      float big = 0.0f, weighed = 0.0f;
#pragma unroll
      for (int j = 0; j < 2 * kPasses; ++j) {
        big = __builtin_fmaxf(big, __builtin_fabsf(xs[rt][j]));
        weighed = __builtin_fmaf(__builtin_fabsf(xs[rt][j]), wmx[j], weighed);
      }
      const float most = __uint_as_float(row4(__float_as_uint(big), fmax_u));
      const float part = __uint_as_float(row4(__float_as_uint(weighed), fadd_u));
      const int e = f16_bound_exponent(b1max + part);
      scale[rt] = __builtin_amdgcn_ldexpf(1.0f, kF16Top - e);
      inv[rt] = __builtin_amdgcn_ldexpf(inv_w2_scale, e - kF16Top);
      const int ex = f16_bound_exponent(most);
      inv_x[rt] = __builtin_amdgcn_ldexpf(inv_w1, ex - kF16Top);
#pragma unroll
      for (int pass = 0; pass < kPasses; ++pass) {
        uint32_t hi, lo;
        f16_pair_scaled(xs[rt][2 * pass], xs[rt][2 * pass + 1], __builtin_amdgcn_ldexpf(1.0f, kF16Top - ex), hi, lo);
#pragma unroll
        for (int j = 0; j < 4; ++j) {  // inputs 8 pass + 2 j, + 1 live in lane l16 + 16 j
          const uint32_t h = __shfl(hi, l16 + 16 * j, kWave), l = __shfl(lo, l16 + 16 * j, kWave);
          xf[rt][pass][j] = (kq & 1) ? l : h;
        }
      }
    }
  };
  constexpr int kXRegs = kMma1 ? 2 * kPasses : kIn;
  [[maybe_unused]] float xc[2][kXRegs];
  float xn[2][kXRegs];
  float sc[2], inv_c[2];
  if constexpr (kMma1) {
    load_pair(xn, blockIdx.x);
    row_fragments(xn, sc, inv_c);
    load_pair(xn, blockIdx.x + stride);
  } else {
    load_x(xc, blockIdx.x);
    load_x(xn, blockIdx.x + stride);
    row_scales(xc, sc, inv_c);
  }

  // ---- W2 ring ------------------------------------------------------------------------------------------
  auto request_chunk = [&](int hs, int stage) {  // sixteen one-KiB pieces, four per wave
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int piece = wave * 4 + u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(w2rsrc, smem + stage * kRowsChunk + piece * 1024, 16, lane * 16,
                                               hs * kRowsChunk + piece * 1024, 0, 0);
    }
  };
  const unsigned b_lane = lds0 + lane * 16;
  // this lane's eight k of a block: + 32 kRec 4 S  (class 8: its M slot of the plane its k block multiplies: + 1024 S
  // + 512 hf; and the eight b1 of its k: + 128 S)
  const unsigned c_lane = kMma1 ? lds0 + kConstOff + (kq >> 1) * 256 + l16 * 16 : lds0 + kConstOff + kq * (8 * kRec * 4);
  [[maybe_unused]] const unsigned b1_lane = lds0 + kConstOff + kW1Planes + kq * 32;

  // ---- layer-1 records: requested one slot ahead of the element they serve ------------------------------------------
  //   DIN = 1: one 16-byte read holds the records of an element PAIR (requested in the odd slot in front of it);
  //   DIN = 2, 3: one read per element;  DIN = 5: two reads per element.  Two register sets each.
  //   class 8: four reads per k BLOCK (two W1 fragments, eight b1), requested in the last slot of the half-step before.
  constexpr int kCSets = kMma1 ? 1 : 2;
  constexpr int kCReads = rows_record_vecs(DIN);
  u32x4 cq[kCSets][kCReads];
  auto request_record = [&](int S, int e) {  // the reads for element e (0..7) of block S
    if constexpr (kMma1) {  // (the whole block's: fragments hf = 0, 1 and b1 of the lane's eight k)
      const unsigned a = c_lane + S * 1024, b = b1_lane + S * 128;
      cq[0][0] = lds_read_b128<0>(a);
      cq[0][1] = lds_read_b128<512>(a);
      cq[0][2] = lds_read_b128<0>(b);
      cq[0][3] = lds_read_b128<16>(b);
      if constexpr (kPasses == 2) {
        cq[0][4] = lds_read_b128<8192>(a);
        cq[0][5] = lds_read_b128<8192 + 512>(a);
      }
      return;
    }
    const unsigned a = c_lane + S * (32 * kRec * 4);
    if constexpr (DIN == 1) {  // e even: the pair (e, e + 1)
      const int set = (e >> 1) & 1;
      cq[set][0] = e == 0 ? lds_read_b128<0>(a) : e == 2 ? lds_read_b128<16>(a) : e == 4 ? lds_read_b128<32>(a) : lds_read_b128<48>(a);
    } else if constexpr (DIN <= 3) {
      const int set = e & 1;
      cq[set][0] = e == 0   ? lds_read_b128<0>(a)
                   : e == 1 ? lds_read_b128<16>(a)
                   : e == 2 ? lds_read_b128<32>(a)
                   : e == 3 ? lds_read_b128<48>(a)
                   : e == 4 ? lds_read_b128<64>(a)
                   : e == 5 ? lds_read_b128<80>(a)
                   : e == 6 ? lds_read_b128<96>(a)
                            : lds_read_b128<112>(a);
    }
  };
  // element e of row tile rt from its record (behind a counted wait that covers it)
  auto h1_element = [&](const float (&xs)[2][kXRegs], int rt, int e) {
    if constexpr (kMma1) {
      return 0.0f;
    } else {
      const int set = DIN == 1 ? (e >> 1) & 1 : e & 1;
#pragma unroll
      for (int r = 0; r < kCReads; ++r) {
        u32x4 &q = cq[set][r];  // (named outside the asm: operands alone do not capture in a generic lambda)
        asm volatile("" : "+v"(q));
      }
      const int at = DIN == 1 ? 2 * (e & 1) : 0;
      float v = __uint_as_float(cq[set][0][at]);
#pragma unroll
      for (int i = 0; i < kIn; ++i) {
        const int j = at + 1 + i;
        v = __builtin_fmaf(xs[rt][i], __uint_as_float(cq[set][j >> 2][j & 3]), v);
      }
      return relu1(v);
    }
  };
  // class 8: z1 of the eight k of block S (its four reads have landed) for row tile rt; then + b1, before the ReLU
  auto l1_products = [&](int rt, f32x4 (&z)[2]) {
    if constexpr (kMma1) {
      asm volatile("" : "+v"(cq[0][0]), "+v"(cq[0][1]), "+v"(cq[0][2]), "+v"(cq[0][3]));
      const f32x4 zero = {0, 0, 0, 0};
      z[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, cq[0][0]), __builtin_bit_cast(half8, xf[rt][0]), zero, 0, 0, 0);
      z[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, cq[0][1]), __builtin_bit_cast(half8, xf[rt][0]), zero, 0, 0, 0);
      if constexpr (kPasses == 2) {  // inputs 8..15 onto the same accumulators
        asm volatile("" : "+v"(cq[0][kCReads - 2]), "+v"(cq[0][kCReads - 1]));
        z[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, cq[0][kCReads - 2]), __builtin_bit_cast(half8, xf[rt][kPasses - 1]), z[0], 0, 0, 0);
        z[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, cq[0][kCReads - 1]), __builtin_bit_cast(half8, xf[rt][kPasses - 1]), z[1], 0, 0, 0);
      }
    }
  };
  auto l1_bias = [&](int rt, const f32x4 (&z)[2], float (&h)[8]) {
    if constexpr (kMma1) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        h[r] = __builtin_fmaf(z[0][r], inv_x[rt], __uint_as_float(cq[0][2][r]));
        h[4 + r] = __builtin_fmaf(z[1][r], inv_x[rt], __uint_as_float(cq[0][3][r]));
      }
    }
  };

  f32x4 acc[2][16];
  u32x4 a_hi[2][2], a_lo[2][2];    // A fragments: [set = k block parity][row tile]
  u32x4 bh[2], bl[2];              // W2 fragments of local column tile ctl live in set ctl % 2, requested one slot ahead
  auto request_block = [&](unsigned br, int ctl, int set) {
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
  // h1 of eight consecutive k of one row (optional output of SAVE = 1): two 16-byte stores through the wave's descriptor
  [[maybe_unused]] __amdgpu_buffer_rsrc_t h1rsrc = buffer_rsrc(nullptr, 0);
  auto store_h1 = [&](const float (&h)[8], int rt, int S) {
    if constexpr (kStore) {
      if (save_h1 != nullptr) {
        const int voff = ((16 * rt + l16) * kHidden + 32 * S + 8 * kq) * 4;
        __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(h[0]), __float_as_uint(h[1]), __float_as_uint(h[2]), __float_as_uint(h[3])},
                                               h1rsrc, voff, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(h[4]), __float_as_uint(h[5]), __float_as_uint(h[6]), __float_as_uint(h[7])},
                                               h1rsrc, voff + 16, 0, 0);
      }
    }
  };

  // One half-step: column tiles 8 C .. 8 C + 7 of k block S (chunk hs = 2 S + C) in eight slots: the fragments of the
  // tile one slot on are requested, the slot's own waited for by COUNT (LDS returns in order; no scalar load is in
  // flight in this loop), 2 x 3 products issued, and one element of row tile C of block S + 1 formed beside them.
  // Behind slot 3 the barrier that publishes the next chunk (its loads were issued two half-steps ago: vector memory
  // returns in order, so "all but the pieces of the kAhead - 2 chunks requested since" covers it; anything else in
  // flight only makes this wait longer) and frees the stage of the chunk before this one, which slots 4..7 refill,
  // one direct-to-LDS piece each.  CUR: the A set of block S.
  // KIND 0: any half-step up to (6, 0); 1: (6, 1) -- its successor produces nothing, so no records are requested for
  // it; 2: (7, 0), nothing produced; 3: (7, 1), the tile's last: nothing requested for a successor either (the
  // epilogue lies in between and needs the registers; open_tile() restarts the stream).
  auto do_half = [&](auto first_tag, auto cur_tag, auto c_tag, auto kind_tag, int hs, int stage) {
    constexpr bool FIRST = decltype(first_tag)::value;
    constexpr int CUR = decltype(cur_tag)::value;
    constexpr int C = decltype(c_tag)::value;
    constexpr int KIND = decltype(kind_tag)::value;
    constexpr bool kProduce = KIND <= 1 && !(DIAG & 1), kNextProduces = KIND == 0 && !(DIAG & 1), kEnd = KIND == 3;
    const int stage_next = stage + 1 == RING ? 0 : stage + 1, stage_free = stage == 0 ? RING - 1 : stage - 1;
    const unsigned br = b_lane + stage * kRowsChunk, br_next = b_lane + stage_next * kRowsChunk;
    const int S = hs >> 1;
    [[maybe_unused]] float h[8];
    [[maybe_unused]] f32x4 z1v[2];
#pragma unroll
    for (int sl = 0; sl < 8; ++sl) {
      const int set = sl & 1, ahead = set ^ 1, ct = 8 * C + sl;
      // the record of the element one slot on first (older than the fragments requested with it, hence covered by
      // the next slot's wait for those) ...
      {
        const bool on = sl == 7 ? kNextProduces : kProduce;
        const int S1 = sl == 7 ? S + 1 + C : S + 1, e1 = (sl + 1) & 7;
        if (on && (kMma1 ? sl == 7 : (DIN != 1 || (sl & 1)))) request_record(S1, e1);
      }
      // ... then the fragments of the column tile one slot on (two waves share a SIMD's matrix pipe: a slot's six
      // products take ~190 cycles of wall time, more than an LDS round trip)
      if (sl < 7) request_block(br, sl + 1, ahead);
      else if (!kEnd) request_block(br_next, 0, ahead);
      // in-order LDS returns: everything but what this slot has just requested has landed
      {
        const int allowed = rows_record_reads<DIN>(sl, kProduce, kNextProduces) + ((kEnd && sl == 7) ? 0 : 2);
        allowed == 0   ? wait_lds<0>(bh[set], bl[set])
        : allowed == 2 ? wait_lds<2>(bh[set], bl[set])
        : allowed == 3 ? wait_lds<3>(bh[set], bl[set])
        : allowed == 4 ? wait_lds<4>(bh[set], bl[set])
        : allowed == 5 ? wait_lds<5>(bh[set], bl[set])
        : allowed == 6 ? wait_lds<6>(bh[set], bl[set])
        : allowed == 8 ? wait_lds<8>(bh[set], bl[set])
                       : wait_lds<0>(bh[set], bl[set]);
      }
      if constexpr (kProduce && kMma1) {  // the two small products of this half-step's k block, ahead of the slot's six
        if (sl == 0) l1_products(C, z1v);
      }
      const f32x4 zero = {0, 0, 0, 0};
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, bh[set]),
                                                             __builtin_bit_cast(half8, a_lo[CUR][rt]),
                                                             FIRST ? zero : acc[rt][ct], 0, 0, 0);
        if constexpr ((DIAG & 4) != 0) {  // (the other operands stay live)
          asm volatile("" ::"v"(bl[set]), "v"(a_hi[CUR][rt]));
          continue;
        }
        acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, bl[set]),
                                                             __builtin_bit_cast(half8, a_hi[CUR][rt]), acc[rt][ct], 0, 0, 0);
        acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, bh[set]),
                                                             __builtin_bit_cast(half8, a_hi[CUR][rt]), acc[rt][ct], 0, 0, 0);
      }
      if constexpr (kProduce) {  // element sl of row tile C of block S + 1
        if constexpr (kMma1) {
          if (sl == 0) l1_bias(C, z1v, h);  // (behind this slot's six products: the two small ones finished under them)
          h[sl] = relu1(h[sl]);
        } else {
          h[sl] = h1_element(xc, C, sl);
        }
        if (sl & 1) {
          uint32_t hi, lo;
          f16_pair_scaled(h[sl - 1], h[sl], sc[C], hi, lo);
          a_hi[CUR ^ 1][C][sl >> 1] = hi;
          a_lo[CUR ^ 1][C][sl >> 1] = lo;
        }
        if (sl == 7) store_h1(h, C, S + 1);
      } else if constexpr (KIND <= 1) {  // (tuning builds without production: the next block re-uses these fragments)
        if (sl == 7) {
          a_hi[CUR ^ 1][C] = a_hi[CUR][C];
          a_lo[CUR ^ 1][C] = a_lo[CUR][C];
        }
      }
      if (sl == 3) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(4 * (kAhead - 2)) : "memory");
      if (sl >= 4) {
        const int piece = wave * 4 + (sl - 4);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(w2rsrc, smem + stage_free * kRowsChunk + piece * 1024, 16, lane * 16,
                                                 ((hs + kAhead) & (kRowsHalfSteps - 1)) * kRowsChunk + piece * 1024, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // Start of a tile's stream: both row tiles' fragments of block 0 into set 0, then what half-step (0, 0) expects to
  // be on its way: [record 0 of block 1][column tile 0] (chunk 0 sits in `stage`, published by the barrier of the
  // half-step before, or the prologue's).  (DIN = 1: the pair (0, 1) -- requested by slot 7's rule, i.e. here too.)
  auto open_tile = [&](int stage) {
    {
      float h[2][8];
      if constexpr (kMma1) {
        request_record(0, 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
          f32x4 z[2];
          l1_products(rt, z);
          l1_bias(rt, z, h[rt]);
#pragma unroll
          for (int e = 0; e < 8; ++e) h[rt][e] = relu1(h[rt][e]);
        }
      } else {
        constexpr int kBatch = DIN == 1 ? 4 : 2;  // elements whose records fit the two register sets at once
#pragma unroll
        for (int e0 = 0; e0 < 8; e0 += kBatch) {
#pragma unroll
          for (int e = e0; e < e0 + kBatch; ++e)
            if (DIN != 1 || (e & 1) == 0) request_record(0, e);
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
          for (int e = e0; e < e0 + kBatch; ++e)
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) h[rt][e] = h1_element(xc, rt, e);
        }
      }
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
          uint32_t hi, lo;
          f16_pair_scaled(h[rt][e], h[rt][e + 1], sc[rt], hi, lo);
          a_hi[0][rt][e >> 1] = hi;
          a_lo[0][rt][e >> 1] = lo;
        }
        store_h1(h[rt], rt, 0);
      }
    }
    const unsigned br = b_lane + stage * kRowsChunk;
    if (!(DIAG & 1)) request_record(1, 0);
    request_block(br, 0, 0);
  };

  using T = std::true_type;
  using F = std::false_type;
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using K0 = I0;
  using K1 = I1;
  using K2 = std::integral_constant<int, 2>;
  using K3 = std::integral_constant<int, 3>;

  // ---- prologue: the first kAhead chunks on their way; chunk 0 readable by every wave ------------------------------
  __syncthreads();  // (the constants are in LDS for every wave; also orders the reads of the bound loops above)
#pragma unroll
  for (int d = 0; d < kAhead; ++d) request_chunk(d, d);
  asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(4 * (kAhead - 1)) : "memory");

  int stage = 0;
  auto next_stage = [&]() { stage = stage + 1 == RING ? 0 : stage + 1; };
  for (int64_t tile = blockIdx.x; tile < tiles; tile += stride) {
    const int64_t r0 = tile * kTile;
    const int wrow0 = 32 * wave;  // first row of this wave in the tile
    const int64_t rows_left = m - r0 - wrow0;
    const int wrows = rows_left <= 0 ? 0 : rows_left < 32 ? (int)rows_left : 32;  // rows of this wave that exist
    if constexpr (kStore) h1rsrc = buffer_rsrc(save_h1 ? save_h1 + (r0 + wrow0) * kHidden : nullptr, save_h1 ? wrows * kHidden * 4 : 0);
    open_tile(stage);
    do_half(T{}, I0{}, I0{}, K0{}, 0, stage);  next_stage();
    do_half(T{}, I0{}, I1{}, K0{}, 1, stage);  next_stage();
    do_half(F{}, I1{}, I0{}, K0{}, 2, stage);  next_stage();
    do_half(F{}, I1{}, I1{}, K0{}, 3, stage);  next_stage();
#pragma unroll 1
    for (int hs = 4; hs < kRowsHalfSteps - 4; hs += 4) {
      do_half(F{}, I0{}, I0{}, K0{}, hs, stage);      next_stage();
      do_half(F{}, I0{}, I1{}, K0{}, hs + 1, stage);  next_stage();
      do_half(F{}, I1{}, I0{}, K0{}, hs + 2, stage);  next_stage();
      do_half(F{}, I1{}, I1{}, K0{}, hs + 3, stage);  next_stage();
    }
    do_half(F{}, I0{}, I0{}, K0{}, kRowsHalfSteps - 4, stage);  next_stage();
    do_half(F{}, I0{}, I1{}, K1{}, kRowsHalfSteps - 3, stage);  next_stage();
    do_half(F{}, I1{}, I0{}, K2{}, kRowsHalfSteps - 2, stage);  next_stage();
    do_half(F{}, I1{}, I1{}, K3{}, kRowsHalfSteps - 1, stage);  next_stage();

    // ---- epilogue, per wave: this lane holds rows 16 rt + l16 and, of column tile ct, columns 16 ct + 4 kq + r ---------
    // (lane coordinates re-derived behind an opaque copy: everything computed from them below is then re-made per
    // tile -- a handful of instructions -- instead of being hoisted out of the tile loop and carried, or spilled,
    // across the matrix loop)
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
    const int l16e = lane_e & 15, kqe = lane_e >> 4;
    const unsigned constp = lds0 + kB2Off + (4 * kqe) * 4;
    constexpr int kChains = kOut <= 2 ? 4 : 2;
    float part[2][kOut][kChains];
    [[maybe_unused]] uint32_t gate_words[2][8];
    [[maybe_unused]] const unsigned t_base = lds0 + kScratchOff + wave * (16 * kRowsPitch);
    [[maybe_unused]] const unsigned t_write = t_base + l16e * kRowsPitch + 16 * kqe;
    [[maybe_unused]] const unsigned t_read = t_base + (lane_e >> 3) * kRowsPitch + (lane_e & 7) * 16;
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t h2rsrc =
        buffer_rsrc(kStore ? save_h2 + (r0 + wrow0) * kHidden : nullptr, wrows * kHidden * 4);
    [[maybe_unused]] const int h2_voff = ((lane_e >> 3) * kHidden + 4 * (lane_e & 7)) * 4;
    // The column tiles in groups of kGroup (one quad each): b2 and the kOut rows of W3 for the group from LDS, then
    // bias + ReLU, the h2 quads into the transpose scratch, gate nibbles and head products; software-pipelined (the
    // next group's b2 quads are requested as soon as bias + ReLU has consumed this group's, its W3 quads as soon as
    // the head products have; the transposed block is waited for by count behind the gate and head arithmetic).
    constexpr int kGroup = kOut >= 4 ? 1 : 2;
    constexpr int kStages = (DIAG & 2) ? 0 : 16 / kGroup;
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
                      : q == 3 ? lds_read_b128<4 * kHidden * 4>(a)
                      : q == 4 ? lds_read_b128<5 * kHidden * 4>(a)
                      : q == 5 ? lds_read_b128<6 * kHidden * 4>(a)
                      : q == 6 ? lds_read_b128<7 * kHidden * 4>(a)
                               : lds_read_b128<8 * kHidden * 4>(a);
      }
    };
    static_assert(kOut <= 8, "request_w3 is written out for eight outputs");
    if constexpr ((DIAG & 2) != 0) {  // tuning builds: the accumulators are consumed, nothing else
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
#pragma unroll
        for (int q = 0; q < kOut; ++q)
#pragma unroll
          for (int j = 0; j < kChains; ++j) part[rt][q][j] = 0.0f;
#pragma unroll
        for (int ct = 0; ct < 16; ++ct) {
          part[rt][0][0] += acc[rt][ct][0] + acc[rt][ct][3];
          if constexpr (SAVE != 0) gate_words[rt][ct >> 1] = 0;
        }
      }
    } else {
      request_b2(0);
      request_w3(0);
    }
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
          for (int e = 0; e < 4; ++e)  // (not __builtin_bit_cast on a vector-element lvalue: it reads element 0)
            acc[rt][ct][e] = relu1(__builtin_fmaf(acc[rt][ct][e], inv_c[rt], __uint_as_float(bq[gi][e])));
        }
      if (st + 1 < kStages) request_b2(st + 1);
      if constexpr (kStore) {
        // block [32 rows][32 columns], one row tile (16 rows) at a time -> the wave's transpose scratch (pitch 144 B),
        // back as eight lanes per row: a store instruction is then eight full 128-byte lines.  Both column tiles of the
        // block are final here (bias + ReLU of the even one ran a stage ago when kGroup = 1); the second row tile's
        // writes follow the first one's reads into the same 16 rows -- a wave's LDS operations execute in order.
        if (last_of_block) {
          const int c0 = (st * kGroup + kGroup - 1) & ~1;
#pragma unroll
          for (int rt = 0; rt < 2; ++rt) {
            lds_write_b128<0>(t_write, __builtin_bit_cast(u32x4, acc[rt][c0]));
            lds_write_b128<64>(t_write, __builtin_bit_cast(u32x4, acc[rt][c0 + 1]));
            t_rows[2 * rt] = lds_read_b128<0>(t_read);
            t_rows[2 * rt + 1] = lds_read_b128<8 * kRowsPitch>(t_read);
          }
        }
      }
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int gi = 0; gi < kGroup; ++gi) {
          const int ct = st * kGroup + gi;
          if constexpr (SAVE != 0) {
            // gate of h2: bit b of word w of a row <=> column 32 w + b > 0; h2 >= +0, so "h2 > 0" is bit 31 of
            // (bits + 0x7fffffff); this lane's nibble of column tile ct sits in word ct / 2 at bit 16 (ct & 1) + 4 kq
            uint32_t nib = 0;
#pragma unroll
            for (int e = 3; e >= 0; --e) nib = __builtin_amdgcn_alignbit(nib, __float_as_uint(acc[rt][ct][e]) + 0x7fffffffu, 31);
            gate_words[rt][ct >> 1] = ((ct & 1) == 0 ? 0u : gate_words[rt][ct >> 1]) | (nib << (16 * (ct & 1) + 4 * kqe));
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
          // the block's four reads are older than the next stage's W3 quads just requested (its b2 quads went out
          // ahead of the block): wait for "all but those"
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
    // a row's columns are spread over the four lanes of equal l16: two swaps complete its sums / words
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
      const int row = 16 * rt + l16e;  // within the wave
      if constexpr (SAVE != 0) {
        if (save_gate2 != nullptr) {
          // full words = the four lanes' nibbles; lane kq stores words 2 kq, 2 kq + 1 of its row
          uint32_t full[8];
#pragma unroll
          for (int w = 0; w < 8; ++w) full[w] = across_row(gate_words[rt][w], [](uint32_t a, uint32_t b) { return a | b; });
          const uint32_t w0 = kqe == 0 ? full[0] : kqe == 1 ? full[2] : kqe == 2 ? full[4] : full[6];
          const uint32_t w1v = kqe == 0 ? full[1] : kqe == 1 ? full[3] : kqe == 2 ? full[5] : full[7];
          if (row < wrows) {
            uint32_t *dst = save_gate2 + (r0 + wrow0) * 8;  // uniform base, 32-bit lane offset
            *reinterpret_cast<u32x2 *>(dst + (unsigned)(row * 8 + 2 * kqe)) = u32x2{w0, w1v};
          }
        }
      }
      if (kqe == 0 && row < wrows) {
        float *dst = out + (r0 + wrow0) * n_out;
#pragma unroll
        for (int q = 0; q < kOut; ++q)
          if (q < n_out) dst[(unsigned)(row * n_out + q)] = total[rt][q] + b3[q];
      }
    }
    // the workgroup's next tile
    if constexpr (kMma1) {
      row_fragments(xn, sc, inv_c);
      load_pair(xn, tile + 2 * stride);
    } else {
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int i = 0; i < kIn; ++i) xc[rt][i] = xn[rt][i];
      load_x(xn, tile + 2 * stride);
      row_scales(xc, sc, inv_c);
    }
  }
  // (the ring's last requests ran past the last tile: nothing may land in LDS after the workgroup is gone)
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#ifdef RL8_ROWS_STAMP
  if (stamps != nullptr && lane == 0) {  // (a buffer of their own: no output depends on the stamps)
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    unsigned long long *dst = stamps + ((size_t)blockIdx.x * 4 + wave) * 4;
    dst[0] = t1 - stamp_t0;
    dst[1] = r1 - stamp_r0;
    dst[2] = (unsigned long long)((tiles - blockIdx.x + stride - 1) / stride);  // tiles this workgroup ran
    dst[3] = 0;
  }
#endif
}

// ---- data gradient of rank-one heads (gate mode) in the same structure ------------------------------------------
// dZ2[s][k] = G[s][k] d[s] w3e[k] (one output, or two outputs with exactly opposite gradients), so
//   dH1[s][i] = d[s] sum_k G[s][k] (w3e[k] W2[k][i]):
// the A operand is the ReLU gate of h2 itself -- ONE fp16 plane of zeros and ones made from the forward's gate BITS
// without any arithmetic -- and B the two planes of w3e[k] W2[k][i] (rl8_mlp_pack_w2_f16_gate): two products per
// fragment pair.  dZ1 = dH1 * (h1 > 0) is folded into db1 / dW1 in the epilogue, the gate of h1 recomputed from the
// observations (the forward's own fma chain).
//
// What differs from the forward kernel above:
//   * the product is NOT transposed (first MFMA operand = the gate fragment): a lane holds, for column
//     16 ct + (lane & 15) and row tile rt, the four rows 16 rt + 4 (lane >> 4) + r -- the column sums over rows that
//     db1 / dW1 are start as in-lane sums over registers;
//   * B arrives in rl8_mlp_pack_w2_f16's 32x32x16 unit order (the pack is shared with the previous kernel): a
//     16x16x32 fragment's units exist there one for one, at other addresses -- a chunk is the two 8-KiB runs of k
//     steps 2 S and 2 S + 1 over column tiles 4 C .. 4 C + 3, read with per-lane bases;
//   * the wave's gate bits (32 rows x 32 B = 1 KiB) come by ONE direct-to-LDS load per tile into a per-wave block,
//     requested behind the last fragment production of the tile before (counted in the barriers' vmcnt);
//   * per-row factors (d[s] / W2's power of two) and observations go through a per-wave LDS exchange: the lane that
//     loads a row is not the lane that holds its accumulators;
//   * running column sums [256][db1 | dW1 row | pad] per WAVE in LDS, updated per tile by the wave alone (no workgroup
//     barrier), added over the four waves in order at the end.
constexpr int rows_dgrad_lds_bytes(int ring, int k_in) {
  return ring * kRowsChunk + rows_record(k_in) * kHidden * 4 + 4 * (1024 + 32 * (1 + k_in) * 4 + rows_record(k_in) * kHidden * 4);
}

template <int OFF>
__device__ __forceinline__ u32x2 lds_read_b64(unsigned addr) {
  u32x2 v;
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
template <int OFF>
__device__ __forceinline__ uint32_t lds_read_u8(unsigned addr) {
  uint32_t v;
  asm volatile("ds_read_u8 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}

// ---- the data gradients' class 8 (d_in = 4..7): layer 1 on the matrix pipe here too ---------------------------------------
// Per element of dZ1 the vector ALU spent 2 d_in + 3 instructions -- d_in fmas for the gate of h1, d_in for dW1 -- and the
// wave's running sums [256][1 + d_in] took (1 + d_in) KiB of LDS, four times per workgroup: d_in = 3 was the last width
// that fit.  Class 8 moves both onto MFMAs:
//   * the gate of h1 is the FORWARD's z1 again -- the same fragments (x scaled per row, W1 per tensor, four plane
//     products in the 32 k slots), one product per column tile and row tile, then fma(z, 1 / scale, b1) > 0: the very
//     value the forward rounded, so the two kernels never disagree on a gate;
//   * dW1 / db1 of the wave's 32 rows: K = rows.  dZ1's eight values per lane ARE a B fragment of that product (column
//     = lane & 15, k slot (kq, j) = row 16 (j >> 2) + 4 kq + (j & 3)) once scaled by a power of two per WAVE and split in
//     two planes; the A fragment holds x~ = x * 2^s (per wave again: rows are summed inside the instruction) with the hi
//     planes of inputs 0..6 and a column of ones in M slots 0..7 and the lo planes in slots 8..15, so
//     mfma(A, Dhi) + mfma(A, Dlo) delivers all four plane products, db1 in slot 7; the lanes of slots 8..15 hand their
//     part to those of slots 0..7 (one swap);
//   * the running sums are ONE array [256][8] per workgroup, updated block by block in wave order: wave w adds its
//     part of column tile ct once wave w - 1 has published "ct done" (a counter in LDS, polled; the sums' read rides
//     on the poll: LDS executes a wave's operations in order, and both are repeated if the counter was not there yet).
//     The order is fixed, so the sums repeat bit for bit; no wave waits for a later one, so the chain cannot lock; the
//     sixteen barriers of the next tile's matrix loop separate one tile's updates from the next one's.
constexpr int kRows8Planes = 8 * 1024;                     // W1 planes, [column tile][hi | lo][unit] x 16 B
constexpr int kRows8Sums = kHidden * 8 * 4;                // [unit][dW1 row 0..6 | db1]
constexpr int kRows8FacOff = 1024, kRows8InvOff = kRows8FacOff + 128, kRows8XtOff = kRows8InvOff + 128;
constexpr int kRows8Wave = kRows8XtOff + 1024;            // gate block | row factors | rows' 1 / x scale | x~ [slot][row] fp16
constexpr int rows8_dgrad_lds_bytes(int ring, int k_out) {
  return ring * kRowsChunk + kRows8Planes + 1024 + kRows8Sums + 64 + 4 * kRows8Wave + k_out * kHidden * 4;
}
__device__ __forceinline__ void lds_write_b16(unsigned addr, uint32_t v) {
  asm volatile("ds_write_b16 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
__device__ __forceinline__ void lds_write_b16_hi(unsigned addr, uint32_t v) {
  asm volatile("ds_write_b16_d16_hi %0, %1" ::"v"(addr), "v"(v) : "memory");
}
template <int OFF>
__device__ __forceinline__ uint32_t lds_read_b32_nowait(unsigned addr) {
  uint32_t v;
  asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
__device__ __forceinline__ float rows8_row4(float v, bool sum) {  // over the four lanes of equal lane & 15
  const auto s16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  const float a = __uint_as_float(s16[0]), b = __uint_as_float(s16[1]);
  const float u = sum ? a + b : __builtin_fmaxf(a, b);
  const auto s32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(u), __float_as_uint(u), false, false);
  const float c = __uint_as_float(s32[0]), d = __uint_as_float(s32[1]);
  return sum ? c + d : __builtin_fmaxf(c, d);
}

// Prologue: W1's planes (the forward's power of two: the tensor's bound) and b1 into LDS, sums zeroed, the chain's
// counters set (word 0: "wave -1", always done).  `red`: sixteen bytes of LDS nobody else uses before the next barrier.
// Returns 1 / (W1's power of two).  Contains a workgroup barrier.
__device__ __forceinline__ float rows8_constants(unsigned char *base, float *red, const float *__restrict__ w1,
                                                 const float *__restrict__ b1, int d_in, int tid) {
  float wv[8], mx = 0.0f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    wv[i] = i < d_in ? w1[tid * d_in + i] : 0.0f;
    mx = __builtin_fmaxf(mx, __builtin_fabsf(wv[i]));
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mx = __builtin_fmaxf(mx, __shfl_xor(mx, off, 64));
  if ((tid & 63) == 0) red[tid >> 6] = mx;
  reinterpret_cast<float *>(base + kRows8Planes)[tid] = b1[tid];
  float *sums = reinterpret_cast<float *>(base + kRows8Planes + 1024);
#pragma unroll
  for (int i = 0; i < 8; ++i) sums[i * kBlock + tid] = 0.0f;
  if (tid < 8) reinterpret_cast<int *>(base + kRows8Planes + 1024 + kRows8Sums)[tid] = tid == 0 ? 0x7fffffff : 0;
  __syncthreads();
  const float all = __builtin_fmaxf(__builtin_fmaxf(red[0], red[1]), __builtin_fmaxf(red[2], red[3]));
  const int ew = f16_bound_exponent(__uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(all))));
  const float sw = __builtin_amdgcn_ldexpf(1.0f, kF16Top - ew);
  u32x4 hi, lo;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    uint32_t h, l;
    f16_pair_scaled(wv[2 * j], wv[2 * j + 1], sw, h, l);
    hi[j] = h;
    lo[j] = l;
  }
  u32x4 *planes = reinterpret_cast<u32x4 *>(base);
  planes[((tid >> 4) * 2 + 0) * 16 + (tid & 15)] = hi;
  planes[((tid >> 4) * 2 + 1) * 16 + (tid & 15)] = lo;
  return __builtin_amdgcn_ldexpf(1.0f, ew - kF16Top);
}

// Tile opening: from this lane's rows' factors (dZ1 = acc * factor) and input pair 2 kq, 2 kq + 1 -- the rows' z1
// fragments (the forward's), factors and 1 / x-scale to the wave's exchange, x~ to its [slot][row] block.
// acc_exponent: |acc| < 2^acc_exponent.
// Scales of the dW1 product.  Rows are summed INSIDE the instruction, so dZ1~[r] x~[r] must carry one common power of two
// S; but one power of two per operand for the whole wave would push the rows of a wave that holds one outlier (|d| 10^6
// times the others') to the bottom of fp16's range -- and where the outlier's own gate is closed those rows ARE the sum.
// So each row gets its own split: with g[r] = (exponent of the wave's largest |factor|) - (the row's), dZ1~[r] is raised
// by 2^(g/2) and x~[r] (the column of ones with it) lowered by the same -- written into the row's factor and x~ here, at
// no cost in the epilogue -- and both low planes are WIDE (f16_pair_scaled_wide: the residual times 2^11, the product
// rescaled when the parts are added), which keeps all 22 bits of an operand down to 2^-13: gaps up to 2^34 lose nothing.
struct Rows8Scales {
  float sd, inv_sum, inv_sd;  // dZ1's power of two; 1 / (x~'s times dZ1's); 1 / dZ1's (M slot 7, the ones: no x scale)
};
__device__ __forceinline__ Rows8Scales rows8_open(const float (&fac)[2], const float (&xs)[2][2], float inv_w1, int acc_exponent,
                                                  unsigned wl, int l16, int kq, u32x4 (&xf)[2]) {
  float most[2];
  int ex[2], ef[2];
#pragma unroll
  for (int rt = 0; rt < 2; ++rt) {
    most[rt] = rows8_row4(__builtin_fmaxf(__builtin_fabsf(xs[rt][0]), __builtin_fabsf(xs[rt][1])), false);
    ex[rt] = f16_bound_exponent(most[rt]);
    ef[rt] = f16_bound_exponent(__builtin_fabsf(fac[rt]));
    uint32_t hi, lo;
    f16_pair_scaled(xs[rt][0], xs[rt][1], __builtin_amdgcn_ldexpf(1.0f, kF16Top - ex[rt]), hi, lo);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t h = __shfl(hi, l16 + 16 * j, kWave), l = __shfl(lo, l16 + 16 * j, kWave);
      xf[rt][j] = (kq & 1) ? l : h;
    }
  }
  int exw = ex[0] > ex[1] ? ex[0] : ex[1], efw = ef[0] > ef[1] ? ef[0] : ef[1];
#pragma unroll
  for (int off = 8; off > 0; off >>= 1) {
    const int a = __shfl_xor(exw, off, kWave), b = __shfl_xor(efw, off, kWave);
    exw = a > exw ? a : exw;
    efw = b > efw ? b : efw;
  }
  const float k2048 = 2048.0f;
#pragma unroll
  for (int rt = 0; rt < 2; ++rt) {
    const int gap = efw - ef[rt], half = (gap > 48 ? 48 : gap) >> 1;
    if (kq == 0) {
      lds_write_b32(wl + kRows8FacOff + (16 * rt + l16) * 4, __builtin_amdgcn_ldexpf(fac[rt], half));
      lds_write_b32(wl + kRows8InvOff + (16 * rt + l16) * 4, __builtin_amdgcn_ldexpf(inv_w1, ex[rt] - kF16Top));
    }
    uint32_t hi, lo;
    f16_pair_scaled_wide(xs[rt][0], xs[rt][1], __builtin_amdgcn_ldexpf(1.0f, kF16Top - exw - half), k2048, hi, lo);
    if (kq == 3) {  // slot 7: the column of ones (db1), 2^-half as the row's x~ is; slot 15: nothing
      const uint32_t one = half <= 14 ? (uint32_t)(15 - half) << 10 : 1u << (24 - half);
      hi = (hi & 0xffffu) | (one << 16);
      lo &= 0xffffu;
    }
    const unsigned at = wl + kRows8XtOff + (2 * kq * 32 + 16 * rt + l16) * 2;
    lds_write_b16(at, hi);
    lds_write_b16_hi(at + 64, hi);
    lds_write_b16(at + 8 * 64, lo);
    lds_write_b16_hi(at + 9 * 64, lo);
  }
  Rows8Scales out;
  out.sd = __builtin_amdgcn_ldexpf(1.0f, kF16Top - acc_exponent - efw);
  out.inv_sum = __builtin_amdgcn_ldexpf(1.0f, (exw - kF16Top) + (efw + acc_exponent - kF16Top));
  out.inv_sd = __builtin_amdgcn_ldexpf(1.0f, efw + acc_exponent - kF16Top);
  return out;
}

// Epilogue: dZ1 = acc * factor * (h1 > 0), folded into the workgroup's running sums (see above).  `base`: LDS address
// of the W1 planes (b1, sums, counters behind them); `done`: column tiles this wave has published before this tile.
__device__ __forceinline__ void rows8_epilogue(const f32x4 (&acc)[2][16], const u32x4 (&xf)[2], unsigned wl, unsigned base,
                                               int wave, int done, Rows8Scales scales, int lane) {
  int lane_e = lane;  // (opaque copy: see the forward kernel's epilogue)
  asm volatile("" : "+v"(lane_e));
  const int l16e = lane_e & 15, kqe = lane_e >> 4;
  const unsigned planes_at = base + (kqe >> 1) * 256 + l16e * 16, b1_at = base + kRows8Planes + l16e * 4;
  const unsigned sums_at = base + kRows8Planes + 1024 + l16e * 32 + (kqe & 1) * 16;
  const unsigned prog_at = base + kRows8Planes + 1024 + kRows8Sums + 4 * wave;  // the predecessor's counter; own: + 4
  u32x4 fq[2], iq[2];
#pragma unroll
  for (int rt = 0; rt < 2; ++rt) {
    fq[rt] = rt == 0 ? lds_read_b128<kRows8FacOff>(wl + 16 * kqe) : lds_read_b128<kRows8FacOff + 64>(wl + 16 * kqe);
    iq[rt] = rt == 0 ? lds_read_b128<kRows8InvOff>(wl + 16 * kqe) : lds_read_b128<kRows8InvOff + 64>(wl + 16 * kqe);
  }
  const u32x2 xa0 = lds_read_b64<kRows8XtOff>(wl + l16e * 64 + kqe * 8), xa1 = lds_read_b64<kRows8XtOff + 32>(wl + l16e * 64 + kqe * 8);
  u32x4 w1f[2];
  uint32_t b1c[2];
  auto request_consts = [&](int ct, int set) {
    w1f[set] = lds_read_b128<0>(planes_at + ct * 512);
    b1c[set] = lds_read_b32_nowait<0>(b1_at + ct * 64);
  };
  request_consts(0, 0);
  u32x4 xa;
  const float k2048 = 2048.0f;
  const float c_first = kqe < 2 ? 1.0f : 0x1p-11f, c_second = kqe < 2 ? 0x1p-11f : 0x1p-22f;
#pragma unroll
  for (int ct = 0; ct < 16; ++ct) {
    const int set = ct & 1;
    uint32_t pr = lds_read_b32_nowait<0>(prog_at);
    u32x4 sq = lds_read_b128<0>(sums_at + ct * 512);
    if (ct + 1 < 16) request_consts(ct + 1, set ^ 1);
    if (ct + 1 < 16) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(w1f[set]), "+v"(b1c[set]));
    else asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(w1f[set]), "+v"(b1c[set]));
    if (ct == 0) {
      u32x2 a0 = xa0, a1 = xa1;
      asm volatile("" : "+v"(fq[0]), "+v"(fq[1]), "+v"(iq[0]), "+v"(iq[1]), "+v"(a0), "+v"(a1));
      xa = u32x4{a0[0], a0[1], a1[0], a1[1]};
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)  // the rows' factors times dZ1's power of two, once per tile: the plane split multiplies
#pragma unroll
        for (int r = 0; r < 4; ++r) fq[rt][r] = __float_as_uint(__uint_as_float(fq[rt][r]) * scales.sd);
    }
    const f32x4 zero = {0, 0, 0, 0};
    float dz[8];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      const f32x4 z = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, xf[rt]), __builtin_bit_cast(half8, w1f[set]), zero, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float pre = __builtin_fmaf(z[r], __uint_as_float(iq[rt][r]), __uint_as_float(b1c[set]));
        dz[4 * rt + r] = select_or_zero(positive_mask(pre), acc[rt][ct][r]);
      }
    }
    u32x4 dh, dl;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      uint32_t h, l;
      f16_pair_scaled2_wide(dz[2 * j], dz[2 * j + 1], __uint_as_float(fq[j >> 1][2 * (j & 1)]), __uint_as_float(fq[j >> 1][2 * (j & 1) + 1]), k2048, h, l);
      dh[j] = h;
      dl[j] = l;
    }
    // slots 0..7: x~hi . (Dhi | Dlo'), slots 8..15: x~lo' . (Dhi | Dlo'); the primes carry 2^11
    const f32x4 o1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, xa), __builtin_bit_cast(half8, dh), zero, 0, 0, 0);
    const f32x4 o2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, xa), __builtin_bit_cast(half8, dl), zero, 0, 0, 0);
    // The lanes of slots 8..15 hand their parts to those of slots 0..7.  One swap of (o1, o2) gives the lower lanes
    // (o1 own, o1 of the partner) and the upper lanes (o2 of the partner below, o2 own): each half weighs its pair
    // (1, 2^-11 below; 2^-11, 2^-22 above), a second swap brings the upper half's sum down.
    float v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const auto s1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(o1[r]), __float_as_uint(o2[r]), false, false);
      const float t = __builtin_fmaf(__uint_as_float(s1[1]), c_second, __uint_as_float(s1[0]) * c_first);
      const auto s2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(t), __float_as_uint(t), false, false);
      v[r] = __uint_as_float(s2[0]) + __uint_as_float(s2[1]);
    }
    // this block's sums as the predecessor left them: once its counter says so (polled inside ONE asm statement: a C++
    // loop per block costs the register allocator its grip on the whole kernel -- 936 bytes of scratch)
    const int need = done + ct + 1;
    {
      const unsigned sa = sums_at + ct * 512;
      int seen;
      if (ct + 1 < 16)
        asm volatile("s_waitcnt lgkmcnt(2)\n"
                     ".Lrows8_poll_%=:\n\t"
                     "v_readfirstlane_b32 %2, %0\n\t"
                     "s_nop 3\n\t"
                     "s_cmp_ge_i32 %2, %5\n\t"
                     "s_cbranch_scc1 .Lrows8_done_%=\n\t"
                     "ds_read_b32 %0, %3\n\t"
                     "ds_read_b128 %1, %4\n\t"
                     "s_waitcnt lgkmcnt(0)\n\t"
                     "s_branch .Lrows8_poll_%=\n"
                     ".Lrows8_done_%=:"
                     : "+v"(pr), "+v"(sq), "=&s"(seen)
                     : "v"(prog_at), "v"(sa), "s"(need)
                     : "scc", "memory");
      else
        asm volatile("s_waitcnt lgkmcnt(0)\n"
                     ".Lrows8_poll_%=:\n\t"
                     "v_readfirstlane_b32 %2, %0\n\t"
                     "s_nop 3\n\t"
                     "s_cmp_ge_i32 %2, %5\n\t"
                     "s_cbranch_scc1 .Lrows8_done_%=\n\t"
                     "ds_read_b32 %0, %3\n\t"
                     "ds_read_b128 %1, %4\n\t"
                     "s_waitcnt lgkmcnt(0)\n\t"
                     "s_branch .Lrows8_poll_%=\n"
                     ".Lrows8_done_%=:"
                     : "+v"(pr), "+v"(sq), "=&s"(seen)
                     : "v"(prog_at), "v"(sa), "s"(need)
                     : "scc", "memory");
    }
    if (kqe < 2) {
      u32x4 w;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        w[r] = __float_as_uint(__builtin_fmaf(v[r], r == 3 && kqe == 1 ? scales.inv_sd : scales.inv_sum, __uint_as_float(sq[r])));
      lds_write_b128<0>(sums_at + ct * 512, w);
    }
    if (lane_e == 0) lds_write_b32(prog_at + 4, __int_as_float(need));
    __builtin_amdgcn_sched_barrier(0);  // (a block's arithmetic stays in its block: sixteen z1 products hoisted are 64 registers)
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

template <int DIN, int NOUT, int RING>
__global__ __launch_bounds__(kBlock, 2) void mlp_rows_backward_gate_kernel(
    const float *__restrict__ x, const float *__restrict__ w1, const float *__restrict__ b1,
    const float *__restrict__ dout, int64_t m, const void *__restrict__ w2ts, float *__restrict__ partials,
    int partial_stride, int head_rows, const uint32_t *__restrict__ gate2, int d_in_arg) {
  constexpr bool kMma1 = DIN > 3;  // class 8: run-time d_in = 4..7, layer 1 on the matrix pipe (rows8_* above)
  constexpr int kIn = DIN;
  const int d_in = kMma1 ? d_in_arg : DIN;
  constexpr int kXRegs = kMma1 ? 2 : kIn;          // inputs a lane loads per row (class 8: the pair 2 kq, 2 kq + 1)
  constexpr int kRowLoads = 2 * (1 + kXRegs);      // a tile's row loads per lane
  constexpr int kTile = 128;
  constexpr int kAhead = RING - 1;
  constexpr int kRec = kMma1 ? 0 : rows_record(DIN);
  static_assert((DIN >= 1 && DIN <= 3) || DIN == 8, "width classes");
  static_assert((kMma1 ? rows8_dgrad_lds_bytes(RING, 0) : rows_dgrad_lds_bytes(RING, kIn)) <= 80 * 1024, "two workgroups per CU");
  static_assert(kAhead >= 2, "the mid-step barrier publishes a chunk requested at least a half-step earlier");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // [ring][layer-1 records [256][kRec]][per wave: gate block 1 KiB | factors [32] | observations [DIN][32] | sums [256][kRec]]
  const unsigned lds0 = lds_offset(smem);
  // (class 8: [ring][W1 planes | b1 | sums [256][8] | counters][per wave: gate block | factors | 1 / x scale | x~])
  constexpr int kRecOff = RING * kRowsChunk;
  constexpr int kWaveOff = kMma1 ? kRecOff + kRows8Planes + 1024 + kRows8Sums + 64 : kRecOff + kRec * kHidden * 4;
  constexpr int kWaveBytes = kMma1 ? kRows8Wave : 1024 + 32 * (1 + kIn) * 4 + kRec * kHidden * 4;
  constexpr int kFacOff = 1024, kObsOff = kFacOff + 32 * 4, kSumOff = kObsOff + 32 * kIn * 4;
  const int tid = threadIdx.x, lane = tid & 63, l16 = lane & 15, kq = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const __amdgpu_buffer_rsrc_t w2rsrc = buffer_rsrc(w2ts, kRowsPacked);
  const float inv_w2_scale = reinterpret_cast<const float *>(static_cast<const unsigned char *>(w2ts) + kRowsPacked)[1];
  const unsigned wave_lds = lds0 + kWaveOff + wave * kWaveBytes;

  const int64_t tiles = (m + kTile - 1) / kTile;
  const int64_t stride = gridDim.x;

  // layer-1 records (columns of this kernel = hidden units of layer 1) and zeroed running sums
  [[maybe_unused]] float inv_w1 = 1.0f;
  if constexpr (kMma1) {
    inv_w1 = rows8_constants(smem + kRecOff, reinterpret_cast<float *>(smem), w1, b1, d_in, tid);
  } else {
    float *rec = reinterpret_cast<float *>(smem + kRecOff);
    rec[tid * kRec] = b1[tid];
#pragma unroll
    for (int i = 0; i < kRec - 1; ++i) rec[tid * kRec + 1 + i] = i < kIn ? w1[tid * d_in + (i < kIn ? i : 0)] : 0.0f;
    float *sums = reinterpret_cast<float *>(smem + kWaveOff + wave * kWaveBytes + kSumOff);
    for (int idx = lane; idx < kRec * kHidden; idx += kWave) sums[idx] = 0.0f;
  }

  // rows of a tile that exist for this wave, and this lane's two rows' d = dOut[row][0] (0 past the end)
  auto wave_rows = [&](int64_t tile) {
    const int64_t left = tile < tiles ? m - tile * kTile - 32 * wave : 0;
    return left <= 0 ? 0 : left < 32 ? (int)left : 32;
  };
  auto load_rows = [&](float (&d)[2], float (&xs)[2][kXRegs], int64_t tile) {
    const int rows = wave_rows(tile);
    const int64_t r0 = tile * kTile + 32 * wave;
    const __amdgpu_buffer_rsrc_t drsrc = buffer_rsrc(rows > 0 ? dout + r0 * NOUT : dout, rows * NOUT * 4);
    const __amdgpu_buffer_rsrc_t xrsrc = buffer_rsrc(rows > 0 ? x + r0 * d_in : x, rows * d_in * 4);
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      d[rt] = buffer_load_f32(drsrc, (16 * rt + l16) * NOUT * 4, 0);
      if constexpr (kMma1) {  // (inputs past d_in: an offset past the descriptor's end, which reads as zero)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          xs[rt][j] = buffer_load_f32(xrsrc, 2 * kq + j < d_in ? ((16 * rt + l16) * d_in + 2 * kq + j) * 4 : 0x7ffffff0, 0);
      } else {
#pragma unroll
        for (int i = 0; i < kIn; ++i) xs[rt][i] = buffer_load_f32(xrsrc, ((16 * rt + l16) * d_in + i) * 4, 0);
      }
    }
  };
  // the wave's gate block of `tile` -> its LDS block (rows past the end arrive as zeros: gate closed)
  auto request_gate = [&](int64_t tile) {
    const int rows = wave_rows(tile);
    const __amdgpu_buffer_rsrc_t rsrc = buffer_rsrc(rows > 0 ? gate2 + (tile * kTile + 32 * wave) * 8 : gate2, rows * 32);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, smem + kWaveOff + wave * kWaveBytes, 16, lane * 16, 0, 0, 0);
  };
  // chunk hs = (S, C): k steps 2 S, 2 S + 1 of the pack, column tiles (of 32) 4 C .. 4 C + 3, both planes: two runs of 8 KiB
  auto chunk_piece = [&](int hs, int stage, int piece) {
    const int S = hs >> 1, C = hs & 1;
    const int src = ((2 * S + (piece >> 3)) * 8 + 4 * C) * 2048 + (piece & 7) * 1024;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(w2rsrc, smem + stage * kRowsChunk + piece * 1024, 16, lane * 16, src, 0, 0);
  };
  // this lane's 16-byte unit of local column tile ctl (of 16), plane p: k step kq / 2, k half kq & 1, column tile ctl / 2
  const unsigned b_lane = lds0 + (kq >> 1) * 8192 + ((kq & 1) * 32 + l16) * 16;
  const unsigned g_lane = wave_lds + l16 * 32 + kq;  // the gate byte of k block S: + 4 S (+ 512 for row tile 1)

  float dc[2], dn[2], xc[2][kXRegs], xn[2][kXRegs];  // this tile's / the next tile's d and observations of this lane's rows
  [[maybe_unused]] u32x4 xf[2];         // class 8: the rows' z1 fragments
  [[maybe_unused]] Rows8Scales scales8 = {1.0f, 1.0f, 1.0f};
  [[maybe_unused]] int done8 = 0;       // class 8: column tiles this wave has published
  f32x4 acc[2][16];
  u32x4 a_g[2][2];     // gate fragments: [set = k block parity][row tile]
  u32x4 bh[2], bl[2];  // B fragments of local column tile ctl: set ctl % 2, requested one slot ahead
  auto request_block = [&](unsigned br, int ctl, int set) {
    bh[set] = ctl == 0   ? lds_read_b128<0>(br)
              : ctl == 1 ? lds_read_b128<256>(br)
              : ctl == 2 ? lds_read_b128<2048>(br)
              : ctl == 3 ? lds_read_b128<2048 + 256>(br)
              : ctl == 4 ? lds_read_b128<4096>(br)
              : ctl == 5 ? lds_read_b128<4096 + 256>(br)
              : ctl == 6 ? lds_read_b128<6144>(br)
                         : lds_read_b128<6144 + 256>(br);
    bl[set] = ctl == 0   ? lds_read_b128<1024>(br)
              : ctl == 1 ? lds_read_b128<1024 + 256>(br)
              : ctl == 2 ? lds_read_b128<3072>(br)
              : ctl == 3 ? lds_read_b128<3072 + 256>(br)
              : ctl == 4 ? lds_read_b128<5120>(br)
              : ctl == 5 ? lds_read_b128<5120 + 256>(br)
              : ctl == 6 ? lds_read_b128<7168>(br)
                         : lds_read_b128<7168 + 256>(br);
  };
  // bits -> fp16 1.0 / 0.0 (element 0 in the low half)
  auto gate_fragment = [&](uint32_t byte) {
    u32x4 g;
#pragma unroll
    for (int e = 0; e < 8; e += 2)
      g[e >> 1] = (((byte >> e) & 1u) ? 0x00003c00u : 0u) | (((byte >> (e + 1)) & 1u) ? 0x3c000000u : 0u);
    return g;
  };

  // One half-step (see the forward kernel): KIND 0 produces row tile C of block S + 1 (its gate byte is requested in
  // slot 0 and used from slot 2 on); 2 = (7, 0): nothing produced, and the NEXT tile's gate block is requested (the
  // block's last reader was half-step (6, 1)); 3 = (7, 1): nothing requested for a successor.  The barriers of (7, 0)
  // and (7, 1) leave one more vector-memory operation in flight: that gate block, younger than the chunk they publish.
  auto do_half = [&](auto first_tag, auto cur_tag, auto c_tag, auto kind_tag, int hs, int stage, int64_t tile) {
    constexpr bool FIRST = decltype(first_tag)::value;
    constexpr int CUR = decltype(cur_tag)::value;
    constexpr int C = decltype(c_tag)::value;
    constexpr int KIND = decltype(kind_tag)::value;
    constexpr bool kProduce = KIND == 0, kEnd = KIND == 3;
    const int stage_next = stage + 1 == RING ? 0 : stage + 1, stage_free = stage == 0 ? RING - 1 : stage - 1;
    const unsigned br = b_lane + stage * kRowsChunk, br_next = b_lane + stage_next * kRowsChunk;
    const int S = hs >> 1;
    [[maybe_unused]] uint32_t byte = 0;
#pragma unroll
    for (int sl = 0; sl < 8; ++sl) {
      const int set = sl & 1, ahead = set ^ 1, ct = 8 * C + sl;
      if (kProduce && sl == 0) byte = C == 0 ? lds_read_u8<0>(g_lane + 4 * (S + 1)) : lds_read_u8<512>(g_lane + 4 * (S + 1));
      if (KIND == 2 && sl == 0) request_gate(tile + stride);
      if (sl < 7) request_block(br, sl + 1, ahead);
      else if (!kEnd) request_block(br_next, 0, ahead);
      {
        const int allowed = ((kProduce && sl == 0) ? 1 : 0) + ((kEnd && sl == 7) ? 0 : 2);
        allowed == 0 ? wait_lds<0>(bh[set], bl[set]) : allowed == 2 ? wait_lds<2>(bh[set], bl[set]) : wait_lds<3>(bh[set], bl[set]);
      }
      const f32x4 zero = {0, 0, 0, 0};
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, a_g[CUR][rt]), __builtin_bit_cast(half8, bl[set]),
                                                             FIRST ? zero : acc[rt][ct], 0, 0, 0);
        acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, a_g[CUR][rt]), __builtin_bit_cast(half8, bh[set]),
                                                             acc[rt][ct], 0, 0, 0);
      }
      if (kProduce && sl == 2) {  // (the byte is older than slot 1's fragment request: landed behind that slot's wait)
        asm volatile("" : "+v"(byte));
        a_g[CUR ^ 1][C] = gate_fragment(byte);
      }
      if (sl == 3) {
        // Younger than the chunk this barrier publishes (chunk hs + 1, whose pieces went out in half-step hs + 1 - kAhead
        // behind THAT half-step's barrier): the pieces of kAhead - 2 chunks, plus
        //   the gate block, requested in slot 0 of half-step 14: behind the pieces of half-step 15 - kAhead (barrier of
        //     half-step 14) always, behind those of half-step 16 - kAhead (barrier of half-step 15) only if kAhead >= 3;
        //   the next tile's row loads (2 (1 + DIN) instructions at the tile's opening): behind the pieces of the previous
        //     tile's half-step 17 - kAhead (barrier of half-step 0) always, behind those published in half-step 1 only if
        //     they too went out in the previous tile (kAhead >= 3; with kAhead = 2 they follow the row loads).
        // (Round 3, found by an intermittent 5e-6 in dW1 at d_in = 3: the three-chunk ring counted the gate block in
        // half-step 15 and the row loads in half-step 1 as well -- one piece, resp. all four, of the chunk being published
        // were allowed to be still in flight.  The four-chunk ring of d_in = 1 was right.)
        constexpr int kExtra = KIND == 2                 ? 1
                               : KIND == 3               ? (kAhead >= 3 ? 1 : 0)
                               : (FIRST && (C == 0 || kAhead >= 3)) ? kRowLoads
                                                         : 0;
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(4 * (kAhead - 2) + kExtra) : "memory");
      }
      if (sl >= 4) chunk_piece((hs + kAhead) & (kRowsHalfSteps - 1), stage_free, wave * 4 + (sl - 4));
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  using T = std::true_type;
  using F = std::false_type;
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using K0 = I0;
  using K2 = std::integral_constant<int, 2>;
  using K3 = std::integral_constant<int, 3>;

  // ---- prologue: the first tile's gate block, then the first kAhead chunks; everything before them landed --------------
  __syncthreads();  // (records and zeroed sums in LDS)
  if ((int64_t)blockIdx.x < tiles) {
    request_gate(blockIdx.x);
#pragma unroll
    for (int d = 0; d < kAhead; ++d)
#pragma unroll
      for (int u = 0; u < 4; ++u) chunk_piece(d, d, wave * 4 + u);
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(4 * (kAhead - 1)) : "memory");
    load_rows(dc, xc, blockIdx.x);
  }

  int stage = 0;
  auto next_stage = [&]() { stage = stage + 1 == RING ? 0 : stage + 1; };
  // INVARIANT (class 8, rows8_epilogue): the trip count of this loop is uniform over the workgroup (it depends on
  // blockIdx.x alone) and EVERY one of the four waves runs rows8_epilogue exactly once per tile, publishing sixteen
  // counters -- waves whose rows lie past m included (their rows carry zero factors).  Wave w's poll of wave w - 1's
  // counter is unbounded: a wave-conditional tile loop or epilogue (skipping empty tail waves, say) turns it into a hang,
  // not a wrong number.  tests/test_mlp_split_gpu.py::test_class8_chain_with_empty_waves_and_one_workgroup runs m < 32,
  // m % 128 != 0 and a one-workgroup grid (RL8_MLP_GRID_CAP=1: the counters reach 16 x tiles).
  for (int64_t tile = blockIdx.x; tile < tiles; tile += stride) {
    // ---- open the tile: this tile's gate block has landed (requested two half-steps and an epilogue ago: only the
    // eight pieces behind it may still be in flight); block 0's gate fragments; factors and observations of the wave's
    // rows to the per-wave exchange; the first B fragments
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    load_rows(dn, xn, tile + stride);  // the next tile's rows: in flight through this tile (counted in its first two barriers)
    {
      uint32_t b0 = lds_read_u8<0>(g_lane), b1v = lds_read_u8<512>(g_lane);
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(b0), "+v"(b1v));
      a_g[0][0] = gate_fragment(b0);
      a_g[0][1] = gate_fragment(b1v);
      if constexpr (kMma1) {
        const float fac[2] = {dc[0] * inv_w2_scale, dc[1] * inv_w2_scale};
        scales8 = rows8_open(fac, xc, inv_w1, 22, wave_lds, l16, kq, xf);  // |acc| < 256 x 2^14: gate x planes below 2^14
      } else if (kq == 0) {
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
          lds_write_b32(wave_lds + kFacOff + (16 * rt + l16) * 4, dc[rt] * inv_w2_scale);
#pragma unroll
          for (int i = 0; i < kIn; ++i) lds_write_b32(wave_lds + kObsOff + (32 * i + 16 * rt + l16) * 4, xc[rt][i]);
        }
      }
      request_block(b_lane + stage * kRowsChunk, 0, 0);
    }
    do_half(T{}, I0{}, I0{}, K0{}, 0, stage, tile);  next_stage();
    do_half(T{}, I0{}, I1{}, K0{}, 1, stage, tile);  next_stage();
    do_half(F{}, I1{}, I0{}, K0{}, 2, stage, tile);  next_stage();
    do_half(F{}, I1{}, I1{}, K0{}, 3, stage, tile);  next_stage();
#pragma unroll 1
    for (int hs = 4; hs < kRowsHalfSteps - 4; hs += 4) {
      do_half(F{}, I0{}, I0{}, K0{}, hs, stage, tile);      next_stage();
      do_half(F{}, I0{}, I1{}, K0{}, hs + 1, stage, tile);  next_stage();
      do_half(F{}, I1{}, I0{}, K0{}, hs + 2, stage, tile);  next_stage();
      do_half(F{}, I1{}, I1{}, K0{}, hs + 3, stage, tile);  next_stage();
    }
    do_half(F{}, I0{}, I0{}, K0{}, kRowsHalfSteps - 4, stage, tile);  next_stage();
    do_half(F{}, I0{}, I1{}, K0{}, kRowsHalfSteps - 3, stage, tile);  next_stage();
    do_half(F{}, I1{}, I0{}, K2{}, kRowsHalfSteps - 2, stage, tile);  next_stage();
    do_half(F{}, I1{}, I1{}, K3{}, kRowsHalfSteps - 1, stage, tile);  next_stage();

    // ---- epilogue: dZ1 = dH1 * factor * (h1 > 0), folded into the wave's running column sums ---------------------------
    if constexpr (kMma1) {
      rows8_epilogue(acc, xf, lds0 + kWaveOff + wave * kWaveBytes, lds0 + kRecOff, wave, done8, scales8, lane);
      done8 += 16;
    } else {
      // this lane: columns 16 ct + l16; rows 16 rt + 4 kq + r.  Factors / observations of those rows from the exchange.
      int lane_e = lane;  // (opaque copy: see the forward kernel's epilogue)
      asm volatile("" : "+v"(lane_e));
      const int l16e = lane_e & 15, kqe = lane_e >> 4;
      const unsigned wl = lds0 + kWaveOff + wave * kWaveBytes;
      // (narrow observations: dZ1 = a x factor is never formed -- the sums take a with the factor, resp. factor x
      // observation, as the fma's multiplier: one instruction per element less; d_in = 3 has no registers for 24 products)
      constexpr bool kFoldFactor = kIn <= 2;
      [[maybe_unused]] float fx[2][kFoldFactor ? kIn : 1][4];
      u32x4 fq[2], xq[2][kIn];
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        fq[rt] = rt == 0 ? lds_read_b128<kFacOff>(wl + 16 * kqe) : lds_read_b128<kFacOff + 64>(wl + 16 * kqe);
#pragma unroll
        for (int i = 0; i < kIn; ++i)
          xq[rt][i] = rt == 0 ? (i == 0   ? lds_read_b128<kObsOff>(wl + 16 * kqe)
                                 : i == 1 ? lds_read_b128<kObsOff + 128>(wl + 16 * kqe)
                                          : lds_read_b128<kObsOff + 256>(wl + 16 * kqe))
                              : (i == 0   ? lds_read_b128<kObsOff + 64>(wl + 16 * kqe)
                                 : i == 1 ? lds_read_b128<kObsOff + 128 + 64>(wl + 16 * kqe)
                                          : lds_read_b128<kObsOff + 256 + 64>(wl + 16 * kqe));
      }
      // per column tile: the column's layer-1 record and its running sums (read-modify-write by the lanes kq == ct % 4)
      const unsigned rec_at = lds0 + kRecOff + l16e * (kRec * 4), sum_at = wl + kSumOff + l16e * (kRec * 4);
      typedef typename std::conditional<kRec == 2, u32x2, u32x4>::type rec_t;
      rec_t rq[2], sq[2];
      auto request_col = [&](int ct, int set) {
        const unsigned ra = rec_at + ct * (16 * kRec * 4), sa = sum_at + ct * (16 * kRec * 4);
        if constexpr (kRec == 2) {
          rq[set] = lds_read_b64<0>(ra);
          sq[set] = lds_read_b64<0>(sa);
        } else {
          rq[set] = lds_read_b128<0>(ra);
          sq[set] = lds_read_b128<0>(sa);
        }
      };
      request_col(0, 0);
#pragma unroll
      for (int ct = 0; ct < 16; ++ct) {
        const int set = ct & 1;
        if (ct + 1 < 16) request_col(ct + 1, set ^ 1);
        if (ct + 1 < 16) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(rq[set]), "+v"(sq[set]));
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(rq[set]), "+v"(sq[set]));
        if (ct == 0) {
#pragma unroll
          for (int rt = 0; rt < 2; ++rt) {
            asm volatile("" : "+v"(fq[rt]));
#pragma unroll
            for (int i = 0; i < kIn; ++i) asm volatile("" : "+v"(xq[rt][i]));
          }
          if constexpr (kFoldFactor) {  // factor x observation per row, once per tile: one fma per sum and element below
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
              for (int i = 0; i < kIn; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) fx[rt][i][r] = __uint_as_float(fq[rt][r]) * __uint_as_float(xq[rt][i][r]);
          }
        }
        const float b1c = __uint_as_float(rq[set][0]);
        float w1c[kIn];
#pragma unroll
        for (int i = 0; i < kIn; ++i) w1c[i] = __uint_as_float(rq[set][1 + i]);
        float db = 0.0f, dw[kIn];
#pragma unroll
        for (int i = 0; i < kIn; ++i) dw[i] = 0.0f;
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
          float pre[4];
          unsigned long long open[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            pre[r] = b1c;
#pragma unroll
            for (int i = 0; i < kIn; ++i) pre[r] = __builtin_fmaf(__uint_as_float(xq[rt][i][r]), w1c[i], pre[r]);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) open[r] = positive_mask(pre[r]);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if constexpr (kFoldFactor) {
              const float a = select_or_zero(open[r], acc[rt][ct][r]);
              db = __builtin_fmaf(a, __uint_as_float(fq[rt][r]), db);
#pragma unroll
              for (int i = 0; i < kIn; ++i) dw[i] = __builtin_fmaf(a, fx[rt][i][r], dw[i]);
            } else {
              const float dz = select_or_zero(open[r], acc[rt][ct][r]) * __uint_as_float(fq[rt][r]);
              db += dz;
#pragma unroll
              for (int i = 0; i < kIn; ++i) dw[i] = __builtin_fmaf(dz, __uint_as_float(xq[rt][i][r]), dw[i]);
            }
          }
        }
        // the four lanes of a column (kq = 0..3) in a fixed order, then the running sums by one of them
        auto across = [&](float v) {
          const auto s16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
          const float u = __uint_as_float(s16[0]) + __uint_as_float(s16[1]);
          const auto s32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(u), __float_as_uint(u), false, false);
          return __uint_as_float(s32[0]) + __uint_as_float(s32[1]);
        };
        db = across(db);
#pragma unroll
        for (int i = 0; i < kIn; ++i) dw[i] = across(dw[i]);
        if (kqe == (ct & 3)) {
          const unsigned sa = sum_at + ct * (16 * kRec * 4);
          lds_write_b32(sa, __uint_as_float(sq[set][0]) + db);
#pragma unroll
          for (int i = 0; i < kIn; ++i) lds_write_b32(sa + 4 + 4 * i, __uint_as_float(sq[set][1 + i]) + dw[i]);
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    // the workgroup's next tile
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      dc[rt] = dn[rt];
#pragma unroll
      for (int i = 0; i < kXRegs; ++i) xc[rt][i] = xn[rt][i];
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();

  // Workgroup partial row: [dW1 (256 * d_in) | db1 (256) | head gradients (the weight-gradient kernel's)], the four
  // waves' sums added in wave order.
  float *row = partials + (int64_t)blockIdx.x * partial_stride;
  {
    const int t = tid;
    if constexpr (kMma1) {  // the workgroup's one array: [unit][dW1 row 0..6 | db1]
      const float *sums = reinterpret_cast<const float *>(smem + kRecOff + kRows8Planes + 1024);
      row[kHidden * d_in + t] = sums[t * 8 + 7];
#pragma unroll
      for (int i = 0; i < 7; ++i)
        if (i < d_in) row[t * d_in + i] = sums[t * 8 + i];
    } else {
      float tot[1 + kIn];
#pragma unroll
      for (int i = 0; i < 1 + kIn; ++i) tot[i] = 0.0f;
      for (int w = 0; w < 4; ++w) {
        const float *sums = reinterpret_cast<const float *>(smem + kWaveOff + w * kWaveBytes + kSumOff);
#pragma unroll
        for (int i = 0; i < 1 + kIn; ++i) tot[i] += sums[t * kRec + i];
      }
      row[kHidden * d_in + t] = tot[0];
#pragma unroll
      for (int i = 0; i < kIn; ++i) row[t * d_in + i] = tot[1 + i];
    }
    // the head-gradient segments of the first head_rows rows belong to the weight-gradient kernel; rows beyond them are zero
    if ((int)blockIdx.x >= head_rows)
      for (int idx = kHidden * d_in + kHidden + t; idx < partial_stride; idx += kBlock) row[idx] = 0.0f;
  }
}

template <int DIN, int NOUT>
static int launch_rows_backward_gate(int grid, hipStream_t s, const float *x, const float *w1, const float *b1, const float *dout,
                                     int64_t m, const void *w2ts, float *partials, int stride, int head_rows,
                                     const uint32_t *gate2, int d_in) {
  constexpr int kRing = DIN == 1 ? 4 : 3;
  auto kernel = &mlp_rows_backward_gate_kernel<DIN, NOUT, kRing>;
  static LdsOptIn lds_attr_set_0;
  if (const int e_lds_attr_set_0 = allow_dynamic_lds(lds_attr_set_0, reinterpret_cast<const void *>(kernel), 160 * 1024)) return e_lds_attr_set_0;
  constexpr int kLds = DIN > 3 ? rows8_dgrad_lds_bytes(kRing, 0) : rows_dgrad_lds_bytes(kRing, DIN);
  kernel<<<grid, kBlock, kLds, s>>>(x, w1, b1, dout, m, w2ts, partials, stride, head_rows, gate2, d_in);
  return launch_status();
}

// The gate-mode data gradient behind rl8_mlp_tower_backward_gate_f16_f32 for d_in <= 3 (-1: no variant: the caller keeps
// the previous kernel).  Same grid and partial rows as that kernel.
int mlp_rows_backward_gate_dispatch(int grid, hipStream_t s, const float *x, const float *w1, const float *b1, const float *dout,
                                    int64_t m, int d_in, const void *w2ts, int n_out, float *partials, int stride, int head_rows,
                                    const uint32_t *gate2) {
  const int dc = d_in <= 3 ? d_in : d_in <= 7 ? 8 : 0;  // (class 8 keeps M slot 7 for db1: seven inputs)
#define RL8_ROWS_BWD(D, N) \
  if (dc == D && n_out == N) return launch_rows_backward_gate<D, N>(grid, s, x, w1, b1, dout, m, w2ts, partials, stride, head_rows, gate2, d_in);
  RL8_ROWS_BWD(1, 1) RL8_ROWS_BWD(1, 2) RL8_ROWS_BWD(2, 1) RL8_ROWS_BWD(2, 2) RL8_ROWS_BWD(3, 1) RL8_ROWS_BWD(3, 2)
  RL8_ROWS_BWD(8, 1) RL8_ROWS_BWD(8, 2)
#undef RL8_ROWS_BWD
  return -1;
}

// ---- data gradient of GENERAL heads in the same structure (round 5) ------------------------
// dZ2[s][k] = G[s][k] * sum_o dOut[s][o] W3[o][k] has no rank-one form, so the A operand is what the forward's is: TWO
// fp16 planes, here of dZ2 scaled by a power of two per row (bound: sum_o |dOut[s][o]| max_k |W3[o][k]|), produced by the
// wave in fragment layout one element per slot -- the forward's producer with W3's column k as the "layer-1 record"
// ([256][KOUT] floats in LDS), the row's dOut as its observations, no bias, and the gate BIT of h2 where the forward has
// the ReLU -- and three plane products per fragment pair against the planes of W2^T (rl8_mlp_pack_w2_f16, transposed).
// Everything else is the gate-mode kernel above: the product is not transposed, B arrives in the 32x32x16 unit order of
// the shared pack, the wave's gate block comes by one direct-to-LDS load per tile, row factors and observations go
// through the per-wave exchange, running column sums [256][db1 | dW1 row] per wave in LDS.  d_in <= 3 (the sums' LDS);
// n_out <= KOUT in {2, 4} at run time (rows of W3 past n_out are zero records, dOut past n_out reads as zero).
constexpr int rows_dgrad_general_lds_bytes(int ring, int k_in, int k_out) {
  return rows_dgrad_lds_bytes(ring, k_in) + k_out * kHidden * 4;
}

template <int DIN, int KOUT, int RING>
__global__ __launch_bounds__(kBlock, 2) void mlp_rows_backward_general_kernel(
    const float *__restrict__ x, const float *__restrict__ w1, const float *__restrict__ b1,
    const float *__restrict__ dout, int64_t m, const void *__restrict__ w2ts, const float *__restrict__ w3,
    float *__restrict__ partials, int partial_stride, int head_rows, const uint32_t *__restrict__ gate2, int n_out,
    int d_in_arg) {
  constexpr bool kMma1 = DIN > 3;  // class 8: run-time d_in = 4..7, layer 1 on the matrix pipe (rows8_* above)
  constexpr int kIn = DIN;
  const int d_in = kMma1 ? d_in_arg : DIN;
  constexpr int kXRegs = kMma1 ? 2 : kIn;
  constexpr int kTile = 128;
  constexpr int kAhead = RING - 1;
  constexpr int kRec = kMma1 ? 0 : rows_record(DIN);
  static_assert((DIN >= 1 && DIN <= 3) || DIN == 8, "width classes");
  static_assert(KOUT == 2 || KOUT == 4, "W3 records of 8 or 16 bytes");
  static_assert((kMma1 ? rows8_dgrad_lds_bytes(RING, KOUT) : rows_dgrad_general_lds_bytes(RING, kIn, KOUT)) <= 80 * 1024, "two workgroups per CU");
  static_assert(kAhead >= 2, "the mid-step barrier publishes a chunk requested at least a half-step earlier");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // [ring][layer-1 records [256][kRec]][W3 records [256][KOUT]][per wave: gate block | factors [32] | observations [DIN][32] | sums [256][kRec]]
  const unsigned lds0 = lds_offset(smem);
  constexpr int kRecOff = RING * kRowsChunk;
  // (class 8: [ring][W1 planes | b1 | sums [256][8] | counters][W3 records][per wave: gate block | factors | 1 / x scale | x~])
  constexpr int kW3Off = kRecOff + (kMma1 ? kRows8Planes + 1024 + kRows8Sums + 64 : kRec * kHidden * 4);
  constexpr int kWaveOff = kW3Off + KOUT * kHidden * 4;
  constexpr int kWaveBytes = kMma1 ? kRows8Wave : 1024 + 32 * (1 + kIn) * 4 + kRec * kHidden * 4;
  constexpr int kFacOff = 1024, kObsOff = kFacOff + 32 * 4, kSumOff = kObsOff + 32 * kIn * 4;
  constexpr int kFar = 0x7fffff00;  // beyond every descriptor: reads as zero
  const int tid = threadIdx.x, lane = tid & 63, l16 = lane & 15, kq = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const __amdgpu_buffer_rsrc_t w2rsrc = buffer_rsrc(w2ts, kRowsPacked);
  const float inv_w2_scale = reinterpret_cast<const float *>(static_cast<const unsigned char *>(w2ts) + kRowsPacked)[1];
  const unsigned wave_lds = lds0 + kWaveOff + wave * kWaveBytes;

  const int64_t tiles = (m + kTile - 1) / kTile;
  const int64_t stride = gridDim.x;

  // layer-1 records, W3 records (column k of W3: [k][KOUT]), zeroed running sums; max_k |W3[o][k]| over the workgroup
  float w3max[KOUT];
  [[maybe_unused]] float inv_w1 = 1.0f;
  if constexpr (kMma1) inv_w1 = rows8_constants(smem + kRecOff, reinterpret_cast<float *>(smem) + 16, w1, b1, d_in, tid);
  {
    if constexpr (!kMma1) {
      float *rec = reinterpret_cast<float *>(smem + kRecOff);
      rec[tid * kRec] = b1[tid];
#pragma unroll
      for (int i = 0; i < kRec - 1; ++i) rec[tid * kRec + 1 + i] = i < kIn ? w1[tid * d_in + (i < kIn ? i : 0)] : 0.0f;
    }
    float *w3r = reinterpret_cast<float *>(smem + kW3Off);
    float mine[KOUT];
#pragma unroll
    for (int q = 0; q < KOUT; ++q) {
      const float v = q < n_out ? w3[q * kHidden + tid] : 0.0f;
      w3r[tid * KOUT + q] = v;
      mine[q] = __builtin_fabsf(v);
    }
    if constexpr (!kMma1) {
      float *sums = reinterpret_cast<float *>(smem + kWaveOff + wave * kWaveBytes + kSumOff);
      for (int idx = lane; idx < kRec * kHidden; idx += kWave) sums[idx] = 0.0f;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
#pragma unroll
      for (int q = 0; q < KOUT; ++q) mine[q] = __builtin_fmaxf(mine[q], __shfl_xor(mine[q], off, 64));
    float *red = reinterpret_cast<float *>(smem);  // [wave][KOUT] (the ring's first bytes: nothing is requested into it yet)
    if (lane == 0) {
#pragma unroll
      for (int q = 0; q < KOUT; ++q) red[wave * KOUT + q] = mine[q];
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < KOUT; ++q) {
      const float mx = __builtin_fmaxf(__builtin_fmaxf(red[q], red[KOUT + q]), __builtin_fmaxf(red[2 * KOUT + q], red[3 * KOUT + q]));
      w3max[q] = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(mx)));
    }
  }

  auto wave_rows = [&](int64_t tile) {
    const int64_t left = tile < tiles ? m - tile * kTile - 32 * wave : 0;
    return left <= 0 ? 0 : left < 32 ? (int)left : 32;
  };
  // this lane's two rows: dOut[row][0 .. n_out) (zeros past n_out and past the end) and the observations: ALWAYS
  // 2 (KOUT + DIN) load instructions (the barriers count them)
  auto load_rows = [&](float (&d)[2][KOUT], float (&xs)[2][kXRegs], int64_t tile) {
    const int rows = wave_rows(tile);
    const int64_t r0 = tile * kTile + 32 * wave;
    const __amdgpu_buffer_rsrc_t drsrc = buffer_rsrc(rows > 0 ? dout + r0 * n_out : dout, rows * n_out * 4);
    const __amdgpu_buffer_rsrc_t xrsrc = buffer_rsrc(rows > 0 ? x + r0 * d_in : x, rows * d_in * 4);
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
#pragma unroll
      for (int q = 0; q < KOUT; ++q) d[rt][q] = buffer_load_f32(drsrc, q < n_out ? ((16 * rt + l16) * n_out + q) * 4 : kFar, 0);
      if constexpr (kMma1) {  // the pair 2 kq, 2 kq + 1 (inputs past d_in read as zero)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          xs[rt][j] = buffer_load_f32(xrsrc, 2 * kq + j < d_in ? ((16 * rt + l16) * d_in + 2 * kq + j) * 4 : kFar, 0);
      } else {
#pragma unroll
        for (int i = 0; i < kIn; ++i) xs[rt][i] = buffer_load_f32(xrsrc, ((16 * rt + l16) * d_in + i) * 4, 0);
      }
    }
  };
  auto request_gate = [&](int64_t tile) {
    const int rows = wave_rows(tile);
    const __amdgpu_buffer_rsrc_t rsrc = buffer_rsrc(rows > 0 ? gate2 + (tile * kTile + 32 * wave) * 8 : gate2, rows * 32);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, smem + kWaveOff + wave * kWaveBytes, 16, lane * 16, 0, 0, 0);
  };
  auto chunk_piece = [&](int hs, int stage, int piece) {
    const int S = hs >> 1, C = hs & 1;
    const int src = ((2 * S + (piece >> 3)) * 8 + 4 * C) * 2048 + (piece & 7) * 1024;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(w2rsrc, smem + stage * kRowsChunk + piece * 1024, 16, lane * 16, src, 0, 0);
  };
  const unsigned b_lane = lds0 + (kq >> 1) * 8192 + ((kq & 1) * 32 + l16) * 16;
  const unsigned g_lane = wave_lds + l16 * 32 + kq;              // the gate byte of k block S: + 4 S (+ 512 for row tile 1)
  const unsigned c_lane = lds0 + kW3Off + kq * (8 * KOUT * 4);   // this lane's eight W3 records of block S: + 32 KOUT 4 S

  float dc[2][KOUT], xc[2][kXRegs];  // this tile's dOut rows and observations of this lane's rows
  [[maybe_unused]] u32x4 xf[2];         // class 8: the rows' z1 fragments
  [[maybe_unused]] Rows8Scales scales8 = {1.0f, 1.0f, 1.0f};
  [[maybe_unused]] int done8 = 0;       // class 8: column tiles this wave has published
  float sc[2];          // power of two that places this tile's dZ2 rows in fp16's range
  f32x4 acc[2][16];
  u32x4 a_hi[2][2], a_lo[2][2];  // dZ2 fragments: [set = k block parity][row tile]
  u32x4 bh[2], bl[2];
  uint32_t gbyte[2];    // gate bytes of the block being produced: [row tile]
  auto request_block = [&](unsigned br, int ctl, int set) {
    bh[set] = ctl == 0   ? lds_read_b128<0>(br)
              : ctl == 1 ? lds_read_b128<256>(br)
              : ctl == 2 ? lds_read_b128<2048>(br)
              : ctl == 3 ? lds_read_b128<2048 + 256>(br)
              : ctl == 4 ? lds_read_b128<4096>(br)
              : ctl == 5 ? lds_read_b128<4096 + 256>(br)
              : ctl == 6 ? lds_read_b128<6144>(br)
                         : lds_read_b128<6144 + 256>(br);
    bl[set] = ctl == 0   ? lds_read_b128<1024>(br)
              : ctl == 1 ? lds_read_b128<1024 + 256>(br)
              : ctl == 2 ? lds_read_b128<3072>(br)
              : ctl == 3 ? lds_read_b128<3072 + 256>(br)
              : ctl == 4 ? lds_read_b128<5120>(br)
              : ctl == 5 ? lds_read_b128<5120 + 256>(br)
              : ctl == 6 ? lds_read_b128<7168>(br)
                         : lds_read_b128<7168 + 256>(br);
  };
  // W3 records: KOUT = 2: one 16-byte read holds an element PAIR (requested in the odd slot in front of it, or at e = 0);
  // KOUT = 4: one read per element.  Two register sets.
  u32x4 cq[2];
  auto request_record = [&](int S, int e) {
    const unsigned a = c_lane + S * (32 * KOUT * 4);
    if constexpr (KOUT == 2) {  // e even: the pair (e, e + 1)
      const int set = (e >> 1) & 1;
      cq[set] = e == 0 ? lds_read_b128<0>(a) : e == 2 ? lds_read_b128<16>(a) : e == 4 ? lds_read_b128<32>(a) : lds_read_b128<48>(a);
    } else {
      const int set = e & 1;
      cq[set] = e == 0   ? lds_read_b128<0>(a)
                : e == 1 ? lds_read_b128<16>(a)
                : e == 2 ? lds_read_b128<32>(a)
                : e == 3 ? lds_read_b128<48>(a)
                : e == 4 ? lds_read_b128<64>(a)
                : e == 5 ? lds_read_b128<80>(a)
                : e == 6 ? lds_read_b128<96>(a)
                         : lds_read_b128<112>(a);
    }
  };
  auto request_byte = [&](int S, int C) {  // gate byte of block S, row tile C
    gbyte[C] = C == 0 ? lds_read_u8<0>(g_lane + 4 * S) : lds_read_u8<512>(g_lane + 4 * S);
  };
  // dZ2 of element e (k = 32 S + 8 kq + e) of row tile rt, behind a wait that covers its record and the byte
  auto dz_element = [&](const float (&d)[2][KOUT], int rt, int e) {
    const int set = KOUT == 2 ? (e >> 1) & 1 : e & 1;
    {
      u32x4 &q = cq[set];
      asm volatile("" : "+v"(q));
    }
    const int at = KOUT == 2 ? 2 * (e & 1) : 0;
    float g = d[rt][0] * __uint_as_float(cq[set][at]);
#pragma unroll
    for (int o = 1; o < KOUT; ++o) g = __builtin_fmaf(d[rt][o], __uint_as_float(cq[set][at + o]), g);
    const uint32_t open = (uint32_t)__builtin_amdgcn_sbfe((int)gbyte[rt], (unsigned)e, 1u);  // 0 or ~0
    return __uint_as_float(open & __float_as_uint(g));
  };
  constexpr int kRowLoads = 2 * (KOUT + kXRegs);

  auto do_half = [&](auto first_tag, auto cur_tag, auto c_tag, auto kind_tag, int hs, int stage, int64_t tile) {
    constexpr bool FIRST = decltype(first_tag)::value;
    constexpr int CUR = decltype(cur_tag)::value;
    constexpr int C = decltype(c_tag)::value;
    constexpr int KIND = decltype(kind_tag)::value;
    // KIND 0: produces row tile C of block S + 1 and its successor produces too; 1 = (6, 1): produces, the successor does
    // not; 2 = (7, 0): nothing produced, the NEXT tile's gate block is requested; 3 = (7, 1): nothing requested for a successor
    constexpr bool kProduce = KIND <= 1, kNextProduces = KIND == 0, kEnd = KIND == 3;
    const int stage_next = stage + 1 == RING ? 0 : stage + 1, stage_free = stage == 0 ? RING - 1 : stage - 1;
    const unsigned br = b_lane + stage * kRowsChunk, br_next = b_lane + stage_next * kRowsChunk;
    const int S = hs >> 1;
    [[maybe_unused]] float h[2];
#pragma unroll
    for (int sl = 0; sl < 8; ++sl) {
      const int set = sl & 1, ahead = set ^ 1, ct = 8 * C + sl;
      // requests of this slot, oldest first: [record of the element one slot on] [gate byte of the next half-step's
      // block (slot 7)] [fragments of the column tile one slot on]
      int newer = 0;
      {
        const bool on = sl == 7 ? kNextProduces : kProduce;
        const int S1 = sl == 7 ? S + 1 + C : S + 1, e1 = (sl + 1) & 7;
        if (on && (KOUT != 2 || (sl & 1))) {
          request_record(S1, e1);
          ++newer;
        }
        if (sl == 7 && kNextProduces) {
          request_byte(S + 1 + C, C ^ 1);
          ++newer;
        }
      }
      if (KIND == 2 && sl == 0) request_gate(tile + stride);
      if (sl < 7) request_block(br, sl + 1, ahead);
      else if (!kEnd) request_block(br_next, 0, ahead);
      newer += (kEnd && sl == 7) ? 0 : 2;
      newer == 0   ? wait_lds<0>(bh[set], bl[set])
      : newer == 2 ? wait_lds<2>(bh[set], bl[set])
      : newer == 3 ? wait_lds<3>(bh[set], bl[set])
                   : wait_lds<4>(bh[set], bl[set]);
      const f32x4 zero = {0, 0, 0, 0};
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, a_lo[CUR][rt]), __builtin_bit_cast(half8, bh[set]),
                                                             FIRST ? zero : acc[rt][ct], 0, 0, 0);
        acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, a_hi[CUR][rt]), __builtin_bit_cast(half8, bl[set]),
                                                             acc[rt][ct], 0, 0, 0);
        acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, a_hi[CUR][rt]), __builtin_bit_cast(half8, bh[set]),
                                                             acc[rt][ct], 0, 0, 0);
      }
      if constexpr (kProduce) {  // element sl of row tile C of block S + 1
        if (sl == 0) asm volatile("" : "+v"(gbyte[C]));  // (requested a half-step ago: landed behind this slot's wait)
        h[sl & 1] = dz_element(dc, C, sl);
        if (sl & 1) {
          uint32_t hi, lo;
          f16_pair_scaled(h[0], h[1], sc[C], hi, lo);
          a_hi[CUR ^ 1][C][sl >> 1] = hi;
          a_lo[CUR ^ 1][C][sl >> 1] = lo;
        }
      }
      if (sl == 3) {
        // (the counts of the gate-mode kernel above, with this kernel's row loads)
        constexpr int kExtra = KIND == 2                 ? 1
                               : KIND == 3               ? (kAhead >= 3 ? 1 : 0)
                               : (FIRST && (C == 0 || kAhead >= 3)) ? kRowLoads
                                                         : 0;
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(4 * (kAhead - 2) + kExtra) : "memory");
      }
      if (sl >= 4) chunk_piece((hs + kAhead) & (kRowsHalfSteps - 1), stage_free, wave * 4 + (sl - 4));
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  using T = std::true_type;
  using F = std::false_type;
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using K0 = I0;
  using K1 = I1;
  using K2 = std::integral_constant<int, 2>;
  using K3 = std::integral_constant<int, 3>;

  // ---- prologue: the first tile's gate block, then the first kAhead chunks; everything before them landed --------------
  __syncthreads();  // (records and zeroed sums in LDS; the reduction's words in the ring are read)
  if ((int64_t)blockIdx.x < tiles) {
    request_gate(blockIdx.x);
#pragma unroll
    for (int d = 0; d < kAhead; ++d)
#pragma unroll
      for (int u = 0; u < 4; ++u) chunk_piece(d, d, wave * 4 + u);
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(4 * (kAhead - 1)) : "memory");
    load_rows(dc, xc, blockIdx.x);
  }

  int stage = 0;
  auto next_stage = [&]() { stage = stage + 1 == RING ? 0 : stage + 1; };
  // (INVARIANT of rows8_epilogue's chain: uniform trip count, every wave runs the epilogue once per tile -- see the
  // gate-mode kernel's tile loop)
  for (int64_t tile = blockIdx.x; tile < tiles; tile += stride) {
    // ---- open the tile (see the gate-mode kernel): the gate block has landed; the next tile's rows requested; row
    // scales; factors and observations to the exchange; block 0's fragments of both row tiles; the first requests
    // (the next tile's rows are requested from the middle of this tile's epilogue, into dc / xc themselves: dOut is last
    // read in half-step 13, the observations here -- and in the order of the wave's vector-memory operations they still
    // lie between half-step 15's pieces and the next tile's half-step 0's, as the barriers' counts assume)
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    {
      float fac[2];
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        float bound = 0.0f;
#pragma unroll
        for (int q = 0; q < KOUT; ++q) bound = __builtin_fmaf(__builtin_fabsf(dc[rt][q]), w3max[q], bound);
        const int e = f16_bound_exponent(bound);
        sc[rt] = __builtin_amdgcn_ldexpf(1.0f, kF16Top - e);
        fac[rt] = __builtin_amdgcn_ldexpf(inv_w2_scale, e - kF16Top);
      }
      if constexpr (kMma1) {
        scales8 = rows8_open(fac, xc, inv_w1, 36, wave_lds, l16, kq, xf);  // |acc| < 256 x 2^14 x 2^14
      } else if (kq == 0) {
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
          lds_write_b32(wave_lds + kFacOff + (16 * rt + l16) * 4, fac[rt]);
#pragma unroll
          for (int i = 0; i < kIn; ++i) lds_write_b32(wave_lds + kObsOff + (32 * i + 16 * rt + l16) * 4, xc[rt][i]);
        }
      }
      request_byte(0, 0);
      request_byte(0, 1);
      float hv[2][8];
      constexpr int kBatch = KOUT == 2 ? 4 : 2;  // elements whose records fit the two register sets at once
#pragma unroll
      for (int e0 = 0; e0 < 8; e0 += kBatch) {
#pragma unroll
        for (int e = e0; e < e0 + kBatch; ++e)
          if (KOUT != 2 || (e & 1) == 0) request_record(0, e);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(gbyte[0]), "+v"(gbyte[1]));
#pragma unroll
        for (int e = e0; e < e0 + kBatch; ++e)
#pragma unroll
          for (int rt = 0; rt < 2; ++rt) hv[rt][e] = dz_element(dc, rt, e);
      }
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
          uint32_t hi, lo;
          f16_pair_scaled(hv[rt][e], hv[rt][e + 1], sc[rt], hi, lo);
          a_hi[0][rt][e >> 1] = hi;
          a_lo[0][rt][e >> 1] = lo;
        }
      // what half-step (0, 0) expects to be on its way, oldest first: [record 0 of block 1] [gate byte of (block 1, row
      // tile 0)] [column tile 0]
      request_record(1, 0);
      request_byte(1, 0);
      request_block(b_lane + stage * kRowsChunk, 0, 0);
    }
    do_half(T{}, I0{}, I0{}, K0{}, 0, stage, tile);  next_stage();
    do_half(T{}, I0{}, I1{}, K0{}, 1, stage, tile);  next_stage();
    do_half(F{}, I1{}, I0{}, K0{}, 2, stage, tile);  next_stage();
    do_half(F{}, I1{}, I1{}, K0{}, 3, stage, tile);  next_stage();
#pragma unroll 1
    for (int hs = 4; hs < kRowsHalfSteps - 4; hs += 4) {
      do_half(F{}, I0{}, I0{}, K0{}, hs, stage, tile);      next_stage();
      do_half(F{}, I0{}, I1{}, K0{}, hs + 1, stage, tile);  next_stage();
      do_half(F{}, I1{}, I0{}, K0{}, hs + 2, stage, tile);  next_stage();
      do_half(F{}, I1{}, I1{}, K0{}, hs + 3, stage, tile);  next_stage();
    }
    do_half(F{}, I0{}, I0{}, K0{}, kRowsHalfSteps - 4, stage, tile);  next_stage();
    do_half(F{}, I0{}, I1{}, K1{}, kRowsHalfSteps - 3, stage, tile);  next_stage();
    do_half(F{}, I1{}, I0{}, K2{}, kRowsHalfSteps - 2, stage, tile);  next_stage();
    do_half(F{}, I1{}, I1{}, K3{}, kRowsHalfSteps - 1, stage, tile);  next_stage();

    // ---- epilogue: dZ1 = dH1 * factor * (h1 > 0), folded into the wave's running column sums (the gate-mode kernel's) -------
    if constexpr (kMma1) {
      load_rows(dc, xc, tile + stride);  // (both are dead since half-step 13 / the opening)
      rows8_epilogue(acc, xf, lds0 + kWaveOff + wave * kWaveBytes, lds0 + kRecOff, wave, done8, scales8, lane);
      done8 += 16;
    } else {
      int lane_e = lane;
      asm volatile("" : "+v"(lane_e));
      const int l16e = lane_e & 15, kqe = lane_e >> 4;
      const unsigned wl = lds0 + kWaveOff + wave * kWaveBytes;
      // (narrow observations: dZ1 = a x factor is never formed -- the sums take a with the factor, resp. factor x
      // observation, as the fma's multiplier: one instruction per element less; d_in = 3 has no registers for 24 products)
      constexpr bool kFoldFactor = kIn <= 2;
      [[maybe_unused]] float fx[2][kFoldFactor ? kIn : 1][4];
      u32x4 fq[2], xq[2][kIn];
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        fq[rt] = rt == 0 ? lds_read_b128<kFacOff>(wl + 16 * kqe) : lds_read_b128<kFacOff + 64>(wl + 16 * kqe);
#pragma unroll
        for (int i = 0; i < kIn; ++i)
          xq[rt][i] = rt == 0 ? (i == 0   ? lds_read_b128<kObsOff>(wl + 16 * kqe)
                                 : i == 1 ? lds_read_b128<kObsOff + 128>(wl + 16 * kqe)
                                          : lds_read_b128<kObsOff + 256>(wl + 16 * kqe))
                              : (i == 0   ? lds_read_b128<kObsOff + 64>(wl + 16 * kqe)
                                 : i == 1 ? lds_read_b128<kObsOff + 128 + 64>(wl + 16 * kqe)
                                          : lds_read_b128<kObsOff + 256 + 64>(wl + 16 * kqe));
      }
      const unsigned rec_at = lds0 + kRecOff + l16e * (kRec * 4), sum_at = wl + kSumOff + l16e * (kRec * 4);
      typedef typename std::conditional<kRec == 2, u32x2, u32x4>::type rec_t;
      rec_t rq[2], sq[2];
      auto request_col = [&](int ct, int set) {
        const unsigned ra = rec_at + ct * (16 * kRec * 4), sa = sum_at + ct * (16 * kRec * 4);
        if constexpr (kRec == 2) {
          rq[set] = lds_read_b64<0>(ra);
          sq[set] = lds_read_b64<0>(sa);
        } else {
          rq[set] = lds_read_b128<0>(ra);
          sq[set] = lds_read_b128<0>(sa);
        }
      };
      request_col(0, 0);
#pragma unroll
      for (int ct = 0; ct < 16; ++ct) {
        const int set = ct & 1;
        if (ct == 8) load_rows(dc, xc, tile + stride);
        if (ct + 1 < 16) request_col(ct + 1, set ^ 1);
        if (ct + 1 < 16) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(rq[set]), "+v"(sq[set]));
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(rq[set]), "+v"(sq[set]));
        if (ct == 0) {
#pragma unroll
          for (int rt = 0; rt < 2; ++rt) {
            asm volatile("" : "+v"(fq[rt]));
#pragma unroll
            for (int i = 0; i < kIn; ++i) asm volatile("" : "+v"(xq[rt][i]));
          }
          if constexpr (kFoldFactor) {  // factor x observation per row, once per tile: one fma per sum and element below
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
              for (int i = 0; i < kIn; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) fx[rt][i][r] = __uint_as_float(fq[rt][r]) * __uint_as_float(xq[rt][i][r]);
          }
        }
        const float b1c = __uint_as_float(rq[set][0]);
        float w1c[kIn];
#pragma unroll
        for (int i = 0; i < kIn; ++i) w1c[i] = __uint_as_float(rq[set][1 + i]);
        float db = 0.0f, dw[kIn];
#pragma unroll
        for (int i = 0; i < kIn; ++i) dw[i] = 0.0f;
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
          float pre[4];
          unsigned long long open[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            pre[r] = b1c;
#pragma unroll
            for (int i = 0; i < kIn; ++i) pre[r] = __builtin_fmaf(__uint_as_float(xq[rt][i][r]), w1c[i], pre[r]);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) open[r] = positive_mask(pre[r]);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if constexpr (kFoldFactor) {
              const float a = select_or_zero(open[r], acc[rt][ct][r]);
              db = __builtin_fmaf(a, __uint_as_float(fq[rt][r]), db);
#pragma unroll
              for (int i = 0; i < kIn; ++i) dw[i] = __builtin_fmaf(a, fx[rt][i][r], dw[i]);
            } else {
              const float dz = select_or_zero(open[r], acc[rt][ct][r]) * __uint_as_float(fq[rt][r]);
              db += dz;
#pragma unroll
              for (int i = 0; i < kIn; ++i) dw[i] = __builtin_fmaf(dz, __uint_as_float(xq[rt][i][r]), dw[i]);
            }
          }
        }
        auto across = [&](float v) {
          const auto s16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
          const float u = __uint_as_float(s16[0]) + __uint_as_float(s16[1]);
          const auto s32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(u), __float_as_uint(u), false, false);
          return __uint_as_float(s32[0]) + __uint_as_float(s32[1]);
        };
        db = across(db);
#pragma unroll
        for (int i = 0; i < kIn; ++i) dw[i] = across(dw[i]);
        if (kqe == (ct & 3)) {
          const unsigned sa = sum_at + ct * (16 * kRec * 4);
          lds_write_b32(sa, __uint_as_float(sq[set][0]) + db);
#pragma unroll
          for (int i = 0; i < kIn; ++i) lds_write_b32(sa + 4 + 4 * i, __uint_as_float(sq[set][1 + i]) + dw[i]);
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();

  // Workgroup partial row: [dW1 (256 * d_in) | db1 (256) | head gradients (the weight-gradient kernel's)], the four
  // waves' sums added in wave order.
  float *row = partials + (int64_t)blockIdx.x * partial_stride;
  {
    const int t = tid;
    if constexpr (kMma1) {  // the workgroup's one array: [unit][dW1 row 0..6 | db1]
      const float *sums = reinterpret_cast<const float *>(smem + kRecOff + kRows8Planes + 1024);
      row[kHidden * d_in + t] = sums[t * 8 + 7];
#pragma unroll
      for (int i = 0; i < 7; ++i)
        if (i < d_in) row[t * d_in + i] = sums[t * 8 + i];
    } else {
      float tot[1 + kIn];
#pragma unroll
      for (int i = 0; i < 1 + kIn; ++i) tot[i] = 0.0f;
      for (int w = 0; w < 4; ++w) {
        const float *sums = reinterpret_cast<const float *>(smem + kWaveOff + w * kWaveBytes + kSumOff);
#pragma unroll
        for (int i = 0; i < 1 + kIn; ++i) tot[i] += sums[t * kRec + i];
      }
      row[kHidden * d_in + t] = tot[0];
#pragma unroll
      for (int i = 0; i < kIn; ++i) row[t * d_in + i] = tot[1 + i];
    }
    if ((int)blockIdx.x >= head_rows)
      for (int idx = kHidden * d_in + kHidden + t; idx < partial_stride; idx += kBlock) row[idx] = 0.0f;
  }
}

template <int DIN, int KOUT>
static int launch_rows_backward_general(int grid, hipStream_t s, const float *x, const float *w1, const float *b1, const float *dout,
                                        int64_t m, const void *w2ts, const float *w3, float *partials, int stride, int head_rows,
                                        const uint32_t *gate2, int n_out, int d_in) {
  constexpr int kRing = 3;
  auto kernel = &mlp_rows_backward_general_kernel<DIN, KOUT, kRing>;
  static LdsOptIn opt;
  if (const int e = allow_dynamic_lds(opt, reinterpret_cast<const void *>(kernel), 160 * 1024)) return e;
  constexpr int kLds = DIN > 3 ? rows8_dgrad_lds_bytes(kRing, KOUT) : rows_dgrad_general_lds_bytes(kRing, DIN, KOUT);
  kernel<<<grid, kBlock, kLds, s>>>(x, w1, b1, dout, m, w2ts, w3, partials, stride, head_rows, gate2, n_out, d_in);
  return launch_status();
}

// The general data gradient behind rl8_mlp_tower_backward_f16_f32 for d_in <= 3, n_out 2..4 (-1: no variant: the caller
// keeps the tile kernel).  Same grid and partial rows as that kernel.
int mlp_rows_backward_general_dispatch(int grid, hipStream_t s, const float *x, const float *w1, const float *b1, const float *dout,
                                       int64_t m, int d_in, const void *w2ts, const float *w3, int n_out, float *partials, int stride,
                                       int head_rows, const uint32_t *gate2) {
  const int dc = d_in <= 3 ? d_in : d_in <= 7 ? 8 : 0;  // (class 8 keeps M slot 7 for db1: seven inputs)
#define RL8_ROWS_BWD_GENERAL(D) \
  if (dc == D && n_out == 2) return launch_rows_backward_general<D, 2>(grid, s, x, w1, b1, dout, m, w2ts, w3, partials, stride, head_rows, gate2, n_out, d_in); \
  if (dc == D && (n_out == 3 || n_out == 4)) return launch_rows_backward_general<D, 4>(grid, s, x, w1, b1, dout, m, w2ts, w3, partials, stride, head_rows, gate2, n_out, d_in);
  RL8_ROWS_BWD_GENERAL(1) RL8_ROWS_BWD_GENERAL(2) RL8_ROWS_BWD_GENERAL(3) RL8_ROWS_BWD_GENERAL(8)
#undef RL8_ROWS_BWD_GENERAL
  // one output in GENERAL mode (h2 given, no gate pack: the product takes the gate mode for it) at d_in 6, 7 -- widths the
  // tile kernel of mlp_f16_kernels.hip is not compiled for: the two-output class with a zero second record
  if (dc == 8 && d_in >= 6 && n_out == 1)
    return launch_rows_backward_general<8, 2>(grid, s, x, w1, b1, dout, m, w2ts, w3, partials, stride, head_rows, gate2, n_out, d_in);
  return -1;
}

template <int DIN, int NOUT, int SAVE>
static int launch_rows_forward(hipStream_t s, const float *x, int64_t m, const float *w1, const float *b1, const void *w2s,
                               const float *b2, const float *w3, const float *b3, float *out, float *h1, float *h2,
                               uint32_t *gate, int d_in, int n_out) {
  constexpr int kOut = NOUT;
  // four chunks of W2 in flight where they fit beside the class's constants (and the h2 transpose scratch), else three
  constexpr int kRing = rows_lds_bytes(4, DIN, kOut, SAVE == 1) <= 80 * 1024 ? 4 : 3;
  constexpr int kLds = rows_lds_bytes(kRing, DIN, kOut, SAVE == 1);
  auto kernel = &mlp_rows_forward_kernel<DIN, NOUT, SAVE, kRing>;
#ifdef RL8_ROWS_STAMP
  if constexpr (DIN == 1 && NOUT == 2 && SAVE == 0) {  // tuning builds: RL8_ROWS_DIAG selects a cut-down variant
    const int diag = env_int("RL8_ROWS_DIAG");
    kernel = diag == 1   ? &mlp_rows_forward_kernel<DIN, NOUT, SAVE, kRing, 1>
             : diag == 2 ? &mlp_rows_forward_kernel<DIN, NOUT, SAVE, kRing, 2>
             : diag == 3 ? &mlp_rows_forward_kernel<DIN, NOUT, SAVE, kRing, 3>
             : diag == 4 ? &mlp_rows_forward_kernel<DIN, NOUT, SAVE, kRing, 4>
             : diag == 7 ? &mlp_rows_forward_kernel<DIN, NOUT, SAVE, kRing, 7>
                         : kernel;
  }
  // (tuning builds pick the kernel at run time: the attribute is set on every call)
  if (const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      e != hipSuccess)
    return (int)e;
#else
  static LdsOptIn lds_forward;
  if (const int e = allow_dynamic_lds(lds_forward, reinterpret_cast<const void *>(kernel), 160 * 1024)) return e;
#endif
  const int64_t tiles = (m + 127) / 128;
  static const int cap = env_int("RL8_MLP_GRID_CAP");
  const int max_grid = cap > 0 ? cap : 2 * kCUs;
  const int grid = (int)(tiles < max_grid ? tiles : max_grid);
#ifdef RL8_ROWS_STAMP
  const char *sp = getenv("RL8_ROWS_STAMP_PTR");
  kernel<<<grid, kBlock, kLds, s>>>(x, m, w1, b1, w2s, b2, w3, b3, out, h1, h2, gate, d_in, n_out,
                                    sp ? reinterpret_cast<unsigned long long *>(strtoull(sp, nullptr, 0)) : nullptr);
#else
  kernel<<<grid, kBlock, kLds, s>>>(x, m, w1, b1, w2s, b2, w3, b3, out, h1, h2, gate, d_in, n_out);
#endif
  return launch_status();
}

template <int DIN, int NOUT>
static int launch_rows_forward_save(hipStream_t s, const float *x, int64_t m, const float *w1, const float *b1, const void *w2s,
                                    const float *b2, const float *w3, const float *b3, float *out, float *h1, float *h2,
                                    uint32_t *gate, int d_in, int n_out) {
  return h2     ? launch_rows_forward<DIN, NOUT, 1>(s, x, m, w1, b1, w2s, b2, w3, b3, out, h1, h2, gate, d_in, n_out)
         : gate ? launch_rows_forward<DIN, NOUT, 2>(s, x, m, w1, b1, w2s, b2, w3, b3, out, h1, h2, gate, d_in, n_out)
                : launch_rows_forward<DIN, NOUT, 0>(s, x, m, w1, b1, w2s, b2, w3, b3, out, h1, h2, gate, d_in, n_out);
}

// Width classes (round 5): the smallest compiled class that holds the run-time width.
int rows_in_class(int d_in) { return d_in <= 3 ? d_in : d_in <= 8 ? 8 : d_in <= 16 ? 16 : 0; }
int rows_out_class(int n_out) { return n_out >= 1 && n_out <= 8 ? pad_out(n_out) : 0; }

// The forward behind rl8_mlp_tower_forward_f16_f32 (mlp_f16_kernels.hip checks the arguments and dispatches here).
int mlp_rows_forward_dispatch(hipStream_t s, const float *x, int64_t m, int d_in, const float *w1, const float *b1,
                              const void *w2s, const float *b2, const float *w3, const float *b3, int n_out, float *out,
                              float *h1, float *h2, uint32_t *gate) {
  const int dc = rows_in_class(d_in), nc = rows_out_class(n_out);
#define RL8_ROWS(D, N) \
  if (dc == D && nc == N) return launch_rows_forward_save<D, N>(s, x, m, w1, b1, w2s, b2, w3, b3, out, h1, h2, gate, d_in, n_out);
  RL8_ROWS(1, 1) RL8_ROWS(1, 2) RL8_ROWS(1, 4) RL8_ROWS(1, 8)
  RL8_ROWS(2, 1) RL8_ROWS(2, 2) RL8_ROWS(2, 4) RL8_ROWS(2, 8)
  RL8_ROWS(3, 1) RL8_ROWS(3, 2) RL8_ROWS(3, 4) RL8_ROWS(3, 8)
  RL8_ROWS(8, 1) RL8_ROWS(8, 2) RL8_ROWS(8, 4) RL8_ROWS(8, 8)
  RL8_ROWS(16, 1) RL8_ROWS(16, 2) RL8_ROWS(16, 4)  // (sixteen inputs x eight outputs: no room for the h2 scratch at two workgroups per CU)
#undef RL8_ROWS
  return RL8_ESIZE;
}

}  // namespace rl8
