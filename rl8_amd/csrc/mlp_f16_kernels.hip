// N1, the fp16 two-plane scheme (round 2): the 256x256 layer as an
// fp32-accurate product on the fp16 matrix pipe with TWO planes per operand and THREE products
// per fp32 product -- half the matrix-pipe cycles of the bf16 three-plane scheme of
// mlp_split_kernels.hip.  This file: the pack kernel, the data-gradient kernel (structure of the
// bf16-plane one: both operands through LDS in 16-k chunks, W2 by direct-to-LDS loads, the next
// chunk produced on the VALU beside the current step's MFMAs) and the ABI entry points.  The
// FORWARD kernel of the scheme lives in mlp_rows_kernels.hip (round 3: rows-per-wave, 16x16x32).
//
// Why it is still fp32-accurate.  With x scaled by a power of two into fp16's range,
//   hi = fp16_rn(x), lo = fp16_rn(x - hi)    =>   |x - hi - lo| <= 2^-23 |x|  (11 + 11 bits, two
// roundings to nearest), and a*b = ah*bh + (ah*bl + al*bh) + O(2^-22 |a*b|): three fp16 MFMAs
// (exact products, fp32 accumulate).  The dropped al*bl terms have random signs: over K = 256
// they add ~16 * 2^-22 of a typical term, below what an fp32 fma chain loses to its own
// roundings.  Host-side model on the test distributions (tools/diag/fp16_two_plane_model.py,
// profiles/r02_fp16_three_product_error.txt): max error relative to the largest output
// 1.4e-7 (fp32 chain 3.7e-7, six bf16 products 1.5e-7); the GPU tests hold this kernel to the
// same fp64 bars as the bf16-plane one (tests/test_mlp_split_gpu.py, parametrised over both).
// What fp16 costs is RANGE (5 exponent bits): hence the power-of-two scaling per activation
// row and per weight matrix described at set_row_scale().
#include "split_tile.hip.h"

namespace rl8 {

constexpr int kF16ABytes = 2 * kSplitPlaneStride;          // [plane][k-half][row] x 16 B
constexpr int kF16BBytes = 2 * 8 * 1024;                   // [column tile][plane] x 1 KiB
constexpr int kF16StageBytes = kF16ABytes + kF16BBytes;    // 24 832
constexpr int kF16PackedBytes = kSplitSteps * kF16BBytes;  // 262 144 (+ 16 bytes: scale, 1 / scale)

// w2 [256][256] fp32 -> two fp16 planes of w2 * 2^k in fragment order, k chosen so that
// max |w2| * 2^k < 2^14: 16-byte unit ((s*8 + ct)*2 + p)*64 + l holds, for plane p,
//   B(col = 32 ct + (l & 31), k = 16 s + 8 (l >> 5) + e), e = 0..7  (B as in mlp_pack_w2_split),
// and behind the planes two floats: 2^k and 2^-k.  One workgroup: the maximum first.
// w3 != NULL (with transposed): the planes are those of W2[k][col] * w3e[k], w3e = W3[0] (n_w3 = 1) or
// W3[0] - W3[1] (n_w3 = 2) -- the B operand of the data-gradient kernel's gate mode, where the ReLU gate is
// the A operand and dZ2's dense factors moved into B.
// layout 1 (the 16x16x32 kernels of mlp_rows_kernels.hip): unit ((hs*8 + ctl)*2 + p)*64 + l holds
//   B(col = 16 (8 (hs & 1) + ctl) + (l & 15), k = 32 (hs >> 1) + 8 (l >> 4) + e), e = 0..7.
constexpr int kF16PackGrid = 8;  // 8192 units of 16 bytes per plane pair: one per thread
__global__ __launch_bounds__(1024) void mlp_pack_w2_f16_kernel(const float *__restrict__ w2, int transposed,
                                                               uint32_t *__restrict__ packed,
                                                               const float *__restrict__ w3 = nullptr, int n_w3 = 0,
                                                               int layout = 0) {
  __shared__ float red[1024];
  __shared__ float w3e[kHidden];
  const int tid = threadIdx.x;
  if (tid < kHidden) w3e[tid] = w3 == nullptr ? 1.0f : n_w3 == 2 ? w3[tid] - w3[kHidden + tid] : w3[tid];
  __syncthreads();
  float mx = 0.0f;
  // (row k of w2 is w2[k][.]: the factor belongs to the REDUCTION index of the transposed product)
  // (every workgroup of the launch takes the maximum over the whole matrix -- 256 KiB out of L2 -- and packs its share)
#pragma clang loop vectorize(disable) interleave(disable)  // (no packed fp32 ops anywhere in this file: the ISA test's rule)
  for (int i = 4 * tid; i < kHidden * kHidden; i += 4 * 1024) {
    const f32x4 v = *reinterpret_cast<const f32x4 *>(w2 + i);
    const float f = w3e[i >> 8];
    mx = __builtin_fmaxf(mx, __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(v[0] * f), __builtin_fabsf(v[1] * f)),
                                             __builtin_fmaxf(__builtin_fabsf(v[2] * f), __builtin_fabsf(v[3] * f))));
  }
  red[tid] = mx;
  __syncthreads();
  for (int half = 512; half > 0; half >>= 1) {
    if (tid < half) red[tid] = __builtin_fmaxf(red[tid], red[tid + half]);
    __syncthreads();
  }
  const int e = f16_bound_exponent(red[0]);  // max < 2^e
  const float scale = __builtin_amdgcn_ldexpf(1.0f, kF16Top - e);
  if (tid == 0 && blockIdx.x == 0) {
    float *tail = reinterpret_cast<float *>(reinterpret_cast<unsigned char *>(packed) + kF16PackedBytes);
    tail[0] = scale;
    tail[1] = __builtin_amdgcn_ldexpf(1.0f, e - kF16Top);
    tail[2] = tail[3] = 0.0f;
  }
  for (int unit = blockIdx.x * 1024 + tid; unit < kSplitSteps * 8 * 64; unit += gridDim.x * 1024) {
    const int l = unit & 63, ct = (unit >> 6) & 7, s = unit >> 9;
    const int col = layout == 1 ? 16 * (8 * (s & 1) + ct) + (l & 15) : 32 * ct + (l & 31);
    const int k0 = layout == 1 ? 32 * (s >> 1) + 8 * (l >> 4) : 16 * s + 8 * (l >> 5);
    u32x4 hi, lo;
#pragma unroll
    for (int e2 = 0; e2 < 8; e2 += 2) {
      const float v0 = (transposed ? w2[(k0 + e2) * kHidden + col] * w3e[k0 + e2] : w2[col * kHidden + k0 + e2]) * scale;
      const float v1 = (transposed ? w2[(k0 + e2 + 1) * kHidden + col] * w3e[k0 + e2 + 1] : w2[col * kHidden + k0 + e2 + 1]) * scale;
      uint32_t h, lw;
      f16_pair(v0, v1, h, lw);
      hi[e2 >> 1] = h;
      lo[e2 >> 1] = lw;
    }
    reinterpret_cast<u32x4 *>(packed)[((s * 8 + ct) * 2 + 0) * 64 + l] = hi;
    reinterpret_cast<u32x4 *>(packed)[((s * 8 + ct) * 2 + 1) * 64 + l] = lo;
  }
}

// ---- backward ("dgrad" half) on the same scheme ----------------------------------
//   dZ2 = (dOut x W3) * gate2      per k-chunk on the VALU (thread = row, eight columns),
//                                  scaled by the row's power of two and split into the A planes;
//   dH1 = dZ2 x W2                 three fp16 plane products (B = W2 packed transposed);
//   dZ1 = dH1 * (h1 > 0)           accumulator epilogue (scaling undone), folded into dW1 / db1.
// Structure of mlp_tower_backward_split_kernel (mlp_split_kernels.hip), fused mode only:
// the ReLU gate of h2 comes as bits (the forward kernel's save_gate2), dZ2 is not stored and
// the head gradients are left to the weight-gradient kernel, rl8_mlp_wgrad_fused_split_f32 --
// which stays on bf16 planes.  Two fp16 versions of it were built and measured (round 2):
// its reduction runs over SAMPLES, so a power of two per sample can be taken out of the sum
// only if the two operands' factors multiply to the same constant for every sample.
//   * one power of two per operand and launch: 600 us per 2^20 rows against 586 for the
//     six-product kernel, and entries of dW2 made only of small rows lose relative accuracy
//     (6e-5 of the entry's own sum of |terms| with rows 10^6 apart; bf16 planes 7e-7);
//   * per-sample factors 2^a(s), 2^b(s) with a(s) + b(s) constant, each operand placed half
//     the sample's deficit below fp16's top (exponents from a streaming pre-pass, factors
//     built by scalar integer arithmetic): 764 us, worst entry 4e-5 in the host model
//     (tools/diag/f16_wgrad_mixed_rows.py; nearly dead units whose only terms are small).
// Both operands of that kernel are produced on the VALU every step, so halving the MFMAs
// leaves it bound by operand production (the fp16 split is three conversion instructions
// per element and plane pair), while its accuracy depends on single elements, not on dot
// products along a scaled row as here and in the forward kernel.
// The row bound that places the planes in fp16's range is
//   |dZ2[row][k]| <= sum_q |dOut[row][q]| * max_k |W3[q][k]|,
// a function of the row's dOut alone, so the producer thread has it in registers.
constexpr int f16_backward_lds_bytes(int k_in) {
  return 2 * kF16StageBytes + kHidden * (1 + k_in) * 4 + kSplitRows * 32 + 2 * kSplitRows * 4;
}

// (Round 5: this tile-shaped kernel is the REFERENCE implementation of the general data gradient -- what the rows-shape
// kernels of mlp_rows_kernels.hip are checked against in tests/, RL8_MLP_DGRAD_TILE=1 --
// and the path of single-output heads when the gate kernels are switched off; its own gate mode went with the rows
// kernels' class 8.)
template <int DIN, int NOUT>
__global__ __launch_bounds__(kBlock, 2) void mlp_tower_backward_f16_kernel(
    const float *__restrict__ x, const float *__restrict__ w1, const float *__restrict__ b1,
    const float *__restrict__ dout, int64_t m, const void *__restrict__ w2ts, const float *__restrict__ w3,
    float *__restrict__ partials, int partial_stride, int head_rows, const uint32_t *__restrict__ gate2) {
  static_assert(DIN > 0 && NOUT > 0, "compiled widths only");
  constexpr int kIn = DIN, d_in = DIN, n_out = NOUT;
  constexpr int kOut = pad_out(NOUT);
  static_assert(f16_backward_lds_bytes(kIn) <= 80 * 1024, "two workgroups per CU");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // [stage 0: A | B][stage 1][column sums [256][1 + kIn]][gate block [128 rows][8 words]][row factors [2][128]]
  const unsigned lds0 = lds_offset(smem);
  const int tid = threadIdx.x, lane = tid & 63, l32 = lane & 31, hh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int prow = tid & 127, pkh = wave >> 1;  // producer role: row, wave-uniform k-half (see the forward kernel)
  constexpr int kColsumOff = 2 * kF16StageBytes;
  constexpr int kGateOff = kColsumOff + kHidden * (1 + kIn) * 4;
  constexpr int kScaleOff = kGateOff + kSplitRows * 32;
  const unsigned gate_lds = lds0 + kGateOff;
  const unsigned scale_lds = lds0 + kScaleOff;
  uint32_t g0 = 0u;  // word 0 of row prow of the NEXT tile (its chunk 0 is produced before the block lands)
  const unsigned a_read = lds0 + hh * kSplitKhStride + (64 * wr + l32) * 16;
  const unsigned b_read = lds0 + kF16ABytes + (4 * wc * 2) * 1024 + lane * 16;
  const unsigned a_write = lds0 + pkh * kSplitKhStride + prow * 16;
  const __amdgpu_buffer_rsrc_t w2rsrc = buffer_rsrc(w2ts, kF16PackedBytes);
  const float inv_w2_scale = reinterpret_cast<const float *>(static_cast<const unsigned char *>(w2ts) + kF16PackedBytes)[1];

  const int64_t tiles = (m + kSplitRows - 1) / kSplitRows;
  const int64_t stride = gridDim.x;

  // max_k |W3[q][k]| (uniform: scalar registers), before any stage is in use
  float w3max[kOut];
  {
    float *red = reinterpret_cast<float *>(smem);
#pragma unroll
    for (int q = 0; q < kOut; ++q) {
      __syncthreads();
      red[tid] = q < n_out ? __builtin_fabsf(w3[q * kHidden + tid]) : 0.0f;
      __syncthreads();
      float mx = 0.0f;
      for (int i = 0; i < kHidden; ++i) mx = __builtin_fmaxf(mx, red[i]);
      w3max[q] = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(mx)));
    }
    __syncthreads();
  }

  // Producer state (one step ahead of the consumer; moves to the next tile before step 15).
  int64_t p_tile = blockIdx.x;
  float dr[kOut], dn[kOut];  // dOut of row prow of the producer's tile / of the tile after
  auto rows_in_tile = [&](int64_t tile) {
    const int64_t left = m - tile * kSplitRows;
    return left <= 0 ? 0 : left < kSplitRows ? (int)left : kSplitRows;
  };
  auto load_dout = [&](float (&dst)[kOut], int64_t tile) {
    const int rows = rows_in_tile(tile);
    const float *base = dout + tile * kSplitRows * n_out;
#pragma unroll
    for (int q = 0; q < kOut; ++q) dst[q] = (prow < rows && q < n_out) ? base[(unsigned)(prow * n_out + q)] : 0.0f;
  };
  load_dout(dr, p_tile);
  load_dout(dn, p_tile + stride);
  float p_scale = 1.0f;  // of row prow of the producer's tile
  int p_parity = 0, c_parity = 0;
  auto set_row_scale = [&]() {
    float bound = 0.0f;
#pragma unroll
    for (int q = 0; q < kOut; ++q) bound = __builtin_fmaf(__builtin_fabsf(dr[q]), w3max[q], bound);
    const int e = f16_bound_exponent(bound);  // bound < 2^e
    p_scale = __builtin_amdgcn_ldexpf(1.0f, kF16Top - e);
    if (tid < kSplitRows) lds_write_b32(scale_lds + (p_parity * kSplitRows + prow) * 4, __builtin_amdgcn_ldexpf(inv_w2_scale, e - kF16Top));
  };
  [[maybe_unused]] int diag_b_loads = 0;  // (tuning builds, bit 2^20: only the first two chunks are loaded)
  auto request_b = [&](int ks, int stage) {
    if constexpr ((kSplitDiagSkip & (1 << 20)) != 0) {
      if (diag_b_loads >= 2) return;
      ++diag_b_loads;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int block = wave * 4 + u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(w2rsrc, smem + stage * kF16StageBytes + kF16ABytes + block * 1024,
                                               16, lane * 16, (ks * 16 + block) * 1024, 0, 0);
    }
  };
  // Gate block of `tile` -> LDS: wave w copies rows 32w .. 32w+31 (1 KiB, contiguous);
  // rows past the end of the data arrive as zeros (gate closed).
  auto request_gate = [&](int64_t tile) {
    const int rows = rows_in_tile(tile);
    const __amdgpu_buffer_rsrc_t rsrc = buffer_rsrc(rows > 0 ? gate2 + tile * (kSplitRows * 8) : gate2, rows * 32);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, smem + kGateOff + wave * 1024, 16, lane * 16, wave * 1024, 0, 0);
  };
  auto load_g0 = [&](int64_t tile) {
    const int rows = rows_in_tile(tile);
    g0 = prow < rows ? (gate2 + tile * (kSplitRows * 8))[(unsigned)(prow * 8)] : 0u;
  };
  // from_regs: chunk 0 of a tile whose gate block has not landed yet takes its bits from g0.
  auto produce_a = [&](int ks, u32x4 (&planes)[2], bool from_regs = false) {
    const int kb = __builtin_amdgcn_readfirstlane(16 * ks + 8 * pkh);  // W3[q][kb + e]: uniform, through the scalar cache
    uint32_t gword = g0;
    if (!from_regs) gword = __float_as_uint(lds_read_b32(gate_lds + (prow * 8 + (ks >> 1)) * 4));
    const uint32_t byte = gword >> (16 * (ks & 1) + 8 * pkh);
    float dz[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float g = dr[0] * w3[kb + e];
#pragma unroll
      for (int q = 1; q < kOut; ++q)
        if (q < n_out) g = __builtin_fmaf(dr[q], w3[q * kHidden + kb + e], g);
      dz[e] = ((byte >> e) & 1u) != 0 ? g * p_scale : 0.0f;  // (the power of two: exact)
    }
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
      uint32_t hi, lo;
      f16_pair(dz[e], dz[e + 1], hi, lo);
      planes[0][e >> 1] = hi;
      planes[1][e >> 1] = lo;
    }
  };
  auto write_a = [&](int stage, const u32x4 (&planes)[2]) {
    const unsigned addr = a_write + stage * kF16StageBytes;
    lds_write_b128<0>(addr, planes[0]);
    lds_write_b128<kSplitPlaneStride>(addr, planes[1]);
  };
  auto step_barrier = [&]() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

  f32x16 acc[2][4];
  // Running column sums [256][1 + kIn] (db1 | dW1 row) live in LDS behind the two
  // chunk stages, not in registers: the matrix loop has none to spare.
  for (int idx = tid; idx < kHidden * (1 + kIn); idx += kBlock) lds_write_b32(lds0 + kColsumOff + idx * 4, 0.0f);
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

  auto do_step = [&](auto first_tag, auto parity_tag, int s) {
    constexpr bool FIRST = decltype(first_tag)::value;
    constexpr int P = decltype(parity_tag)::value;
    const int ks = (s + 1) & (kSplitSteps - 1);
    request_b(ks, P ^ 1);
    const unsigned ar = a_read + P * kF16StageBytes, br = b_read + P * kF16StageBytes;
    SplitFrags f;  // ah / bh: hi planes, am / bm: lo planes
    f.ah[0] = lds_read_b128<0>(ar);
    f.ah[1] = lds_read_b128<512>(ar);
    f.bh[0] = lds_read_b128<0>(br);
    f.bh[1] = lds_read_b128<2 * 1024>(br);
    f.bh[2] = lds_read_b128<4 * 1024>(br);
    f.bh[3] = lds_read_b128<6 * 1024>(br);
    f.am[0] = lds_read_b128<kSplitPlaneStride>(ar);
    f.am[1] = lds_read_b128<kSplitPlaneStride + 512>(ar);
    f.bm[0] = lds_read_b128<1024>(br);
    f.bm[1] = lds_read_b128<3 * 1024>(br);
    f.bm[2] = lds_read_b128<5 * 1024>(br);
    f.bm[3] = lds_read_b128<7 * 1024>(br);
    if (s == kSplitSteps - 1) {  // the producer moves on to the next tile
#pragma unroll
      for (int q = 0; q < kOut; ++q) dr[q] = dn[q];
      p_tile += stride;
      load_dout(dn, p_tile + stride);
      p_parity ^= 1;
      set_row_scale();
      // every wave is past barrier(14): nobody reads the old gate block any more;
      // the new one lands by this step's barrier, chunk 0 uses g0 meanwhile
      request_gate(p_tile);
    }
    u32x4 planes[2];
    produce_a(ks, planes, s == kSplitSteps - 1);
    if (s == kSplitSteps - 3) load_g0(p_tile + stride);  // two steps ahead of its use
    wait_lds_all(f);
    f16_mma<FIRST>(f.am, f.bh, acc);  // lo x hi
    f16_mma<false>(f.ah, f.bm, acc);  // hi x lo
    __builtin_amdgcn_sched_barrier(0);
    write_a(P ^ 1, planes);
    __builtin_amdgcn_sched_barrier(0);
    f16_mma<false>(f.ah, f.bh, acc);  // hi x hi
    __builtin_amdgcn_sched_barrier(0);
    step_barrier();
  };
  using T = std::true_type;
  using F = std::false_type;
  using P0 = std::integral_constant<int, 0>;
  using P1 = std::integral_constant<int, 1>;

  if ((int64_t)blockIdx.x < tiles) {
    set_row_scale();
    request_gate(p_tile);
    step_barrier();  // (once per kernel: the first gate block has landed)
    request_b(0, 0);
    u32x4 planes[2];
    produce_a(0, planes);
    write_a(0, planes);
    step_barrier();
  }

  for (int64_t tile = blockIdx.x; tile < tiles; tile += stride) {
    const int64_t r0 = tile * kSplitRows;
    const int rows = (int)((m - r0) < kSplitRows ? (m - r0) : kSplitRows);
    do_step(T{}, P0{}, 0);
    do_step(F{}, P1{}, 1);
#pragma unroll 1
    for (int s = 2; s < kSplitSteps - 2; s += 2) {
      do_step(F{}, P0{}, s);
      do_step(F{}, P1{}, s + 1);
    }
    do_step(F{}, P0{}, kSplitSteps - 2);
    do_step(F{}, P1{}, kSplitSteps - 1);

    // Epilogue: dZ1 = dH1 * (h1 > 0) folded into db1 / dW1, the gate RECOMPUTED from the
    // observations (h1 > 0 <=> b1 + x . w1 > 0, the forward pass's own fma chain) -- see
    // the bf16-plane kernel.  Here the accumulators still carry the operand scaling: each
    // row's factor (its own power of two and W2's, inverted) comes from LDS, four rows a read.
    const __amdgpu_buffer_rsrc_t xrsrc = buffer_rsrc(x + r0 * d_in, rows * d_in * 4);
    const int l32 = lane_id() & 31, hh = lane_id() >> 5;  // (recomputed: see lane_id)
    const unsigned colsum = lds_offset(smem) + kColsumOff;
    const unsigned factors = lds_offset(smem) + kScaleOff + (c_parity * kSplitRows + 64 * wr + 4 * hh) * 4;
    c_parity ^= 1;
    float w1c[4][kIn], b1c[4];  // this lane's four columns of layer 1 (reloaded per tile: L1 hits)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const int col = 128 * wc + 32 * nt + l32;
      b1c[nt] = b1[col];
#pragma unroll
      for (int i = 0; i < kIn; ++i) w1c[nt][i] = w1[col * d_in + i];
    }
    float db1[4], dw1[4][kIn];  // this tile: this lane's four columns, its half of the rows
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      db1[nt] = 0.0f;
#pragma unroll
      for (int i = 0; i < kIn; ++i) dw1[nt][i] = 0.0f;
    }
    constexpr int kXRows = kIn <= 2 ? 16 : 4;
    constexpr int kBatches = 2 * 16 / kXRows;  // batches of kXRows rows over both row tiles
    auto load_rows = [&](float (&dst)[kXRows][kIn], int batch) {
      const int mt = batch / (16 / kXRows), rx = (batch % (16 / kXRows)) * kXRows;
#pragma unroll
      for (int u = 0; u < kXRows; ++u) {
        const int r = rx + u;
        const int sr = 64 * wr + 32 * mt + (r & 3) + 8 * (r >> 2);  // + 4*hh
#pragma unroll
        for (int i = 0; i < kIn; ++i) dst[u][i] = buffer_load_f32(xrsrc, (4 * hh * d_in + i) * 4, sr * d_in * 4);
      }
    };
    float xbuf[2][kXRows][kIn];
    load_rows(xbuf[0], 0);
#pragma unroll
    for (int batch = 0; batch < kBatches; ++batch) {
      const int mt = batch / (16 / kXRows), rx = (batch % (16 / kXRows)) * kXRows;
      if (batch + 1 < kBatches) load_rows(xbuf[(batch + 1) & 1], batch + 1);
      float (&xv)[kXRows][kIn] = xbuf[batch & 1];
#pragma unroll
      for (int rb = 0; rb < kXRows; rb += 4) {
        // rows 64 wr + 32 mt + 8 ((rx + rb) / 4) + 4 hh + u, u = 0..3: their four factors
        u32x4 fq = (mt == 0 ? lds_read_b128<0>(factors + 8 * ((rx + rb) >> 2) * 4)
                            : lds_read_b128<32 * 4>(factors + 8 * ((rx + rb) >> 2) * 4));
        wait_lds<0>(fq);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          float pre[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            pre[u] = b1c[nt];
#pragma unroll
            for (int i = 0; i < kIn; ++i) pre[u] = __builtin_fmaf(xv[rb + u][i], w1c[nt][i], pre[u]);
          }
          unsigned long long gate[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) gate[u] = positive_mask(pre[u]);
          __builtin_amdgcn_sched_barrier(0);
          float dz[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) dz[u] = select_or_zero(gate[u], acc[mt][nt][rx + rb + u]) * __uint_as_float(fq[u]);
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            db1[nt] += dz[u];
#pragma unroll
            for (int i = 0; i < kIn; ++i) dw1[nt][i] = __builtin_fmaf(dz[u], xv[rb + u][i], dw1[nt][i]);
          }
        }
      }
    }
    // Into the running sums, in a fixed order: the two row halves of a lane pair, then the
    // wave of rows 0..63, a barrier, the wave of rows 64..127.
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      db1[nt] += __shfl_xor(db1[nt], 32, kWave);
#pragma unroll
      for (int i = 0; i < kIn; ++i) dw1[nt][i] += __shfl_xor(dw1[nt][i], 32, kWave);
    }
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      if (wr == half && hh == 0) {
        float cur[4][1 + kIn];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
          for (int i = 0; i < 1 + kIn; ++i)
            asm volatile("ds_read_b32 %0, %1" : "=v"(cur[nt][i]) : "v"(colsum + ((128 * wc + 32 * nt + l32) * (1 + kIn) + i) * 4));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          const unsigned a = colsum + (128 * wc + 32 * nt + l32) * (1 + kIn) * 4;
          float v0 = cur[nt][0];
          asm volatile("" : "+v"(v0));  // (use behind the wait)
          lds_write_b32(a, v0 + db1[nt]);
#pragma unroll
          for (int i = 0; i < kIn; ++i) {
            float vi = cur[nt][1 + i];
            asm volatile("" : "+v"(vi));
            lds_write_b32(a + 4 + 4 * i, vi + dw1[nt][i]);
          }
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
  }

  // Workgroup partial row: [dW1 (256*d_in) | db1 (256) | head gradients (the weight-gradient kernel's)].
  float *row = partials + (int64_t)blockIdx.x * partial_stride;
  {
    const int t = 64 * wave + lane_id();
    const unsigned a = lds_offset(smem) + kColsumOff + t * (1 + kIn) * 4;
    row[kHidden * d_in + t] = lds_read_b32(a);
#pragma unroll
    for (int i = 0; i < kIn; ++i) row[t * d_in + i] = lds_read_b32(a + 4 + 4 * i);
    // the head-gradient segments of the first head_rows rows belong to the fused
    // weight-gradient kernel; rows beyond them are zero.
    if ((int)blockIdx.x >= head_rows)
      for (int idx = kHidden * d_in + kHidden + t; idx < partial_stride; idx += kBlock) row[idx] = 0.0f;
  }
}

template <int DIN, int NOUT>
static int launch_backward_f16(int grid, hipStream_t s, const float *x, const float *w1, const float *b1, const float *dout,
                               int64_t m, const void *w2ts, const float *w3, float *partials, int stride, int head_rows,
                               const uint32_t *gate2) {
  static LdsOptIn lds_attr_set_0;
  if (const int e_lds_attr_set_0 = allow_dynamic_lds(lds_attr_set_0, reinterpret_cast<const void *>(&mlp_tower_backward_f16_kernel<DIN, NOUT>), 160 * 1024)) return e_lds_attr_set_0;
  mlp_tower_backward_f16_kernel<DIN, NOUT><<<grid, kBlock, f16_backward_lds_bytes(DIN), s>>>(
      x, w1, b1, dout, m, w2ts, w3, partials, stride, head_rows, gate2);
  return launch_status();
}

}  // namespace rl8

using namespace rl8;

RL8_API int64_t rl8_mlp_f16_packed_bytes(void) { return kF16PackedBytes + 16; }

RL8_API int rl8_mlp_pack_w2_f16(const float *w2, int transposed, void *packed, void *stream) {
  if (!w2 || !packed) return RL8_ENULL;
  if (((uintptr_t)packed & 15) != 0 || !aligned16(w2)) return RL8_EALIGN;  // (w2 is read as 16-byte vectors)
  // (forward operand: the 16x16x32 fragment order of mlp_rows_kernels.hip; transposed, for the data gradient: 32x32x16)
  mlp_pack_w2_f16_kernel<<<kF16PackGrid, 1024, 0, (hipStream_t)stream>>>(w2, transposed, reinterpret_cast<uint32_t *>(packed), nullptr, 0,
                                                              transposed ? 0 : 1);
  return launch_status();
}

namespace rl8 {
int mlp_rows_backward_gate_dispatch(int grid, hipStream_t s, const float *x, const float *w1, const float *b1, const float *dout,
                                    int64_t m, int d_in, const void *w2ts, int n_out, float *partials, int stride, int head_rows,
                                    const uint32_t *gate2);
int mlp_rows_backward_general_dispatch(int grid, hipStream_t s, const float *x, const float *w1, const float *b1, const float *dout,
                                       int64_t m, int d_in, const void *w2ts, const float *w3, int n_out, float *partials, int stride,
                                       int head_rows, const uint32_t *gate2);
int mlp_rows_forward_dispatch(hipStream_t s, const float *x, int64_t m, int d_in, const float *w1, const float *b1,
                              const void *w2s, const float *b2, const float *w3, const float *b3, int n_out, float *out,
                              float *h1, float *h2, uint32_t *gate);
}

namespace rl8 {
int rows_in_class(int d_in);
int rows_out_class(int n_out);
}
RL8_API int rl8_mlp_forward_f16_supports(int d_in, int n_out) {
  // (class 16 x output class 8 has no variant: its constants and the h2 scratch do not fit beside three chunks of W2)
  return d_in >= 1 && rows_in_class(d_in) != 0 && rows_out_class(n_out) != 0 && !(rows_in_class(d_in) == 16 && rows_out_class(n_out) == 8);
}

RL8_API int rl8_mlp_tower_forward_f16_f32(const float *x, int64_t m, int d_in, const float *w1,
                                          const float *b1, const void *w2_f16, const float *b2,
                                          const float *w3, const float *b3, int n_out, float *out,
                                          float *save_h1, float *save_h2, uint32_t *save_gate2, void *stream) {
  if (!x || !w1 || !b1 || !w2_f16 || !b2 || !w3 || !b3 || !out) return RL8_ENULL;
  if (m <= 0 || !rl8_mlp_forward_f16_supports(d_in, n_out)) return RL8_ESIZE;
  if (save_h1 && !save_h2) return RL8_ENULL;  // (the gate bits alone are allowed: SAVE mode 2)
  if (((uintptr_t)w2_f16 & 15) != 0 || !aligned16(save_h1) || !aligned16(save_h2) || !aligned16(save_gate2))
    return RL8_EALIGN;
  return mlp_rows_forward_dispatch((hipStream_t)stream, x, m, d_in, w1, b1, w2_f16, b2, w3, b3, n_out, out, save_h1, save_h2,
                                   save_gate2);
}

/* The backward kernels are compiled per width (their observations and dOut arrive through scalar loads of rows whose
 * stride is a compile-time constant): d_in 1..5, n_out 1..4.  Wider towers train through the fp32-MFMA kernels (their
 * rollouts still run the plane forward: rl8_mlp_forward_f16_supports). */
// d_in <= 7 (round 6; 5 until then): the data-gradient kernels' class 8 takes seven inputs (M slot 7 of its dW1 product
// carries the ones of db1), the weight-gradient kernels are compiled per width.  7 x 4 excepted: the exact bf16-plane
// weight gradient (the guard's fallback) keeps a step's observations and dOut in scalar registers -- 8 rows x (7 + 4)
// floats = 88 of them -- and at that width the compiler parks the destination of a load still in flight
// (tests/test_kernel_resources.py::_check_wgrad_scalar_windows finds it): not compiled.
RL8_API int rl8_mlp_backward_f16_supports(int d_in, int n_out) {
  return d_in >= 1 && d_in <= 7 && n_out >= 1 && n_out <= 4 && !(d_in == 7 && n_out == 4);
}

// Grids of the two halves of the fused backward (as in mlp_split_kernels.hip: both derive
// them from m alone, so that each can zero the partial-row segments the other does not cover).
static void f16_backward_grids(int64_t m, int *g1, int *g2) {
  const int64_t tiles = (m + kSplitRows - 1) / kSplitRows;
  const int64_t chunks = (m + 16 - 1) / 16;  // (16-sample chunks of the weight-gradient kernel)
  static const int cap = env_int("RL8_MLP_GRID_CAP");
  const int max_grid = cap > 0 ? cap : 2 * kCUs;
  *g1 = (int)(tiles < max_grid ? tiles : max_grid);
  *g2 = (int)(chunks < kCUs ? chunks : kCUs);
}

RL8_API int rl8_mlp_tower_backward_f16_f32(const float *x, const float *w1, const float *b1, const float *dout,
                                           int64_t m, int d_in, const void *w2t_f16, const float *w3, int n_out,
                                           float *partials, int *partial_rows_out, const uint32_t *gate2,
                                           void *stream) {
  if (!x || !w1 || !b1 || !dout || !w2t_f16 || !w3 || !partials || !partial_rows_out || !gate2) return RL8_ENULL;
  if (m <= 0 || !rl8_mlp_backward_f16_supports(d_in, n_out)) return RL8_ESIZE;
  if (((uintptr_t)w2t_f16 & 15) != 0 || !aligned16(gate2) || !aligned16(w3)) return RL8_EALIGN;
  int grid, g2;
  f16_backward_grids(m, &grid, &g2);
  *partial_rows_out = grid > g2 ? grid : g2;
  const int stride = (int)rl8_mlp_backward_partial_floats(d_in, n_out);
  hipStream_t s = (hipStream_t)stream;
  if (!env_int("RL8_MLP_DGRAD_TILE")) {  // (diagnostics: 1 = the tile kernel for every width; read per call)
    // d_in <= 3, n_out 2..4: the rows-per-wave kernel (mlp_rows_kernels.hip, round 5)
    const int st = mlp_rows_backward_general_dispatch(grid, s, x, w1, b1, dout, m, d_in, w2t_f16, w3, n_out, partials, stride, g2, gate2);
    if (st != -1) return st;
  }
  int status = RL8_ESIZE;
#define RL8_BACKWARD_F16(D, N) \
  if (d_in == D && n_out == N) status = launch_backward_f16<D, N>(grid, s, x, w1, b1, dout, m, w2t_f16, w3, partials, stride, g2, gate2);
  RL8_BACKWARD_F16(1, 1) RL8_BACKWARD_F16(1, 2) RL8_BACKWARD_F16(1, 3) RL8_BACKWARD_F16(1, 4)
  RL8_BACKWARD_F16(2, 1) RL8_BACKWARD_F16(2, 2) RL8_BACKWARD_F16(2, 3) RL8_BACKWARD_F16(2, 4)
  RL8_BACKWARD_F16(3, 1) RL8_BACKWARD_F16(3, 2) RL8_BACKWARD_F16(3, 3) RL8_BACKWARD_F16(3, 4)
  RL8_BACKWARD_F16(4, 1) RL8_BACKWARD_F16(4, 2) RL8_BACKWARD_F16(4, 3) RL8_BACKWARD_F16(4, 4)
  RL8_BACKWARD_F16(5, 1) RL8_BACKWARD_F16(5, 2) RL8_BACKWARD_F16(5, 3) RL8_BACKWARD_F16(5, 4)
#undef RL8_BACKWARD_F16
  return status;
}

/* The B operand of rl8_mlp_tower_backward_gate_f16_f32: planes of W2[k][i] * w3e[k] (w3e = W3[0] for one output,
 * W3[0] - W3[1] for a pair of exactly opposite gradients), layout and size of rl8_mlp_pack_w2_f16(..., transposed = 1). */
RL8_API int rl8_mlp_pack_w2_f16_gate(const float *w2, const float *w3, int n_out, void *packed, void *stream) {
  if (!w2 || !w3 || !packed) return RL8_ENULL;
  if (n_out != 1 && n_out != 2) return RL8_ESIZE;
  if (((uintptr_t)packed & 15) != 0 || !aligned16(w2)) return RL8_EALIGN;
  mlp_pack_w2_f16_kernel<<<kF16PackGrid, 1024, 0, (hipStream_t)stream>>>(w2, 1, reinterpret_cast<uint32_t *>(packed), w3, n_out);
  return launch_status();
}

/* rl8_mlp_tower_backward_f16_f32 for heads whose dZ2 is gate * d[s] * w3e[k]: n_out = 1, or n_out = 2 with
 * dout[s][1] == -dout[s][0] in every row (rl8_mlp_dout_pair_check; column 0 is used).  The gate is the A
 * operand (one plane, no arithmetic), two plane products per 16 k; w2t_gate from rl8_mlp_pack_w2_f16_gate
 * with the same W3.  Same partial rows as the general kernel. */
RL8_API int rl8_mlp_tower_backward_gate_f16_f32(const float *x, const float *w1, const float *b1, const float *dout,
                                                int64_t m, int d_in, const void *w2t_gate, int n_out,
                                                float *partials, int *partial_rows_out, const uint32_t *gate2,
                                                void *stream) {
  if (!x || !w1 || !b1 || !dout || !w2t_gate || !partials || !partial_rows_out || !gate2) return RL8_ENULL;
  if (m <= 0 || (n_out != 1 && n_out != 2) || !rl8_mlp_backward_f16_supports(d_in, n_out)) return RL8_ESIZE;
  if (((uintptr_t)w2t_gate & 15) != 0 || !aligned16(gate2)) return RL8_EALIGN;
  int grid, g2;
  f16_backward_grids(m, &grid, &g2);
  *partial_rows_out = grid > g2 ? grid : g2;
  const int stride = (int)rl8_mlp_backward_partial_floats(d_in, n_out);
  hipStream_t s = (hipStream_t)stream;
  // the rows-per-wave kernels (mlp_rows_kernels.hip): d_in <= 3 with the layer-1 fma chain, 4 and 5 in class 8
  const int st = mlp_rows_backward_gate_dispatch(grid, s, x, w1, b1, dout, m, d_in, w2t_gate, n_out, partials, stride, g2, gate2);
  return st == -1 ? RL8_ESIZE : st;
}
