// N1, second generation of the tower forward: the 256x256 layer as an
// fp32-ACCURATE product on the bf16 matrix pipe.
//
// Why.  gfx950 runs fp32 MFMAs and the VALU on the same multipliers: 155 TFLOP/s,
// and nothing overlaps with them (tools/probes/mfma_valu_probe.hip).  The bf16
// matrix pipe sustains 1.75 PFLOP/s on changing operands and VALU work DOES run
// beside it (tools/probes/bf16_split_probe.hip).  An fp32 value is exactly the sum
// of three bf16 values (8 + 8 + 8 significand bits, by truncation:
// hi = top16(x), mid = top16(x - hi), lo = x - hi - mid), bf16 x bf16 products
// are exact in fp32 and the MFMA accumulates in fp32, so
//   a*b = ah*bh + (ah*bm + am*bh) + (ah*bl + am*bm + al*bh) + O(2^-24 |a*b|)
// -- six bf16 MFMAs per 16 k instead of eight fp32 MFMAs of 1/16 the rate.  The
// dropped terms (am*bl, al*bm, al*bl) are below one fp32 ulp of the product; the
// measured error of a K=256 dot product against fp64 is the same as that of an
// fp32 fma chain.  This is NOT a bf16 tower: inputs,
// outputs, saved activations and accumulation are fp32.
//
// Shape.  Both operands go through LDS, K-chunked, because at this rate the
// weights can no longer stream from L2 per wave: a workgroup (4 waves) owns a
// 128-row x 256-column macro tile, 128 accumulator registers per wave (wave =
// 64 rows x 128 columns), and walks K in 16 steps of 16:
//   B chunk (W2, three planes, fragment-ordered by rl8_mlp_pack_w2_split) comes
//     HBM/L2 -> LDS by direct-to-LDS buffer loads: no registers, no VALU;
//   A chunk (h1 = relu(x W1^T + b1), 128 rows x 16 k) is COMPUTED on the VALU for
//     the next step while the matrix pipe works on the current one, split into
//     the three planes and written to LDS as ready-made 16-byte fragments.
// Two chunk buffers; one barrier per step.  Every LDS access of the loop is inline
// asm with hand-placed s_waitcnt: the compiler serialises compiler-visible LDS
// accesses behind outstanding direct-to-LDS loads (vmcnt(0) before each).
#include <type_traits>
#include "split_tile.hip.h"

namespace rl8 {

// w2 [256][256] row-major fp32 -> three bf16 planes in fragment order:
// 16-byte unit ((s*8 + ct)*3 + p)*64 + l holds, for plane p, the eight values
//   B(col = 32 ct + (l & 31), k = 16 s + 8 (l >> 5) + e), e = 0..7,
// B(col, k) = w2[col][k] (forward) or w2[k][col] (transposed: the data-gradient
// product dH1 = dZ2 x W2).  One k-step of all eight column tiles is 24 KiB
// contiguous: the direct-to-LDS copy of a step is 24 one-KiB loads.
__global__ __launch_bounds__(kBlock) void mlp_pack_w2_split_kernel(const float *__restrict__ w2, int transposed,
                                                                   uint32_t *__restrict__ packed) {
  const int unit = blockIdx.x * kBlock + threadIdx.x;  // (s, ct, l): one 16-byte unit of each plane
  if (unit >= kSplitSteps * 8 * 64) return;
  const int l = unit & 63, ct = (unit >> 6) & 7, s = unit >> 9;
  const int col = 32 * ct + (l & 31), k0 = 16 * s + 8 * (l >> 5);
  uint32_t plane[3][4];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float v = transposed ? w2[(k0 + e) * kHidden + col] : w2[col * kHidden + k0 + e];
    const uint32_t hi = __float_as_uint(v) & 0xffff0000u;
    const float r1 = v - __uint_as_float(hi);
    const uint32_t mid = __float_as_uint(r1) & 0xffff0000u;
    const float r2 = r1 - __uint_as_float(mid);
    const uint32_t lo = __float_as_uint(r2) & 0xffff0000u;
    const uint32_t parts[3] = {hi, mid, lo};
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      if (e & 1)
        plane[p][e >> 1] |= parts[p];
      else
        plane[p][e >> 1] = parts[p] >> 16;
    }
  }
#pragma unroll
  for (int p = 0; p < 3; ++p) {
    uint32_t *dst = packed + ((((s * 8 + ct) * 3 + p) * 64 + l) << 2);
#pragma unroll
    for (int q = 0; q < 4; ++q) dst[q] = plane[p][q];
  }
}

// Kernel-tuning builds only (-DRL8_SPLIT_TRACE): shader-clock stamps of the forward
// kernel's k-steps (tile iteration 3 of every workgroup, every wave), read back with
// rl8_debug_split_trace().  Compiled out of the shipped library.
#ifdef RL8_SPLIT_TRACE
__device__ unsigned long long g_split_trace[512 * 4 * 16 * 4];
__device__ __forceinline__ void split_stamp(int iteration, int wave, int step, int slot) {
  if (iteration == 3 && (threadIdx.x & 63) == 0)
    g_split_trace[((blockIdx.x * 4 + wave) * 16 + step) * 4 + slot] = __builtin_amdgcn_s_memtime();
}
__device__ unsigned long long g_split_trace_epilogue[512 * 4 * 4];
__device__ __forceinline__ void split_stamp_epilogue(int iteration, int wave, int slot) {
  if (iteration == 3 && (threadIdx.x & 63) == 0) g_split_trace_epilogue[(blockIdx.x * 4 + wave) * 4 + slot] = __builtin_amdgcn_s_memtime();
}
#define RL8_SPLIT_STAMP(it, w, s, slot) split_stamp(it, w, s, slot)
#define RL8_SPLIT_STAMP_E(it, w, slot) split_stamp_epilogue(it, w, slot)
#else
#define RL8_SPLIT_STAMP(it, w, s, slot)
#define RL8_SPLIT_STAMP_E(it, w, slot)
#endif

// [stage 0][stage 1][head partials of the upper column half: [128 rows][k_out]][b2 | w3: (1 + k_out) x 1 KiB]
constexpr int split_forward_lds_bytes(int k_out) {
  return 2 * kSplitStageBytes + kSplitRows * k_out * 4 + (1 + k_out) * kHidden * 4;
}
static_assert(split_forward_lds_bytes(4) <= 80 * 1024, "two workgroups per CU");

template <int DIN, int NOUT, bool SAVE>
__global__ __launch_bounds__(kBlock, (DIN == 0 || NOUT == 0) ? 1 : 2) void mlp_tower_forward_split_kernel(
    const float *__restrict__ x, int64_t m, int d_in_rt, const float *__restrict__ w1,
    const float *__restrict__ b1, const void *__restrict__ w2s, const float *__restrict__ b2,
    const float *__restrict__ w3, const float *__restrict__ b3, int n_out_rt,
    float *__restrict__ out, float *__restrict__ save_h1, float *__restrict__ save_h2,
    uint32_t *__restrict__ save_gate2) {
  constexpr int kIn = DIN > 0 ? DIN : kMaxIn;
  constexpr int kOut = NOUT > 0 ? pad_out(NOUT) : kMaxOut;
  const int d_in = DIN > 0 ? DIN : d_in_rt;
  const int n_out = NOUT > 0 ? NOUT : n_out_rt;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // [stage 0: A | B][stage 1: A | B][head partials: [2 column halves][128 rows][kOut]]
  const unsigned lds0 = lds_offset(smem);
  const int tid = threadIdx.x, lane = tid & 63, l32 = lane & 31, hh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;  // rows [64 wr, +64), columns [128 wc, +128)
  // Producer role: row `prow` of the macro tile and k-half `pkh` of every step; the
  // k-half is wave-uniform, so the eight rows of W1 / b1 a step needs come through
  // the scalar cache (no vector loads, no registers) and are used as scalar operands.
  // (Lanes 4i..4i+3 = one row -- 64 contiguous bytes per row for the h1 stores -- was
  // tried while h1 was still stored; it measured no faster and needs per-lane weights.)
  const int prow = tid & 127, pkh = wave >> 1;

  // Per-lane LDS addresses (stage 0; stage 1 = + kSplitStageBytes).
  const unsigned a_read = lds0 + hh * kSplitKhStride + (64 * wr + l32) * 16;
  const unsigned b_read = lds0 + kSplitABytes + (4 * wc * 3) * 1024 + lane * 16;
  const unsigned a_write = lds0 + pkh * kSplitKhStride + prow * 16;
  const __amdgpu_buffer_rsrc_t w2rsrc = buffer_rsrc(w2s, kSplitPackedBytes);

  const int64_t tiles = (m + kSplitRows - 1) / kSplitRows;
  const int64_t stride = gridDim.x;

  // Producer state: the tile whose h1 chunks are being produced (runs one step
  // ahead of the consumer, so it moves to the next tile before step 15).
  float px[kIn];
  [[maybe_unused]] float xn[kIn];
  int64_t p_r0 = (int64_t)blockIdx.x * kSplitRows;
  auto rows_from = [&](int64_t r0) {
    const int64_t left = m - r0;
    return left <= 0 ? 0 : left < kSplitRows ? (int)left : kSplitRows;
  };
  int p_rows = rows_from(p_r0);
  // (addresses: uniform tile base + 32-bit lane offset -- 64-bit per-lane pointers cost
  // registers the matrix loop does not have)
  auto load_x = [&](float (&dst)[kIn], int64_t r0) {
    const int rows = rows_from(r0);
    const float *base = x + r0 * d_in;
#pragma unroll
    for (int i = 0; i < kIn; ++i) dst[i] = (prow < rows && i < d_in) ? base[(unsigned)(prow * d_in + i)] : 0.0f;
  };
  // (wide observations: the next tile's row is loaded at the tile switch instead of a tile ahead)
  constexpr bool kPrefetchX = DIN > 0;  // (compiled widths: a tile ahead; run-time widths: at the tile switch)
  load_x(px, p_r0);
  if constexpr (kPrefetchX) load_x(xn, p_r0 + stride * kSplitRows);

  // Chunk `ks` of the producer's tile -> stage `stage`.
  auto request_b = [&](int ks, int stage) {
#pragma unroll
    for (int u = 0; u < 6; ++u) {
      const int block = wave * 6 + u;  // 24 one-KiB blocks per step, six per wave
      __builtin_amdgcn_raw_ptr_buffer_load_lds(w2rsrc, smem + stage * kSplitStageBytes + kSplitABytes + block * 1024,
                                               16, lane * 16, (ks * 24 + block) * 1024, 0, 0);
    }
  };
  // planes[p]: this thread's fragment (row prow, eight k) of plane p.
  auto produce_a = [&](int ks, u32x4 (&planes)[3]) {
    const int kb = __builtin_amdgcn_readfirstlane(16 * ks + 8 * pkh);
    float h[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float v = b1[kb + e];
#pragma unroll
      for (int i = 0; i < kIn; ++i)
        if (DIN > 0 || i < d_in) v = __builtin_fmaf(px[i], w1[(kb + e) * d_in + i], v);
      h[e] = relu1(v);
    }
    if constexpr (SAVE && !(kSplitDiagSkip & 32)) {
      if (save_h1 != nullptr && prow < p_rows) {  // (h1 is optional: the bf16-plane backward recomputes it)
        f32x4 *dst = reinterpret_cast<f32x4 *>(save_h1 + p_r0 * kHidden + kb + (unsigned)(prow * kHidden));
        dst[0] = f32x4{h[0], h[1], h[2], h[3]};
        dst[1] = f32x4{h[4], h[5], h[6], h[7]};
      }
    }
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
      uint32_t hi, mid, lo;
      split_pair(h[e], h[e + 1], hi, mid, lo);
      planes[0][e >> 1] = hi;
      planes[1][e >> 1] = mid;
      planes[2][e >> 1] = lo;
    }
  };
  auto write_a = [&](int stage, const u32x4 (&planes)[3]) {
    const unsigned addr = a_write + stage * kSplitStageBytes;
    lds_write_b128<0>(addr, planes[0]);
    lds_write_b128<kSplitPlaneStride>(addr, planes[1]);
    lds_write_b128<2 * kSplitPlaneStride>(addr, planes[2]);
  };
  auto step_barrier = [&]() {
    // vmcnt(0): the direct-to-LDS weight loads have landed (and, in-order, every
    // older store -- this step's h1 stores were issued a matrix group or more ago).
    // NOT vmcnt(2) "everything but the step's two stores": a register spill
    // anywhere behind those stores is a vector-memory instruction the count does
    // not know about, and it measured no faster.
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  };

  // Epilogue constants, per register rather than per lane in the transposed
  // accumulator layout: b2 and the rows of W3 (zero rows up to kOut) live in LDS and
  // are fetched as 16-byte quads of four consecutive columns.
  {
    float *consts = reinterpret_cast<float *>(smem + 2 * kSplitStageBytes + kSplitRows * kOut * 4);
    consts[tid] = b2[tid];
#pragma unroll
    for (int q = 0; q < kOut; ++q) consts[(1 + q) * kHidden + tid] = q < n_out ? w3[q * kHidden + tid] : 0.0f;
    // (visible to every wave behind the prologue's step barrier below)
  }

  f32x16 acc[2][4];
  [[maybe_unused]] int trace_it = -1;  // (tuning builds: tile iteration, see RL8_SPLIT_STAMP)
  int64_t r0 = p_r0;  // consumer's tile

  // One k-step: consume stage P (chunk s) while producing chunk s+1 -- of the
  // next tile when s = 15 -- into the other stage.
  auto do_step = [&](auto first_tag, auto parity_tag, int s) {
    constexpr bool FIRST = decltype(first_tag)::value;
    constexpr int P = decltype(parity_tag)::value;
    const int ks = (s + 1) & (kSplitSteps - 1);
    RL8_SPLIT_STAMP(trace_it, wave, s, 0);
    request_b(ks, P ^ 1);
    const unsigned ar = a_read + P * kSplitStageBytes, br = b_read + P * kSplitStageBytes;
    SplitFrags f;
    f.ah[0] = lds_read_b128<0>(ar);
    f.ah[1] = lds_read_b128<512>(ar);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) f.bh[nt] = nt == 0   ? lds_read_b128<0>(br)
                                              : nt == 1 ? lds_read_b128<3 * 1024>(br)
                                              : nt == 2 ? lds_read_b128<6 * 1024>(br)
                                                        : lds_read_b128<9 * 1024>(br);
    f.am[0] = lds_read_b128<kSplitPlaneStride>(ar);
    f.am[1] = lds_read_b128<kSplitPlaneStride + 512>(ar);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) f.bm[nt] = nt == 0   ? lds_read_b128<1024>(br)
                                              : nt == 1 ? lds_read_b128<4 * 1024>(br)
                                              : nt == 2 ? lds_read_b128<7 * 1024>(br)
                                                        : lds_read_b128<10 * 1024>(br);
    if (s == kSplitSteps - 1) {  // the producer moves on to the next tile
      p_r0 += stride * kSplitRows;
      p_rows = rows_from(p_r0);
      if constexpr (kPrefetchX) {
#pragma unroll
        for (int i = 0; i < kIn; ++i) px[i] = xn[i];
        load_x(xn, p_r0 + stride * kSplitRows);
      } else {
        load_x(px, p_r0);
      }
    }
    // The next chunk's arithmetic is left to the scheduler to interleave with the
    // first matrix group (VALU instructions issue beside bf16 MFMAs for free);
    // its fragments go to LDS behind that group.
    u32x4 planes[3];
    produce_a(ks, planes);
    wait_lds_all(f);
    RL8_SPLIT_STAMP(trace_it, wave, s, 1);
    split_mma_t<FIRST>(f.am, f.bm, acc);
    split_mma_t<false>(f.ah, f.bm, acc);
    split_mma_t<false>(f.am, f.bh, acc);
    __builtin_amdgcn_sched_barrier(0);
    write_a(P ^ 1, planes);
    // lo planes into the registers of the mid planes
    f.am[0] = lds_read_b128<2 * kSplitPlaneStride>(ar);
    f.am[1] = lds_read_b128<2 * kSplitPlaneStride + 512>(ar);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) f.bm[nt] = nt == 0   ? lds_read_b128<2 * 1024>(br)
                                              : nt == 1 ? lds_read_b128<5 * 1024>(br)
                                              : nt == 2 ? lds_read_b128<8 * 1024>(br)
                                                        : lds_read_b128<11 * 1024>(br);
    __builtin_amdgcn_sched_barrier(0);
    split_mma_t<false>(f.ah, f.bh, acc);
    __builtin_amdgcn_sched_barrier(0);
    wait_lds_all(f);
    split_mma_t<false>(f.ah, f.bm, acc);
    split_mma_t<false>(f.am, f.bh, acc);
    __builtin_amdgcn_sched_barrier(0);
    RL8_SPLIT_STAMP(trace_it, wave, s, 2);
    step_barrier();
    RL8_SPLIT_STAMP(trace_it, wave, s, 3);
  };
  // Prologue: chunk 0 of the first tile.
  using T = std::true_type;
  using F = std::false_type;
  using P0 = std::integral_constant<int, 0>;
  using P1 = std::integral_constant<int, 1>;

  if ((int64_t)blockIdx.x < tiles) {
    request_b(0, 0);
    u32x4 planes[3];
    produce_a(0, planes);
    write_a(0, planes);
    step_barrier();
  }

  for (int64_t tile = blockIdx.x; tile < tiles; tile += stride) {
    ++trace_it;
    r0 = tile * kSplitRows;
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

    // Epilogue on the TRANSPOSED accumulators (split_mma_t): this lane holds sample
    // row 64 wr + 32 mt + l32 and, of column block 128 wc + 32 nt, the sixteen columns
    // 8 g + 4 hh + e (register r = 4 g + e).  So
    //   * h2 leaves as 16-byte stores (four consecutive columns per lane, the two
    //     half-waves side by side: 32 contiguous bytes per row and instruction, a full
    //     128-byte line per row over g = 0..3) -- 32 stores per wave and tile instead
    //     of 128 dword stores, the largest item of the old epilogue;
    //   * the ReLU gate bits of a row are built in that row's own lane (add + funnel
    //     shift per element) instead of 128 ballots and 256 v_writelane;
    //   * the head is a dot product down the lane's registers against W3 quads from
    //     LDS plus ONE half-wave exchange, instead of a five-level DPP reduction per
    //     eight rows.
    // b2 / W3 come from the constants block in LDS (hand-issued reads, explicit waits).
    // (tuning builds, bit 4096: every tile's h2 lands on the first tile's lines -- the stores are issued but stay in L2)
    RL8_SPLIT_STAMP_E(trace_it, wave, 0);
    const __amdgpu_buffer_rsrc_t h2rsrc =
        buffer_rsrc(SAVE ? save_h2 + ((kSplitDiagSkip & 4096) ? (r0 & 0x1ffff) : r0) * kHidden : nullptr, rows * kHidden * 4);
    const int l32 = lane_id() & 31, hh = lane_id() >> 5;  // (recomputed: see lane_id)
    const unsigned outp = lds_offset(smem) + 2 * kSplitStageBytes;
    const unsigned constp = outp + kSplitRows * kOut * 4 + (128 * wc + 4 * hh) * 4;
    constexpr int kChains = kOut <= 2 ? 4 : 2;  // (registers: three outputs x four chains spilled)
    float part[2][kOut][kChains];
    [[maybe_unused]] uint32_t gate_words[2][4];
    // h2 goes to HBM through a per-wave transpose in LDS (stage 1 is dead between the
    // barrier of step 15 and the barrier of this epilogue, for every wave): written as
    // the accumulators hold it (lane = row), read back eight lanes per row, so a store
    // instruction is eight full 128-byte lines instead of thirty-two 32-byte pieces
    // (scattered 32-byte pieces measured 712 us per 2^20 rows at best, 1 185 us with
    // the streaming policy, against 588 us with the stores compiled out).
    constexpr int kH2Pitch = 128 + 16;
    static_assert(4 * 64 * kH2Pitch <= kSplitStageBytes, "transpose scratch fits in stage 1");
    [[maybe_unused]] const unsigned t_base = lds_offset(smem) + kSplitStageBytes + wave * (64 * kH2Pitch);
    [[maybe_unused]] const unsigned t_write = t_base + l32 * kH2Pitch + 16 * hh;
    [[maybe_unused]] const unsigned t_read = t_base + (lane_id() >> 3) * kH2Pitch + (lane_id() & 7) * 16;
    [[maybe_unused]] const int h2_voff = ((64 * wr + (lane_id() >> 3)) * kHidden + 128 * wc + 4 * (lane_id() & 7)) * 4;
    // The column blocks in groups of kGroup quads (g = eight columns): b2 and the kOut
    // rows of W3 for the group from LDS, then bias + ReLU, the h2 quads into the transpose
    // scratch, gate nibbles and head products.  (Wide heads take two quads at a time:
    // sixteen W3 quads plus the block read back would not fit.)
    // Software-pipelined over the groups: the LDS is busy with the CU's other workgroup's
    // operand reads, so every lgkmcnt(0) in here cost 500+ cycles (twelve of them per tile
    // measured 13 500 cycles for ~800 VALU instructions).  The next group's b2 quads are
    // requested as soon as bias + ReLU has consumed this group's, its W3 quads as soon as
    // the head products have, and the transposed block is waited for by COUNT (in-order
    // LDS returns; no scalar load is in flight here), behind the gate and head arithmetic.
    constexpr int kGroup = kOut >= 4 ? 1 : kOut >= 2 ? 2 : 4;  // (registers: more W3 quads in flight beside the block spilled)
    constexpr int kStages = 4 * (4 / kGroup);  // (nt, g0) pairs
    constexpr bool kStore = SAVE && !(kSplitDiagSkip & 8);
    u32x4 bq[kGroup], wq[kOut][kGroup];
    auto request_b2 = [&](int stage) {
      const int nt = stage / (4 / kGroup), g0 = (stage % (4 / kGroup)) * kGroup;
#pragma unroll
      for (int gi = 0; gi < kGroup; ++gi) bq[gi] = lds_read_b128<0>(constp + (32 * nt + 8 * (g0 + gi)) * 4);
    };
    auto request_w3 = [&](int stage) {
      const int nt = stage / (4 / kGroup), g0 = (stage % (4 / kGroup)) * kGroup;
#pragma unroll
      for (int gi = 0; gi < kGroup; ++gi) {
        const unsigned a = constp + (32 * nt + 8 * (g0 + gi)) * 4;
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
    request_b2(0);
    request_w3(0);
    [[maybe_unused]] u32x4 t_rows[8];
    auto read_block = [&]() {  // the block back, eight lanes per row (in order behind the writes: same wave)
      t_rows[0] = lds_read_b128<0 * 8 * kH2Pitch>(t_read);
      t_rows[1] = lds_read_b128<1 * 8 * kH2Pitch>(t_read);
      t_rows[2] = lds_read_b128<2 * 8 * kH2Pitch>(t_read);
      t_rows[3] = lds_read_b128<3 * 8 * kH2Pitch>(t_read);
      t_rows[4] = lds_read_b128<4 * 8 * kH2Pitch>(t_read);
      t_rows[5] = lds_read_b128<5 * 8 * kH2Pitch>(t_read);
      t_rows[6] = lds_read_b128<6 * 8 * kH2Pitch>(t_read);
      t_rows[7] = lds_read_b128<7 * 8 * kH2Pitch>(t_read);
    };
#pragma unroll
    for (int stage = 0; stage < kStages; ++stage) {
      const int nt = stage / (4 / kGroup), g0 = (stage % (4 / kGroup)) * kGroup;
      const bool last_of_block = g0 + kGroup == 4;
      // this group's quads (requested a stage ago; everything older has landed too)
#pragma unroll
      for (int gi = 0; gi < kGroup; ++gi) wait_lds<0>(bq[gi]);
#pragma unroll
      for (int q = 0; q < kOut; ++q)
#pragma unroll
        for (int gi = 0; gi < kGroup; ++gi) wait_lds<0>(wq[q][gi]);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int gi = 0; gi < kGroup; ++gi) {
          const int g = g0 + gi;
#pragma unroll
          for (int e = 0; e < 4; ++e)  // (not __builtin_bit_cast on a vector-element lvalue: it reads element 0)
            acc[mt][nt][4 * g + e] = relu1(acc[mt][nt][4 * g + e] + __uint_as_float(bq[gi][e]));
        }
      if (stage + 1 < kStages) request_b2(stage + 1);
      if constexpr (kStore) {
        // h2 block [64 rows][32 columns] of this wave -> its transpose scratch (row pitch
        // 144 B: the eight lanes of a b128 phase hit eight distinct 16-byte slots, writing
        // as well as reading)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int gi = 0; gi < kGroup; ++gi) {
            const int g = g0 + gi;
            const f32x4 v = {acc[mt][nt][4 * g], acc[mt][nt][4 * g + 1], acc[mt][nt][4 * g + 2], acc[mt][nt][4 * g + 3]};
            const u32x4 u = __builtin_bit_cast(u32x4, v);
            if (mt == 0) {
              g == 0 ? lds_write_b128<0>(t_write, u) : g == 1 ? lds_write_b128<32>(t_write, u)
              : g == 2 ? lds_write_b128<64>(t_write, u) : lds_write_b128<96>(t_write, u);
            } else {
              g == 0 ? lds_write_b128<32 * kH2Pitch>(t_write, u) : g == 1 ? lds_write_b128<32 * kH2Pitch + 32>(t_write, u)
              : g == 2 ? lds_write_b128<32 * kH2Pitch + 64>(t_write, u) : lds_write_b128<32 * kH2Pitch + 96>(t_write, u);
            }
          }
        if (last_of_block) read_block();
      }
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
        for (int gi = 0; gi < kGroup; ++gi) {
          const int g = g0 + gi;
          if constexpr (SAVE && !(kSplitDiagSkip & 65536)) {  // (tuning builds, bit 65536: no gate bits)
            // gate of h2, bit c of word [row][4 wc + nt] <=> column 32 (4 wc + nt) + c > 0.
            // h2 >= +0 here, so "h2 > 0" is bit 31 of (bits(h2) + 0x7fffffff); four of
            // them are funnel-shifted into a nibble (element 0 lowest), the nibble goes
            // to bit 8 g + 4 hh.  The other half-wave holds the interleaved nibbles.
            uint32_t nib = 0;
#pragma unroll
            for (int e = 3; e >= 0; --e)
              nib = __builtin_amdgcn_alignbit(nib, __float_as_uint(acc[mt][nt][4 * g + e]) + 0x7fffffffu, 31);
            gate_words[mt][nt] = (g == 0 ? 0u : gate_words[mt][nt]) | (nib << (8 * g + 4 * hh));
          }
          // head partials of this row: four (two for wide heads) independent chains per
          // output, 16 (32) terms each over the tile, so the sum is not one 64-term
          // chain and the fmas do not wait on each other
#pragma unroll
          for (int q = 0; q < kOut; ++q) {
            float p = (nt == 0 && g < kChains) ? 0.0f : part[mt][q][g % kChains];
#pragma unroll
            for (int e = 0; e < 4; ++e) p = __builtin_fmaf(acc[mt][nt][4 * g + e], __uint_as_float(wq[q][gi][e]), p);
            part[mt][q][g % kChains] = p;
          }
        }
      }
      if (stage + 1 < kStages) request_w3(stage + 1);
      if constexpr (kStore) {
        if (last_of_block) {
          // the block's eight reads are older than the next stage's W3 quads just requested
          // (its b2 quads went out ahead of the block): wait for "all but those" (in-order
          // returns; tools/check_inflight_regs.py caught the first version counting both)
          constexpr int kNewer = kGroup * kOut;
          if (stage + 1 < kStages) {
            wait_lds<kNewer>(t_rows[0], t_rows[1], t_rows[2], t_rows[3]);
            wait_lds<kNewer>(t_rows[4], t_rows[5], t_rows[6], t_rows[7]);
          } else {
            wait_lds<0>(t_rows[0], t_rows[1], t_rows[2], t_rows[3]);
            wait_lds<0>(t_rows[4], t_rows[5], t_rows[6], t_rows[7]);
          }
#pragma unroll
          for (int i = 0; i < 8; ++i) {  // rows 8 i + (lane >> 3), columns 32 nt + 4 (lane & 7) .. + 3
            if constexpr ((kSplitDiagSkip & 32768) != 0) {  // tuning builds: transposes without the global stores
              asm volatile("" ::"v"(t_rows[i]));
            } else {
              __builtin_amdgcn_raw_buffer_store_b128(t_rows[i], h2rsrc, h2_voff + (8 * i * kHidden + 32 * nt) * 4, 0, RL8_H2_STORE_AUX);
            }
          }
          if constexpr ((kSplitDiagSkip & 16384) != 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // tuning builds: expose the store latency
        }
      }
    }
    RL8_SPLIT_STAMP_E(trace_it, wave, 1);
    // Half-wave exchange: lane (l32, hh) ends with the total of row 64 wr + 32 hh + l32
    // over this wave's 128 columns (v_permlane32_swap: [A_lo, B_lo] and [A_hi, B_hi]).
    float total[kOut];
#pragma unroll
    for (int q = 0; q < kOut; ++q) {
      float p0 = part[0][q][0] + part[0][q][1], p1 = part[1][q][0] + part[1][q][1];
      if constexpr (kChains == 4) {
        p0 += part[0][q][2] + part[0][q][3];
        p1 += part[1][q][2] + part[1][q][3];
      }
      const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(p0), __float_as_uint(p1), false, false);
      total[q] = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
    }
    const int my_row = 64 * wr + lane_id();
    if constexpr (SAVE && !(kSplitDiagSkip & 65536)) {
      if (save_gate2 != nullptr) {
        // full words = own nibbles | the other half-wave's; lane (l32, hh) keeps row
        // 64 wr + 32 hh + l32's four words and stores them as one 16-byte piece
        u32x4 words;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          const auto s0 = __builtin_amdgcn_permlane32_swap(gate_words[0][nt], gate_words[0][nt], false, false);
          const auto s1 = __builtin_amdgcn_permlane32_swap(gate_words[1][nt], gate_words[1][nt], false, false);
          const uint32_t w0 = s0[0] | s0[1], w1 = s1[0] | s1[1];
          words[nt] = hh ? w1 : w0;
        }
        if (my_row < rows) {
          uint32_t *dst = save_gate2 + r0 * 8;  // uniform base, 32-bit lane offset
          *reinterpret_cast<u32x4 *>(dst + (unsigned)(my_row * 8 + 4 * wc)) = words;
        }
      }
    }
    // The two column halves of the workgroup meet in LDS: the upper half (wc = 1)
    // parks its totals, the lower half adds its own and stores the outputs.
    if (wc == 1) {
#pragma unroll
      for (int q = 0; q < kOut; ++q)
        if (q < n_out) lds_write_b32(outp + (my_row * kOut + q) * 4, total[q]);
    }
    RL8_SPLIT_STAMP_E(trace_it, wave, 2);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    RL8_SPLIT_STAMP_E(trace_it, wave, 3);
    if (wc == 0 && my_row < rows) {
#pragma unroll
      for (int q = 0; q < kOut; ++q)
        if (q < n_out)
          (out + r0 * n_out)[(unsigned)(my_row * n_out + q)] = total[q] + lds_read_b32(outp + (my_row * kOut + q) * 4) + b3[q];
    }
    // (the next tile's head partials are parked a full tile later, behind sixteen
    // barriers: no extra barrier needed here)
  }
}

template <int DIN, int NOUT, bool SAVE>
static int launch_forward_split(int grid, hipStream_t s, const float *x, int64_t m, int d_in, const float *w1,
                                const float *b1, const void *w2s, const float *b2, const float *w3,
                                const float *b3, int n_out, float *out, float *h1, float *h2, uint32_t *gate) {
  constexpr int kOut = NOUT > 0 ? pad_out(NOUT) : kMaxOut;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&mlp_tower_forward_split_kernel<DIN, NOUT, SAVE>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipGetLastError();
    attr_set = true;
  }
  mlp_tower_forward_split_kernel<DIN, NOUT, SAVE><<<grid, kBlock, split_forward_lds_bytes(kOut), s>>>(
      x, m, d_in, w1, b1, w2s, b2, w3, b3, n_out, out, h1, h2, gate);
  return launch_status();
}

template <int DIN, int NOUT>
static int launch_forward_split_save(int grid, hipStream_t s, const float *x, int64_t m, int d_in, const float *w1,
                                     const float *b1, const void *w2s, const float *b2, const float *w3,
                                     const float *b3, int n_out, float *out, float *h1, float *h2, uint32_t *gate) {
  return h2 ? launch_forward_split<DIN, NOUT, true>(grid, s, x, m, d_in, w1, b1, w2s, b2, w3, b3, n_out, out, h1, h2, gate)
            : launch_forward_split<DIN, NOUT, false>(grid, s, x, m, d_in, w1, b1, w2s, b2, w3, b3, n_out, out, h1, h2, gate);
}

template <int DIN>
static int dispatch_forward_split_nout(int n_out, int grid, hipStream_t s, const float *x, int64_t m, int d_in,
                                       const float *w1, const float *b1, const void *w2s, const float *b2,
                                       const float *w3, const float *b3, float *out, float *h1, float *h2,
                                       uint32_t *gate) {
  switch (n_out) {
    case 1: return launch_forward_split_save<DIN, 1>(grid, s, x, m, d_in, w1, b1, w2s, b2, w3, b3, n_out, out, h1, h2, gate);
    case 2: return launch_forward_split_save<DIN, 2>(grid, s, x, m, d_in, w1, b1, w2s, b2, w3, b3, n_out, out, h1, h2, gate);
    case 3: return launch_forward_split_save<DIN, 3>(grid, s, x, m, d_in, w1, b1, w2s, b2, w3, b3, n_out, out, h1, h2, gate);
    default: return RL8_ESIZE;
  }
}

// ---- backward ("dgrad" half) on the same scheme --------------------------------
//   dZ2 = (dOut x W3) * (h2 > 0)   computed per k-chunk on the VALU (thread = row,
//                                  eight columns), stored for the weight-gradient
//                                  product and split into the A planes;
//   dH1 = dZ2 x W2                 bf16-plane MFMAs (B = W2 packed transposed);
//   dZ1 = dH1 * (h1 > 0)           accumulator epilogue, folded into dW1 / db1
//                                  (lane = column: per-lane running sums).
// The head gradients (db2, dW3, db3) are column sums over rows of quantities that
// this kernel holds row-per-thread; they are formed by mlp_head_grads_kernel, an
// HBM-streaming kernel with thread = column, into the same partial rows.
template <int DIN, int NOUT>
__global__ __launch_bounds__(kBlock, (DIN == 0 || NOUT == 0) ? 1 : 2) void mlp_tower_backward_split_kernel(
    const float *__restrict__ x, const float *__restrict__ w1, const float *__restrict__ b1,
    const float *__restrict__ h2, const float *__restrict__ dout, int64_t m, int d_in_rt,
    const void *__restrict__ w2ts, const float *__restrict__ w3, int n_out_rt, float *__restrict__ dz2_out,
    float *__restrict__ partials, int partial_stride, int head_rows, const uint32_t *__restrict__ gate2) {
  constexpr int kIn = DIN > 0 ? DIN : kMaxIn;
  constexpr int kOut = NOUT > 0 ? pad_out(NOUT) : kMaxOut;
  const int d_in = DIN > 0 ? DIN : d_in_rt;
  const int n_out = NOUT > 0 ? NOUT : n_out_rt;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const unsigned lds0 = lds_offset(smem);
  const int tid = threadIdx.x, lane = tid & 63, l32 = lane & 31, hh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int prow = tid & 127, pkh = wave >> 1;  // producer role: row, wave-uniform k-half (see the forward kernel)
  // gate2 given: the ReLU gate of h2 comes as bits (the forward kernel's save_gate2),
  // one 4-KiB block per tile copied to LDS by direct-to-LDS loads, instead of 1 KiB
  // of h2 per row through registers a step ahead -- that HBM latency, forced to
  // fit one k-step by the step barrier's vmcnt(0), was 40 % of this kernel.
  const bool use_bits = gate2 != nullptr;
  const unsigned gate_lds = lds0 + 2 * kSplitStageBytes + kHidden * (1 + kIn) * 4;  // [128 rows][8 words]
  uint32_t g0 = 0u;  // word 0 of row prow of the NEXT tile (its chunk 0 is produced before the block lands)
  // Wide observations (d_in >= 3): the running column sums [256][1 + d_in] and the 4-KiB
  // gate block do not both fit beside the two chunk stages in 80 KB, and at one workgroup
  // per CU this kernel lost a third of its speed (CartPole: 29 ms per 2^25 rows against 19
  // for d_in = 1).  There the gate word of (row, two k-steps) is fetched into a register two
  // steps ahead instead (4 bytes per thread and pair of steps, L2 hits after a row's first
  // word): `g0` is the word in use, `gnext` the one on its way.  No gate block in LDS.
  constexpr bool kRegGate = kIn >= 3 && DIN > 0;
  [[maybe_unused]] uint32_t gnext = 0u;
  const unsigned a_read = lds0 + hh * kSplitKhStride + (64 * wr + l32) * 16;
  const unsigned b_read = lds0 + kSplitABytes + (4 * wc * 3) * 1024 + lane * 16;
  const unsigned a_write = lds0 + pkh * kSplitKhStride + prow * 16;
  const __amdgpu_buffer_rsrc_t w2rsrc = buffer_rsrc(w2ts, kSplitPackedBytes);

  const int64_t tiles = (m + kSplitRows - 1) / kSplitRows;
  const int64_t stride = gridDim.x;

  // Producer state (one step ahead of the consumer; moves to the next tile before step 15).
  int64_t p_tile = blockIdx.x;
  int p_rows;                // valid rows of the producer's tile
  float dr[kOut], dn[kOut];  // dOut of row prow of the producer's tile / of the tile after
  auto rows_in_tile = [&](int64_t tile) {
    const int64_t left = m - tile * kSplitRows;
    return left <= 0 ? 0 : left < kSplitRows ? (int)left : kSplitRows;
  };
  // (addresses: uniform tile base + 32-bit lane offset -- 64-bit per-lane pointers cost
  // registers the matrix loop does not have)
  auto load_dout = [&](float (&dst)[kOut], int64_t tile) {
    const int rows = rows_in_tile(tile);
    const float *base = dout + tile * kSplitRows * n_out;
#pragma unroll
    for (int q = 0; q < kOut; ++q) dst[q] = (prow < rows && q < n_out) ? base[(unsigned)(prow * n_out + q)] : 0.0f;
  };
  [[maybe_unused]] auto load_gate_word = [&](int64_t tile, int word) {
    const int rows = rows_in_tile(tile);
    gnext = prow < rows ? (gate2 + tile * (kSplitRows * 8))[(unsigned)(prow * 8 + word)] : 0u;
  };
  load_dout(dr, p_tile);
  load_dout(dn, p_tile + stride);
  // h2 of the chunk produced next (only without gate bits): this thread's eight columns,
  // requested a step ahead; rows past the end read as zero (gate closed).
  float4 hq[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
  const unsigned lane_off = prow * kHidden + 8 * pkh;
  auto load_h2 = [&](float4 (&dst)[2], int64_t tile, int ks) {
    const int rows = rows_in_tile(tile);
    const float *base = h2 + tile * (kSplitRows * kHidden) + 16 * ks;
    dst[0] = dst[1] = make_float4(0.f, 0.f, 0.f, 0.f);  // (rows past the end: gate closed)
    if (!(kSplitDiagSkip & 64) && prow < rows) {
      dst[0] = *reinterpret_cast<const float4 *>(base + lane_off);
      dst[1] = *reinterpret_cast<const float4 *>(base + lane_off + 4);
    }
  };
  auto request_b = [&](int ks, int stage) {
#pragma unroll
    for (int u = 0; u < 6; ++u) {
      const int block = wave * 6 + u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(w2rsrc, smem + stage * kSplitStageBytes + kSplitABytes + block * 1024,
                                               16, lane * 16, (ks * 24 + block) * 1024, 0, 0);
    }
  };
  // Gate block of `tile` -> LDS: wave w copies rows 32w .. 32w+31 (1 KiB, contiguous);
  // rows past the end of the data arrive as zeros (gate closed).
  auto request_gate = [&](int64_t tile) {
    const int rows = rows_in_tile(tile);
    const __amdgpu_buffer_rsrc_t rsrc = buffer_rsrc(rows > 0 ? gate2 + tile * (kSplitRows * 8) : gate2, rows * 32);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, smem + 2 * kSplitStageBytes + kHidden * (1 + kIn) * 4 + wave * 1024, 16,
                                             lane * 16, wave * 1024, 0, 0);
  };
  auto load_g0 = [&](int64_t tile) {
    const int rows = rows_in_tile(tile);
    g0 = prow < rows ? (gate2 + tile * (kSplitRows * 8))[(unsigned)(prow * 8)] : 0u;
  };
  // hv: h2 values (use_bits false) -- or unused; from_regs: chunk 0 of a tile whose
  // gate block has not landed yet takes its bits from g0.
  auto produce_a = [&](const float4 (&hv)[2], int ks, u32x4 (&planes)[3], bool from_regs = false) {
    const int kb = __builtin_amdgcn_readfirstlane(16 * ks + 8 * pkh);  // W3[q][kb + e]: uniform, through the scalar cache
    uint32_t gword = g0;
    if constexpr (!kRegGate) {
      if (use_bits && !from_regs) gword = __float_as_uint(lds_read_b32(gate_lds + (prow * 8 + (ks >> 1)) * 4));
    }
    const uint32_t byte = gword >> (16 * (ks & 1) + 8 * pkh);
    const float hval[8] = {hv[0].x, hv[0].y, hv[0].z, hv[0].w, hv[1].x, hv[1].y, hv[1].z, hv[1].w};
    float dz[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float g = dr[0] * w3[kb + e];
#pragma unroll
      for (int q = 1; q < kOut; ++q)
        if (NOUT > 0 ? q < NOUT : q < n_out) g = __builtin_fmaf(dr[q], w3[q * kHidden + kb + e], g);
      const bool open = use_bits ? ((byte >> e) & 1u) != 0 : hval[e] > 0.0f;
      dz[e] = open ? g : 0.0f;
    }
    if constexpr (!(kSplitDiagSkip & 128)) {
      if (dz2_out != nullptr && prow < p_rows) {  // (optional: the fused weight-gradient kernel re-forms dZ2)
        f32x4 *dst = reinterpret_cast<f32x4 *>(dz2_out + p_tile * (kSplitRows * kHidden) + 16 * ks + lane_off);
        dst[0] = f32x4{dz[0], dz[1], dz[2], dz[3]};
        dst[1] = f32x4{dz[4], dz[5], dz[6], dz[7]};
      }
    }
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
      uint32_t hi, mid, lo;
      split_pair(dz[e], dz[e + 1], hi, mid, lo);
      planes[0][e >> 1] = hi;
      planes[1][e >> 1] = mid;
      planes[2][e >> 1] = lo;
    }
  };
  auto write_a = [&](int stage, const u32x4 (&planes)[3]) {
    const unsigned addr = a_write + stage * kSplitStageBytes;
    lds_write_b128<0>(addr, planes[0]);
    lds_write_b128<kSplitPlaneStride>(addr, planes[1]);
    lds_write_b128<2 * kSplitPlaneStride>(addr, planes[2]);
  };
  auto step_barrier = [&]() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

  f32x16 acc[2][4];
  // Running column sums [256][1 + kIn] (db1 | dW1 row) live in LDS behind the two
  // chunk stages, not in registers: the matrix loop has none to spare.
  const unsigned colsum = lds0 + 2 * kSplitStageBytes;
  for (int idx = tid; idx < kHidden * (1 + kIn); idx += kBlock) lds_write_b32(colsum + idx * 4, 0.0f);
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

  auto do_step = [&](auto first_tag, auto parity_tag, int s) {
    constexpr bool FIRST = decltype(first_tag)::value;
    constexpr int P = decltype(parity_tag)::value;
    const int ks = (s + 1) & (kSplitSteps - 1);
    request_b(ks, P ^ 1);
    const unsigned ar = a_read + P * kSplitStageBytes, br = b_read + P * kSplitStageBytes;
    SplitFrags f;
    f.ah[0] = lds_read_b128<0>(ar);
    f.ah[1] = lds_read_b128<512>(ar);
    f.bh[0] = lds_read_b128<0>(br);
    f.bh[1] = lds_read_b128<3 * 1024>(br);
    f.bh[2] = lds_read_b128<6 * 1024>(br);
    f.bh[3] = lds_read_b128<9 * 1024>(br);
    f.am[0] = lds_read_b128<kSplitPlaneStride>(ar);
    f.am[1] = lds_read_b128<kSplitPlaneStride + 512>(ar);
    f.bm[0] = lds_read_b128<1024>(br);
    f.bm[1] = lds_read_b128<4 * 1024>(br);
    f.bm[2] = lds_read_b128<7 * 1024>(br);
    f.bm[3] = lds_read_b128<10 * 1024>(br);
    if (s == kSplitSteps - 1) {  // the producer moves on to the next tile
#pragma unroll
      for (int q = 0; q < kOut; ++q) dr[q] = dn[q];
      p_tile += stride;
      p_rows = rows_in_tile(p_tile);
      load_dout(dn, p_tile + stride);
      // every wave is past barrier(14): nobody reads the old gate block any more;
      // the new one lands by this step's barrier, chunk 0 uses g0 meanwhile
      if constexpr (!kRegGate) {
        if (use_bits) request_gate(p_tile);
      }
    }
    if constexpr (kRegGate) {
      // chunk ks opens word ks / 2 when ks is even: take the prefetched word, request the
      // next one (ks = 14: word 0 of the producer's next tile; ks = 0: p_tile has moved on)
      if (use_bits && !(ks & 1)) {
        g0 = gnext;
        load_gate_word(ks == kSplitSteps - 2 ? p_tile + stride : p_tile, ((ks >> 1) + 1) & 7);
      }
    }
    u32x4 planes[3];
    produce_a(hq, ks, planes, s == kSplitSteps - 1);
    if (use_bits) {
      if constexpr (!kRegGate) {
        if (s == kSplitSteps - 3) load_g0(p_tile + stride);  // two steps ahead of its use
      }
    } else {
      // h2 of the chunk after next, into the registers just consumed: used by the
      // next step's produce_a (the step barrier waits for it; it has this step to arrive).
      load_h2(hq, s == kSplitSteps - 2 ? p_tile + stride : p_tile, (s + 2) & (kSplitSteps - 1));
    }
    wait_lds_all(f);
    split_mma<FIRST>(f.am, f.bm, acc);
    split_mma<false>(f.ah, f.bm, acc);
    split_mma<false>(f.am, f.bh, acc);
    __builtin_amdgcn_sched_barrier(0);
    write_a(P ^ 1, planes);
    f.am[0] = lds_read_b128<2 * kSplitPlaneStride>(ar);
    f.am[1] = lds_read_b128<2 * kSplitPlaneStride + 512>(ar);
    f.bm[0] = lds_read_b128<2 * 1024>(br);
    f.bm[1] = lds_read_b128<5 * 1024>(br);
    f.bm[2] = lds_read_b128<8 * 1024>(br);
    f.bm[3] = lds_read_b128<11 * 1024>(br);
    __builtin_amdgcn_sched_barrier(0);
    split_mma<false>(f.ah, f.bh, acc);
    __builtin_amdgcn_sched_barrier(0);
    wait_lds_all(f);
    split_mma<false>(f.ah, f.bm, acc);
    split_mma<false>(f.am, f.bh, acc);
    __builtin_amdgcn_sched_barrier(0);
    step_barrier();
  };
  using T = std::true_type;
  using F = std::false_type;
  using P0 = std::integral_constant<int, 0>;
  using P1 = std::integral_constant<int, 1>;

  p_rows = rows_in_tile(p_tile);
  if ((int64_t)blockIdx.x < tiles) {
    if (use_bits) {
      if constexpr (kRegGate) {
        load_gate_word(p_tile, 0);
        g0 = gnext;
        load_gate_word(p_tile, 1);  // opened by chunk 2, i.e. in step 1
      } else {
        request_gate(p_tile);
        step_barrier();  // (once per kernel: the first gate block has landed)
      }
    } else {
      load_h2(hq, p_tile, 0);
    }
    request_b(0, 0);
    u32x4 planes[3];
    produce_a(hq, 0, planes);
    write_a(0, planes);
    if (!use_bits) load_h2(hq, p_tile, 1);
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

    // Epilogue: dZ1 = dH1 * (h1 > 0) folded into db1 / dW1.  The gate is RECOMPUTED
    // from the observations -- h1 > 0 <=> b1 + x . w1 > 0, the same fma chain as the
    // forward pass, so the same decision bit for bit -- instead of reading 1 KiB of
    // h1 per row back from HBM (d_in is tiny; the observations are needed for dW1
    // anyway).  They arrive through a tile-sized descriptor: rows past the end of
    // a partial tile read as zero and their dH1 is zero (zero dOut).
    const __amdgpu_buffer_rsrc_t xrsrc = buffer_rsrc(x + r0 * d_in, rows * d_in * 4);
    const int l32 = lane_id() & 31, hh = lane_id() >> 5;  // (recomputed: see lane_id)
    const unsigned colsum = lds_offset(smem) + 2 * kSplitStageBytes;
    float w1c[4][kIn], b1c[4];  // this lane's four columns of layer 1 (reloaded per tile: L1 hits)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const int col = 128 * wc + 32 * nt + l32;
      b1c[nt] = b1[col];
#pragma unroll
      for (int i = 0; i < kIn; ++i) w1c[nt][i] = (DIN > 0 || i < d_in) ? w1[col * d_in + i] : 0.0f;
    }
    float db1[4], dw1[4][kIn];  // this tile: this lane's four columns, its half of the rows
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      db1[nt] = 0.0f;
#pragma unroll
      for (int i = 0; i < kIn; ++i) dw1[nt][i] = 0.0f;
    }
    // (narrow observations: the sixteen rows a row tile needs are requested together, up
    // front -- per batch of four, their L1 round trip sat in front of every batch)
    constexpr int kXRows = kIn <= 2 ? 16 : 4;
    constexpr int kBatches = 2 * 16 / kXRows;  // batches of kXRows rows over both row tiles
    auto load_rows = [&](float (&dst)[kXRows][kIn], int batch) {
      const int mt = batch / (16 / kXRows), rx = (batch % (16 / kXRows)) * kXRows;
#pragma unroll
      for (int u = 0; u < kXRows; ++u) {
        const int r = rx + u;
        const int sr = 64 * wr + 32 * mt + (r & 3) + 8 * (r >> 2);  // + 4*hh
#pragma unroll
        for (int i = 0; i < kIn; ++i)
          dst[u][i] = (DIN > 0 || i < d_in) ? buffer_load_f32(xrsrc, (4 * hh * d_in + i) * 4, sr * d_in * 4) : 0.0f;
      }
    };
    // (wide observations: batches of four rows, the NEXT batch's rows requested before the
    // current batch is folded -- eight L1 round trips per tile sat in front of the batches)
    float xbuf[2][kXRows][kIn];
    load_rows(xbuf[0], 0);
#pragma unroll
    for (int batch = 0; batch < kBatches; ++batch) {
      const int mt = batch / (16 / kXRows), rx = (batch % (16 / kXRows)) * kXRows;
      if (batch + 1 < kBatches) load_rows(xbuf[(batch + 1) & 1], batch + 1);
      float (&xv)[kXRows][kIn] = xbuf[batch & 1];
#pragma unroll
      for (int rb = 0; rb < kXRows; rb += 4)
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
          for (int u = 0; u < 4; ++u) dz[u] = select_or_zero(gate[u], acc[mt][nt][rx + rb + u]);
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            db1[nt] += dz[u];
#pragma unroll
            for (int i = 0; i < kIn; ++i) dw1[nt][i] = __builtin_fmaf(dz[u], xv[rb + u][i], dw1[nt][i]);
          }
        }
    }
    // Into the running sums, in a fixed order: the two row halves of a lane pair
    // (DPP-free: one cross-half shuffle), then the wave of rows 0..63, a barrier,
    // the wave of rows 64..127.  (The next tile touches them sixteen barriers later.)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      db1[nt] += __shfl_xor(db1[nt], 32, kWave);
#pragma unroll
      for (int i = 0; i < kIn; ++i) dw1[nt][i] += __shfl_xor(dw1[nt][i], 32, kWave);
    }
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      if (wr == half && hh == 0) {
        // all reads, one wait, all writes (one read-modify-write at a time is a chain of LDS round trips)
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

  // Workgroup partial row: [dW1 (256*d_in) | db1 (256) | ... head gradients (mlp_head_grads_kernel)].
  float *row = partials + (int64_t)blockIdx.x * partial_stride;
  {
    const int t = 64 * wave + lane_id();
    const unsigned a = lds_offset(smem) + 2 * kSplitStageBytes + t * (1 + kIn) * 4;
    row[kHidden * d_in + t] = lds_read_b32(a);
#pragma unroll
    for (int i = 0; i < kIn; ++i)
      if (DIN > 0 || i < d_in) row[t * d_in + i] = lds_read_b32(a + 4 + 4 * i);
    // head_rows >= 0: the head-gradient segments of the first head_rows rows belong
    // to the fused weight-gradient kernel; rows beyond them are zero.
    if (head_rows >= 0 && (int)blockIdx.x >= head_rows)
      for (int idx = kHidden * d_in + kHidden + t; idx < partial_stride; idx += kBlock) row[idx] = 0.0f;
  }
}

// db2 = sum_rows dZ2, dW3 = dOut^T h2, db3 = sum_rows dOut: thread = column, rows
// streamed from HBM (h2: 1 KiB per row; dOut through the scalar cache).  Workgroup
// b covers the same 128-row tiles as workgroup b of the kernel above and writes
// the remaining segments of the same partial row.
template <int NOUT>
__global__ __launch_bounds__(kBlock) void mlp_head_grads_kernel(
    const float *__restrict__ h2, const float *__restrict__ dout, int64_t m, const float *__restrict__ w3,
    int n_out_rt, int d_in, float *__restrict__ partials, int partial_stride) {
  constexpr int kOut = NOUT > 0 ? pad_out(NOUT) : kMaxOut;
  const int n_out = NOUT > 0 ? NOUT : n_out_rt;
  const int tid = threadIdx.x;
  float w3r[kOut];
#pragma unroll
  for (int q = 0; q < kOut; ++q) w3r[q] = q < n_out ? w3[q * kHidden + tid] : 0.0f;
  float db2[2] = {0.0f, 0.0f}, dw3[kOut][2], db3[kOut];
#pragma unroll
  for (int q = 0; q < kOut; ++q) dw3[q][0] = dw3[q][1] = db3[q] = 0.0f;
  constexpr int kBatch = 16;
  const int64_t tiles = (m + kSplitRows - 1) / kSplitRows;
  for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const int64_t r0 = tile * kSplitRows;
    const int rows = (int)((m - r0) < kSplitRows ? (m - r0) : kSplitRows);
    const __amdgpu_buffer_rsrc_t h2rsrc = buffer_rsrc(h2 + r0 * kHidden, rows * kHidden * 4);
    for (int s0 = 0; s0 < rows; s0 += kBatch) {
      float hv[kBatch];
#pragma unroll
      for (int u = 0; u < kBatch; ++u) hv[u] = buffer_load_f32(h2rsrc, tid * 4, (s0 + u) * (kHidden * 4));
#pragma unroll
      for (int u = 0; u < kBatch; ++u) {
        const int s = s0 + u;
        const bool valid = s < rows;
        const int64_t srow = r0 + (valid ? s : rows - 1);
        float g = 0.0f;
#pragma unroll
        for (int q = 0; q < kOut; ++q) {
          if (NOUT > 0 ? q < NOUT : q < n_out) {
            const float d = valid ? dout[srow * n_out + q] : 0.0f;
            g = __builtin_fmaf(d, w3r[q], g);
            dw3[q][u & 1] = __builtin_fmaf(d, hv[u], dw3[q][u & 1]);
            db3[q] += d;
          }
        }
        db2[u & 1] += hv[u] > 0.0f ? g : 0.0f;
      }
    }
  }
  float *row = partials + (int64_t)blockIdx.x * partial_stride;
  const int off_db2 = kHidden * d_in + kHidden, off_dw3 = off_db2 + kHidden, off_db3 = off_dw3 + n_out * kHidden;
  row[off_db2 + tid] = db2[0] + db2[1];
#pragma unroll
  for (int q = 0; q < kOut; ++q)
    if (q < n_out) {
      row[off_dw3 + q * kHidden + tid] = dw3[q][0] + dw3[q][1];
      if (tid == q) row[off_db3 + q] = db3[q];
    }
}

template <int DIN, int NOUT>
static int launch_backward_split(int grid, hipStream_t s, const float *x, const float *w1, const float *b1,
                                 const float *h2, const float *dout, int64_t m, int d_in, const void *w2ts,
                                 const float *w3, int n_out, float *dz2_out, float *partials, int stride,
                                 int head_rows = -1, const uint32_t *gate2 = nullptr) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&mlp_tower_backward_split_kernel<DIN, NOUT>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipGetLastError();
    attr_set = true;
  }
  constexpr int kIn = DIN > 0 ? DIN : kMaxIn;
  constexpr int kGateBlock = (kIn >= 3 && DIN > 0) ? 0 : kSplitRows * 32;  // (wide inputs: gate words through registers)
  static_assert(DIN == 0 || NOUT == 0 || 2 * kSplitStageBytes + kHidden * (1 + kIn) * 4 + kGateBlock <= 80 * 1024,
                "two workgroups per CU");
  mlp_tower_backward_split_kernel<DIN, NOUT><<<grid, kBlock, 2 * kSplitStageBytes + kHidden * (1 + kIn) * 4 + kGateBlock, s>>>(
      x, w1, b1, h2, dout, m, d_in, w2ts, w3, n_out, dz2_out, partials, stride, head_rows, gate2);
  const int status = launch_status();
  if (status != 0 || head_rows >= 0) return status;  // (fused: the weight-gradient kernel forms the head gradients)
  mlp_head_grads_kernel<NOUT><<<grid, kBlock, 0, s>>>(h2, dout, m, w3, n_out, d_in, partials, stride);
  return launch_status();
}

template <int DIN>
static int dispatch_backward_split_nout(int n_out, int grid, hipStream_t s, const float *x, const float *w1,
                                        const float *b1, const float *h2, const float *dout, int64_t m, int d_in,
                                        const void *w2ts, const float *w3, float *dz2_out, float *partials,
                                        int stride, int head_rows, const uint32_t *gate2) {
  switch (n_out) {
    case 1: return launch_backward_split<DIN, 1>(grid, s, x, w1, b1, h2, dout, m, d_in, w2ts, w3, n_out, dz2_out, partials, stride, head_rows, gate2);
    case 2: return launch_backward_split<DIN, 2>(grid, s, x, w1, b1, h2, dout, m, d_in, w2ts, w3, n_out, dz2_out, partials, stride, head_rows, gate2);
    case 3: return launch_backward_split<DIN, 3>(grid, s, x, w1, b1, h2, dout, m, d_in, w2ts, w3, n_out, dz2_out, partials, stride, head_rows, gate2);
    default: return RL8_ESIZE;
  }
}

// ---- weight gradient of the 256x256 layer on the same scheme -------------------
//   dW2[j][i] = sum over samples s of dZ2[s][j] * h1[s][i]
// The reduction runs over SAMPLES, so an MFMA operand fragment is "eight
// consecutive samples of one column" -- and with thread = column that is exactly
// what a thread collects from row-major data with coalesced loads: no transpose
// anywhere.  One workgroup of 8 waves per CU owns the whole 256x256 output (wave =
// 2 j-tiles x 4 i-tiles = 128 accumulator registers) and a strided share of the
// 16-sample chunks; per chunk each thread
//   * takes eight samples of column j of dZ2 (HBM, prefetched two chunks ahead),
//   * RECOMPUTES eight samples of column i of h1 = relu(b1[i] + x . w1[i]) -- layer 1
//     is a handful of fmas per element, the observations come through the scalar
//     cache -- instead of reading 1 KiB of h1 per sample back from HBM,
//   splits both into the three planes and writes them to LDS as ready fragments,
// all on the VALU beside the bf16 MFMAs of the previous chunk.  The workgroup has the CU
// to itself, so nothing else fills its gaps and the step is ordered to have none: the
// wave-uniform operands (observations, dOut) arrive in scalar registers a step ahead,
// the one barrier per step sits in front of the step's LAST group of products and the
// next step's first fragments are fetched right behind it (see do_step).  Partial sums leave
// as one slab per workgroup, added up in slab order by mlp_wgrad_split_reduce_kernel
// (bitwise reproducible; no atomics).
// A launch covers at most this many samples; longer inputs are summed segment by segment
// (dW2 through the slab reduction's accumulate mode, the head-gradient partial rows in
// place).  Every accumulator is an fp32 chain over its workgroup's share of the samples:
// at 2^25 rows in one launch dW2 was 7e-5 (of its largest entry) off an fp64 evaluation,
// at 2^23 per launch 4e-5 -- the length the parity tests were validated at, kept
// whatever the caller's pass size.
constexpr int64_t kWgradSegmentRows = (int64_t)1 << 23;
constexpr int kWsThreads = 512;
constexpr int kWsChunk = 16;                             // samples per k-step
constexpr int kWsOperandBytes = 3 * 2 * kHidden * 16;    // [plane][sample half][column] x 16 B
constexpr int kWsPlane = 2 * kHidden * 16;
constexpr int kWsStageBytes = 2 * kWsOperandBytes;       // dZ2^T | h1

// FUSED (= n_out) > 0: the first operand is not read but formed on the spot,
//   dZ2 = (dOut x W3) * (h2 > 0)  from h2 (`dz2` then points at h2), dOut and W3 --
// thread = column, so W3's column is per-thread constants and dOut comes through
// the scalar cache -- and, since h2, dOut and dZ2 are all in hand column-wise, the
// head gradients db2 = sum dZ2 and dW3 = dOut^T h2 are accumulated too (per thread;
// one partial row per workgroup; db3 = sum dOut is left to the caller: it is a
// column sum of a [M][n_out] array and would cost this kernel 4 VALU per sample).  The data-gradient kernel then
// neither stores dZ2 nor needs a separate head-gradient pass.
struct WgradFusedArgs {
  const float *dout, *w3;
  float *partials;     // rows of `partial_stride` floats: [dW1 | db1 | db2 | dW3 | db3]
  int partial_stride;
  int other_rows;      // rows whose [dW1 | db1] segment the data-gradient kernel fills
  int accumulate;      // a later segment of the same rows-of-samples sum: add to the partial rows
  // gate-plane kernel in BITS mode only: the gate bits of h2 ([m][8] words) instead of h2 itself, and b2
  const uint32_t *gate2 = nullptr;
  const float *b2 = nullptr;
  // ... on fp16 planes (F16): max |dOut column q| at [q], max |x column c| at [4 + c], over the rows of the whole call
  // (wgrad_bounds_kernel), as the bit patterns of non-negative floats
  const uint32_t *bounds = nullptr;
};

// LOADH: both operands come from memory -- dZ rows at pitch `ops.dz_pitch`, h1 rows at
// ops.h + row * ops.h_pitch -- instead of h1 being recomputed from the observations:
// the recurrent weight gradient of the LSTM, dW_hh[q] = dG_q^T h_{t-1}, with dG_q one gate
// of [M][4][256] rows (pitch 1024) and h_{t-1} [M][256].  Same loop; the producer splits a
// second prefetched register set instead of running layer 1.
struct WgradOperands {
  const float *h;
  int dz_pitch, h_pitch;  // floats
  // Optional column sums over the rows of this launch (the LSTM's input-weight and bias
  // gradients, from the dZ values the producers hold anyway): one partial row per
  // workgroup, [256 x DIN (sum of dZ[row][col] * x[row][i]) | 256 (sum of dZ[row][col])],
  // x dense [m][DIN] in the kernel's `x` argument.  NULL: none.
  float *colsums;
  int colsum_accumulate;  // a later segment: add to the rows the first one wrote
  // F16: the operands as two fp16 planes each, times a power of two from *dz_bound / *h_bound (device words: bit
  // patterns of bounds on |dZ| / |h| over everything this call reads, e.g. rl8_lstm_rows_backward_f32's dg_bound_out)
  const uint32_t *dz_bound = nullptr, *h_bound = nullptr;
};

// F16 (round 3, fused mode with compiled widths): BOTH operands as two fp16 planes, each scaled by a power of two per
// column of the OUTPUT it indexes -- dZ2[s][j] by 2^a(j) from sum_q max|dOut_q| |W3[q][j]|, h1[s][i] by 2^b(i) from
// |b1[i]| + sum_c max|x_c| |w1[i][c]| -- so the factors leave the sum over samples; THREE plane products per 16 samples
// instead of six (see mlp_wgrad_gate_kernel's F16 for the accuracy argument; here both operands carry 22 bits).
template <int DIN, int FUSED = 0, bool LOADH = false, bool F16 = false>
__global__ __launch_bounds__(kWsThreads, 1) void mlp_wgrad_split_kernel(
    const float *__restrict__ dz2, const float *__restrict__ x, const float *__restrict__ w1,
    const float *__restrict__ b1, int64_t m, int d_in_rt, float *__restrict__ slabs, WgradFusedArgs fused,
    WgradOperands ops) {
  static_assert(!LOADH || (DIN > 0 && FUSED == 0), "the two-operand mode: compiled input widths, no head fusion");
  static_assert(!F16 || (DIN > 0 && (FUSED > 0 || LOADH)), "fp16 planes: the fused mode of compiled widths, or both operands loaded");
  constexpr int kIn = DIN > 0 ? DIN : kMaxIn;
  constexpr int kOut = FUSED > 0 ? FUSED : 1;
  const int d_in = DIN > 0 ? DIN : d_in_rt;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const unsigned lds0 = lds_offset(smem);
  const int tid = threadIdx.x, lane = tid & 63, l32 = lane & 31, hh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wj = wave >> 1, wi = wave & 1;  // j-tiles {2wj, 2wj+1}, i-tiles {4wi .. 4wi+3}
  const int col = tid & 255;                // producer: column (j of dZ2 and i of h1) ...
  const int kh = wave >> 2;                 // ... and which eight samples of the chunk (wave-uniform)

  const unsigned a_read = lds0 + (hh * kHidden + 64 * wj + l32) * 16;
  const unsigned b_read = lds0 + kWsOperandBytes + (hh * kHidden + 128 * wi + l32) * 16;
  const unsigned p_write = lds0 + (kh * kHidden + col) * 16;

  float w1r[kIn];
#pragma unroll
  for (int c = 0; c < kIn; ++c) w1r[c] = LOADH ? 0.0f : (DIN > 0 || c < d_in) ? w1[col * d_in + c] : 0.0f;
  const float b1r = LOADH ? 0.0f : b1[col];
  const int dz_pitch = LOADH ? ops.dz_pitch : kHidden;
  [[maybe_unused]] float w3r[kOut], db2a = 0.0f, dw3a[kOut];
  if constexpr (FUSED > 0) {
#pragma unroll
    for (int q = 0; q < kOut; ++q) {
      w3r[q] = fused.w3[q * kHidden + col];
      dw3a[q] = 0.0f;
    }
  }
  [[maybe_unused]] float scale_a = 1.0f, scale_b = 1.0f;  // F16: this thread's column as j of dZ2 and as i of h1
  [[maybe_unused]] float *inv_a = reinterpret_cast<float *>(smem + 2 * kWsStageBytes), *inv_b = inv_a + kHidden;
  if constexpr (F16) {
    float za = 0.0f, hb = __builtin_fabsf(b1r);
    if constexpr (LOADH) {  // one power of two per operand for the launch
      za = __uint_as_float(*ops.dz_bound);
      hb = __uint_as_float(*ops.h_bound);
    } else {
#pragma unroll
      for (int q = 0; q < kOut; ++q) za = __builtin_fmaf(__uint_as_float(fused.bounds[q]), __builtin_fabsf(w3r[q]), za);
#pragma unroll
      for (int c = 0; c < kIn; ++c) hb = __builtin_fmaf(__uint_as_float(fused.bounds[4 + c]), __builtin_fabsf(w1r[c]), hb);
    }
    const int ea = f16_bound_exponent(za * 1.0001f), eb = f16_bound_exponent(hb * 1.0001f);
    scale_a = __builtin_amdgcn_ldexpf(1.0f, kF16Top - ea);
    scale_b = __builtin_amdgcn_ldexpf(1.0f, kF16Top - eb);
    if (kh == 0) {
      inv_a[col] = __builtin_amdgcn_ldexpf(1.0f, ea - kF16Top);
      inv_b[col] = __builtin_amdgcn_ldexpf(1.0f, eb - kF16Top);
    }
  }

  const int64_t chunks = (m + kWsChunk - 1) / kWsChunk;
  const int64_t stride = gridDim.x;
  const int64_t mine = (chunks - blockIdx.x + stride - 1) / stride;  // >= 1 (grid <= chunks)

  // dZ2 of chunk number n of this workgroup: this thread's column, its eight samples.
  // Chunks past the end (and samples past m) read as zero through the descriptor.
  auto load_dz = [&](float (&dst)[8], int64_t n) {
    const int64_t chunk = blockIdx.x + n * stride;
    const int64_t left = m - chunk * kWsChunk;
    const int rows = left <= 0 ? 0 : left < kWsChunk ? (int)left : kWsChunk;
    const __amdgpu_buffer_rsrc_t rsrc = buffer_rsrc(rows > 0 ? dz2 + chunk * kWsChunk * dz_pitch : dz2,
                                                    rows > 0 ? ((rows - 1) * dz_pitch + kHidden) * 4 : 0);
#pragma unroll
    for (int e = 0; e < 8; ++e) dst[e] = buffer_load_f32(rsrc, col * 4, (8 * kh + e) * (dz_pitch * 4));
  };
  [[maybe_unused]] auto load_h = [&](float (&dst)[8], int64_t n) {  // LOADH: the same for the h1 operand
    const int64_t chunk = blockIdx.x + n * stride;
    const int64_t left = m - chunk * kWsChunk;
    const int rows = left <= 0 ? 0 : left < kWsChunk ? (int)left : kWsChunk;
    const __amdgpu_buffer_rsrc_t rsrc = buffer_rsrc(rows > 0 ? ops.h + chunk * kWsChunk * ops.h_pitch : ops.h,
                                                    rows > 0 ? ((rows - 1) * ops.h_pitch + kHidden) * 4 : 0);
#pragma unroll
    for (int e = 0; e < 8; ++e) dst[e] = buffer_load_f32(rsrc, col * 4, (8 * kh + e) * (ops.h_pitch * 4));
  };
  auto split8 = [&](const float (&v)[8], u32x4 (&planes)[3]) {
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
      uint32_t hi, mid, lo;
      split_pair(v[e], v[e + 1], hi, mid, lo);
      planes[0][e >> 1] = hi;
      planes[1][e >> 1] = mid;
      planes[2][e >> 1] = lo;
    }
  };
  [[maybe_unused]] auto split8h = [&](const float (&v)[8], float scale, u32x4 (&planes)[3]) {  // F16: hi, lo (planes[2] unused)
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
      uint32_t hi, lo;
      f16_pair_scaled(v[e], v[e + 1], scale, hi, lo);
      planes[0][e >> 1] = hi;
      planes[1][e >> 1] = lo;
    }
  };
  auto lds_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
  // Observations and dOut of the eight samples a wave produces are wave-uniform: with
  // DIN known they sit in scalar registers, fetched (kIn + kOut loads of eight floats)
  // in front of the PREVIOUS step's barrier, so that inside a step nothing but
  // ds_reads is in flight on lgkmcnt and the first MFMAs can go as soon as THEIR
  // fragments are in (wait_lds<N>), not after all twelve.
  constexpr bool kScalars = DIN > 0;  // (two-operand mode: the observations feed the column sums only)
  constexpr int kXq = kScalars ? kIn : 1, kDq = (kScalars && FUSED > 0) ? kOut : 1;
  [[maybe_unused]] f32x8 xq[kXq], dq[kDq];
  auto row0_of = [&](int64_t n) { return (blockIdx.x + n * stride) * kWsChunk + 8 * kh; };
  // ... of chunk n of this workgroup: requested through buffer descriptors that end at
  // sample m, so samples past the end read as zero (as their h2 / dZ2 do: whatever
  // they contribute is multiplied by zero) and no step needs a slow path.  No wait
  // here: the step barrier's lgkmcnt(0) is the wait, scalars_landed() the fence.
  auto request_scalars = [&](int64_t n) {
    if constexpr (kScalars) {
      const int64_t row0 = row0_of(n);
      const int64_t left = m - row0;
      // (two-operand mode without column sums passes no observations: descriptors of size 0)
      const int rows = (left <= 0 || x == nullptr) ? 0 : left < 8 ? (int)left : 8;
      const u32x4 rx = scalar_rsrc(x != nullptr ? x + row0 * kIn : reinterpret_cast<const float *>(slabs), rows * kIn * 4);
      xq[0] = scalar_buffer_load_x8<0>(rx);
      if constexpr (kIn > 1) xq[1] = scalar_buffer_load_x8<32>(rx);
      if constexpr (kIn > 2) xq[2] = scalar_buffer_load_x8<64>(rx);
      if constexpr (kIn > 3) xq[3] = scalar_buffer_load_x8<96>(rx);
      if constexpr (kIn > 4) xq[4] = scalar_buffer_load_x8<128>(rx);
      if constexpr (FUSED > 0) {
        const u32x4 rd = scalar_rsrc(fused.dout + row0 * kOut, rows * kOut * 4);
        dq[0] = scalar_buffer_load_x8<0>(rd);
        if constexpr (kOut > 1) dq[1] = scalar_buffer_load_x8<32>(rd);
        if constexpr (kOut > 2) dq[2] = scalar_buffer_load_x8<64>(rd);
      }
    }
  };
  auto scalars_landed = [&]() {  // directly behind a barrier / lgkmcnt(0)
    if constexpr (kScalars) {
#pragma unroll
      for (int i = 0; i < kXq; ++i) scalar_tie(xq[i]);
      if constexpr (FUSED > 0) {
#pragma unroll
        for (int i = 0; i < kDq; ++i) scalar_tie(dq[i]);
      }
    }
  };
  auto produce = [&](const float (&dzv)[8], [[maybe_unused]] const float (&hv)[8], int64_t n, u32x4 (&pa)[3],
                     u32x4 (&pb)[3]) {
    if constexpr (LOADH) {
      if constexpr (F16) {
        split8h(dzv, scale_a, pa);
        split8h(hv, scale_b, pb);
      } else {
        split8(dzv, pa);
        split8(hv, pb);
      }
      return;
    }
    // (runtime d_in only: per-row loads, rows past the end clamped as above)
    auto row_of = [&](int e) {
      const int64_t row = row0_of(n) + e;
      return row < m ? row : m - 1;
    };
    if constexpr (FUSED > 0) {
      float dz[8];  // dzv holds h2
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float g = 0.0f;
#pragma unroll
        for (int q = 0; q < kOut; ++q) {
          float d;
          if constexpr (kScalars) d = dq[(e * kOut + q) >> 3][(e * kOut + q) & 7];
          else d = fused.dout[row_of(e) * kOut + q];
          g = __builtin_fmaf(d, w3r[q], g);
          dw3a[q] = __builtin_fmaf(d, dzv[e], dw3a[q]);
        }
        dz[e] = dzv[e] > 0.0f ? g : 0.0f;
        db2a += dz[e];
      }
      if constexpr (F16) split8h(dz, scale_a, pa);
      else split8(dz, pa);
    } else {
      split8(dzv, pa);
    }
    float h[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float v = b1r;
#pragma unroll
      for (int c = 0; c < kIn; ++c) {
        if constexpr (kScalars) v = __builtin_fmaf(xq[(e * kIn + c) >> 3][(e * kIn + c) & 7], w1r[c], v);
        else if (c < d_in) v = __builtin_fmaf(x[row_of(e) * d_in + c], w1r[c], v);
      }
      h[e] = relu1(v);
    }
    if constexpr (F16) split8h(h, scale_b, pb);
    else split8(h, pb);
  };
  auto write_planes = [&](int stage, const u32x4 (&pa)[3], const u32x4 (&pb)[3]) {
    const unsigned addr = p_write + stage * kWsStageBytes;
    lds_write_b128<0>(addr, pa[0]);
    lds_write_b128<kWsPlane>(addr, pa[1]);
    if constexpr (!F16) lds_write_b128<2 * kWsPlane>(addr, pa[2]);
    lds_write_b128<kWsOperandBytes>(addr, pb[0]);
    lds_write_b128<kWsOperandBytes + kWsPlane>(addr, pb[1]);
    if constexpr (!F16) lds_write_b128<kWsOperandBytes + 2 * kWsPlane>(addr, pb[2]);
  };

  f32x16 acc[2][4];
  float dzq[2][8];
  [[maybe_unused]] float hq[1][8];
  [[maybe_unused]] float cs_b = 0.0f, cs_w[kIn];  // LOADH column sums of this thread's column and sample half
#pragma unroll
  for (int c = 0; c < kIn; ++c) cs_w[c] = 0.0f;
  const bool want_colsums = LOADH && ops.colsums != nullptr;

  // Chunk n (stage n & 1) is consumed while chunk n+1 is produced from dzq[(n+1) & 1]
  // and the dZ2 of chunk n+2 is requested into dzq[n & 1].
  using T = std::true_type;
  using F = std::false_type;
  using P0 = std::integral_constant<int, 0>;
  using P1 = std::integral_constant<int, 1>;

  // One barrier per step, placed in FRONT of the step's last group of products: behind
  // it the fragments the next step opens with (both A and four B of the mid planes)
  // are fetched into the registers the second-to-last group has just released, and
  // land while the last group runs -- a step opens with eight products, not with an
  // LDS round trip.  The B registers trade roles each step for that: BM (mid, then
  // lo planes) is f.bm in even steps and f.bh in odd ones, BH (hi planes) the other.
  SplitFrags f;
  auto first_reads = [&](auto parity_tag) {  // ... of the chunk in stage P, into am and that step's BM
    constexpr int P = decltype(parity_tag)::value;
    if constexpr (F16) return;  // (the fp16 step fetches its twelve fragments itself, under its own production)
    const unsigned ar = a_read + P * kWsStageBytes, br = b_read + P * kWsStageBytes;
    u32x4(&BM)[4] = *(P == 0 ? &f.bm : &f.bh);
    f.am[0] = lds_read_b128<kWsPlane>(ar);
    BM[0] = lds_read_b128<kWsPlane>(br);
    BM[1] = lds_read_b128<kWsPlane + 512>(br);
    BM[2] = lds_read_b128<kWsPlane + 1024>(br);
    BM[3] = lds_read_b128<kWsPlane + 1536>(br);
    f.am[1] = lds_read_b128<kWsPlane + 512>(ar);
  };
  auto do_step = [&](auto first_tag, auto parity_tag, int64_t n) {
    constexpr bool FIRST = decltype(first_tag)::value;
    constexpr int P = decltype(parity_tag)::value;
    const unsigned ar = a_read + P * kWsStageBytes, br = b_read + P * kWsStageBytes;
    if constexpr (F16) {
      // ah / am: hi / lo planes of dZ2^T, bh / bm: of h1.  The first product's six fragments are requested in front of
      // the next chunk's production (which does not depend on them), the other six behind it: they land under the
      // first product.  (All twelve in front: the wider variants spilled.)
      f.am[0] = lds_read_b128<kWsPlane>(ar);
      f.am[1] = lds_read_b128<kWsPlane + 512>(ar);
      f.bh[0] = lds_read_b128<0>(br);
      f.bh[1] = lds_read_b128<512>(br);
      f.bh[2] = lds_read_b128<1024>(br);
      f.bh[3] = lds_read_b128<1536>(br);
      if constexpr (LOADH) {
        // one register set per operand (as in the bf16 step below): split, write, re-request
        const unsigned addr = p_write + (P ^ 1) * kWsStageBytes;
        if (want_colsums) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            cs_b += dzq[0][e];
#pragma unroll
            for (int c = 0; c < kIn; ++c) cs_w[c] = __builtin_fmaf(dzq[0][e], xq[(e * kIn + c) >> 3][(e * kIn + c) & 7], cs_w[c]);
          }
        }
        u32x4 pl[3];
        split8h(dzq[0], scale_a, pl);
        lds_write_b128<0>(addr, pl[0]);
        lds_write_b128<kWsPlane>(addr, pl[1]);
        load_dz(dzq[0], n + 2);
        split8h(hq[0], scale_b, pl);
        lds_write_b128<kWsOperandBytes>(addr, pl[0]);
        lds_write_b128<kWsOperandBytes + kWsPlane>(addr, pl[1]);
        load_h(hq[0], n + 2);
      } else {
        // (the other stage's last readers passed the previous step's barrier: its planes can go out as soon as they exist)
        u32x4 pa[3], pb[3];
        load_dz(dzq[P], n + 2);
        produce(dzq[P ^ 1], dzq[P ^ 1], n + 1, pa, pb);
        write_planes(P ^ 1, pa, pb);
      }
      f.ah[0] = lds_read_b128<0>(ar);
      f.ah[1] = lds_read_b128<512>(ar);
      f.bm[0] = lds_read_b128<kWsPlane>(br);
      f.bm[1] = lds_read_b128<kWsPlane + 512>(br);
      f.bm[2] = lds_read_b128<kWsPlane + 1024>(br);
      f.bm[3] = lds_read_b128<kWsPlane + 1536>(br);
      wait_lds<6>(f.am[0], f.am[1], f.bh[0], f.bh[1], f.bh[2], f.bh[3]);  // only LDS operations in flight: in-order count
      __builtin_amdgcn_sched_barrier(0);
      f16_mma<FIRST>(f.am, f.bh, acc);  // lo x hi
      request_scalars(n + 2);
      wait_lds<0>(f.ah[0], f.ah[1], f.bm[0], f.bm[1], f.bm[2], f.bm[3]);
      __builtin_amdgcn_sched_barrier(0);
      f16_mma<false>(f.ah, f.bm, acc);  // hi x lo
      f16_mma<false>(f.ah, f.bh, acc);  // hi x hi
      __builtin_amdgcn_sched_barrier(0);
      lds_barrier();
      scalars_landed();
      return;
    }
    u32x4(&BM)[4] = *(P == 0 ? &f.bm : &f.bh);
    u32x4(&BH)[4] = *(P == 0 ? &f.bh : &f.bm);
    // am and BM are in (previous step / prologue); the hi planes:
    f.ah[0] = lds_read_b128<0>(ar);
    f.ah[1] = lds_read_b128<512>(ar);
    BH[0] = lds_read_b128<0>(br);
    BH[1] = lds_read_b128<512>(br);
    BH[2] = lds_read_b128<1024>(br);
    BH[3] = lds_read_b128<1536>(br);
    __builtin_amdgcn_sched_barrier(0);
    split_mma_row<FIRST>(f.am[0], BM, acc[0]);  // at once: nothing in front of the first products
    __builtin_amdgcn_sched_barrier(0);
    split_mma_row<FIRST>(f.am[1], BM, acc[1]);
    u32x4 pa[3], pb[3];
    if constexpr (LOADH) {
      // One register set per operand and one set of planes: the chunk requested a step
      // ago is split and written operand by operand (the other stage's last readers
      // passed the previous step's barrier), then the set is re-requested for the chunk
      // after it.  (Two sets each, planes of both operands held to the usual place
      // behind the second product group: 576 B of scratch -- whole accumulator tuples.)
      const unsigned addr = p_write + (P ^ 1) * kWsStageBytes;
      if (want_colsums) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          cs_b += dzq[0][e];
#pragma unroll
          for (int c = 0; c < kIn; ++c) cs_w[c] = __builtin_fmaf(dzq[0][e], xq[(e * kIn + c) >> 3][(e * kIn + c) & 7], cs_w[c]);
        }
      }
      split8(dzq[0], pa);
      lds_write_b128<0>(addr, pa[0]);
      lds_write_b128<kWsPlane>(addr, pa[1]);
      lds_write_b128<2 * kWsPlane>(addr, pa[2]);
      load_dz(dzq[0], n + 2);
      split8(hq[0], pa);
      lds_write_b128<kWsOperandBytes>(addr, pa[0]);
      lds_write_b128<kWsOperandBytes + kWsPlane>(addr, pa[1]);
      lds_write_b128<kWsOperandBytes + 2 * kWsPlane>(addr, pa[2]);
      load_h(hq[0], n + 2);
    } else {
      load_dz(dzq[P], n + 2);
      produce(dzq[P ^ 1], dzq[P ^ 1], n + 1, pa, pb);
    }
    if constexpr (kScalars) {
      wait_lds<4>(f.ah[0], f.ah[1]);  // only ds_reads are in flight: in-order count
      split_mma<false>(f.ah, BM, acc);
      wait_lds<0>(BH[0], BH[1], BH[2], BH[3]);
    } else {
      wait_lds_all(f);
      split_mma<false>(f.ah, BM, acc);
    }
    split_mma<false>(f.am, BH, acc);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (!LOADH) write_planes(P ^ 1, pa, pb);
    f.am[0] = lds_read_b128<2 * kWsPlane>(ar);
    f.am[1] = lds_read_b128<2 * kWsPlane + 512>(ar);
    BM[0] = lds_read_b128<2 * kWsPlane>(br);
    BM[1] = lds_read_b128<2 * kWsPlane + 512>(br);
    BM[2] = lds_read_b128<2 * kWsPlane + 1024>(br);
    BM[3] = lds_read_b128<2 * kWsPlane + 1536>(br);
    request_scalars(n + 2);  // for the chunk the NEXT step produces
    __builtin_amdgcn_sched_barrier(0);
    split_mma<false>(f.ah, BH, acc);
    __builtin_amdgcn_sched_barrier(0);
    wait_lds_all(f);
    split_mma<false>(f.am, BH, acc);  // last use of am and BH in this step
    __builtin_amdgcn_sched_barrier(0);
    lds_barrier();
    scalars_landed();
    if constexpr (P == 0) first_reads(P1{});
    else first_reads(P0{});
    __builtin_amdgcn_sched_barrier(0);
    split_mma<false>(f.ah, BM, acc);
    __builtin_amdgcn_sched_barrier(0);
    // landed before anything can copy or carry these registers (loop back-edge)
    wait_lds<0>(f.am[0], f.am[1], BH[0], BH[1], BH[2], BH[3]);
  };

  {
    load_dz(dzq[0], 0);
    if constexpr (LOADH) load_h(hq[0], 0);
    else load_dz(dzq[1], 1);
    request_scalars(0);
    lds_barrier();
    scalars_landed();
    u32x4 pa[3], pb[3];
    if constexpr (LOADH) {
      if (want_colsums) {  // chunk 0 (its observations were requested above)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          cs_b += dzq[0][e];
#pragma unroll
          for (int c = 0; c < kIn; ++c) cs_w[c] = __builtin_fmaf(dzq[0][e], xq[(e * kIn + c) >> 3][(e * kIn + c) & 7], cs_w[c]);
        }
      }
    }
    produce(dzq[0], hq[0], 0, pa, pb);
    if constexpr (LOADH) {
      load_dz(dzq[0], 1);
      load_h(hq[0], 1);
    }
    write_planes(0, pa, pb);
    request_scalars(1);
    lds_barrier();
    scalars_landed();
    first_reads(P0{});
    wait_lds<0>(f.am[0], f.am[1], f.bm[0], f.bm[1], f.bm[2], f.bm[3]);
  }
  do_step(T{}, P0{}, 0);
  int64_t n = 1;
  if constexpr (LOADH) {
    // an odd number of steps, the last one on an all-zero chunk past the end if need be
    // (descriptors of size 0): with a remainder step behind the loop the optimizer merged
    // it into the loop body, and the accumulators went through copies (524 B of scratch)
    const int64_t steps = mine | 1;
#pragma unroll 1
    for (; n < steps; n += 2) {
      do_step(F{}, P1{}, n);
      do_step(F{}, P0{}, n + 1);
    }
  } else {
#pragma unroll 1
    for (; n + 1 < mine; n += 2) {
      do_step(F{}, P1{}, n);
      do_step(F{}, P0{}, n + 1);
    }
    if (n < mine) do_step(F{}, P1{}, n);
  }

  float *slab = slabs + (int64_t)blockIdx.x * kHidden * kHidden;
#pragma unroll
  for (int ja = 0; ja < 2; ++ja)
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int j = 64 * wj + 32 * ja + (r & 3) + 8 * (r >> 2) + 4 * hh;
        const int i = 128 * wi + 32 * t + l32;
        slab[j * kHidden + i] = F16 ? acc[ja][t][r] * (inv_a[j] * inv_b[i]) : acc[ja][t][r];
      }

  if constexpr (LOADH) {
    if (want_colsums) {  // fold the two sample halves of a column through LDS, fixed order
      float *red = reinterpret_cast<float *>(smem);  // [256][1 + kIn]
      __syncthreads();
      if (kh == 1) {
        red[col * (1 + kIn)] = cs_b;
#pragma unroll
        for (int c = 0; c < kIn; ++c) red[col * (1 + kIn) + 1 + c] = cs_w[c];
      }
      __syncthreads();
      if (kh == 0) {
        float *row = ops.colsums + (int64_t)blockIdx.x * (kHidden * (kIn + 1));
        const bool more = ops.colsum_accumulate != 0;
        const float b = cs_b + red[col * (1 + kIn)];
        row[kHidden * kIn + col] = more ? row[kHidden * kIn + col] + b : b;
#pragma unroll
        for (int c = 0; c < kIn; ++c) {
          const float w = cs_w[c] + red[col * (1 + kIn) + 1 + c];
          row[col * kIn + c] = more ? row[col * kIn + c] + w : w;
        }
      }
    }
  }
  if constexpr (FUSED > 0) {
    // Head gradients: fold the two sample halves (kh) of a column through LDS in a
    // fixed order and write this workgroup's partial row; zero the other kernel's
    // segment where it has no row of its own.
    float *red = reinterpret_cast<float *>(smem);  // [256][1 + kOut] + [kOut]
    __syncthreads();
    if (kh == 1) {
      red[col * (1 + kOut)] = db2a;
#pragma unroll
      for (int q = 0; q < kOut; ++q) red[col * (1 + kOut) + 1 + q] = dw3a[q];
    }
    __syncthreads();
    float *row = fused.partials + (int64_t)blockIdx.x * fused.partial_stride;
    const int off_db2 = kHidden * d_in + kHidden, off_dw3 = off_db2 + kHidden, off_db3 = off_dw3 + kOut * kHidden;
    const bool more = fused.accumulate != 0;  // (a later segment: the first one wrote / zeroed the row)
    if (kh == 0) {
      const float sum_b2 = db2a + red[col * (1 + kOut)];
      row[off_db2 + col] = more ? row[off_db2 + col] + sum_b2 : sum_b2;
#pragma unroll
      for (int q = 0; q < kOut; ++q) {
        const float sum_w3 = dw3a[q] + red[col * (1 + kOut) + 1 + q];
        row[off_dw3 + q * kHidden + col] = more ? row[off_dw3 + q * kHidden + col] + sum_w3 : sum_w3;
      }
      // db3 = sum of dOut needs no matrix kernel: the caller forms it (the segment is zeroed)
      if (col < kOut && !more) row[off_db3 + col] = 0.0f;
    }
    if (!more && (int)blockIdx.x >= fused.other_rows)
      for (int idx = tid; idx < kHidden * d_in + kHidden; idx += kWsThreads) row[idx] = 0.0f;
  }
}

// The fused weight gradient of a tower with ONE output (value towers): there
//   dZ2[s][j] = G[s][j] * dOut[s] * W3[j],   G = (h2 > 0) in {0, 1},
// so   dW2[j][i] = W3[j] * sum_s G[s][j] * (dOut[s] * h1[s][i]):
// the first operand is the gate itself -- ONE bf16 plane, exact -- and the second,
// dOut[s] * h1[s][i], takes the three planes: THREE exact plane products per 16 samples
// instead of six (and one operand less to split on the VALU); W3[j] multiplies the finished
// sums.  Same loop shape as mlp_wgrad_split_kernel: thread = column, 16-sample chunks, one
// barrier per step in front of the step's last group of products, the next step's first
// fragments fetched behind it.  Head gradients on the way, as there:
//   db2[j] = W3[j] * sum_s G[s][j] dOut[s],   dW3[j] = sum_s dOut[s] h2[s][j].
// PAIR: the same for a head of TWO outputs whose gradients are exact negatives of each other,
// dOut[s][1] == -dOut[s][0] (a two-way categorical: the loss kernel makes it exact; the host
// checks it on the data before choosing this kernel): dZ2 = G * dOut[s][0] * (W3[0] - W3[1]),
// and dW3[1] = -dW3[0].  dOut is then [m][2] and column 0 is taken.
// BITS: the gate comes as the forward's gate BITS (32 bytes per row) and h2 is not read at all -- nor was it
// stored: the only other use of h2 here, dW3[j] = sum_s dOut[s] h2[s][j], follows from the sums this kernel
// forms anyway.  With h2 = G * (h1 W2^T + b2):
//   dW3[j] = sum_i W2[j][i] * M[j][i] + b2[j] * sum_s G[s][j] dOut[s],   M[j][i] = sum_s G[s][j] dOut[s] h1[s][i]
// and M is the accumulator (dW2[j][i] = W3[j] M[j][i]).  The slabs then hold M; mlp_wgrad_gate_reduce_kernel
// applies W3 and forms the row dots.  In exact arithmetic the same number; in fp32 it differs from the direct
// sum by the rounding of h2 itself (the forward's 256-term dot products), which is what the reference's own h2
// differs from ours by.
constexpr int kWgOperandA = 2 * kHidden * 16;             // gate plane: [sample half][column] x 16 B
constexpr int kWgStageBytes = kWgOperandA + kWsOperandBytes;  // gate | three planes of dOut * h1

// F16 (round 3, with BITS): the second operand dOut[s] * h1[s][i] as TWO fp16 planes of its value times a power of two
// per COLUMN i -- the reduction runs over samples, so a factor that depends on i alone comes out of the sum -- chosen
// from a bound on the column: max_s |dOut[s]| * (|b1[i]| + sum_c max_s |x[s][c]| |w1[i][c]|), placed below 2^14.  The
// maxima over the call's rows come from one pass over dOut and x in front of the first segment
// (wgrad_gate_bounds_kernel: 0.5 % of the call).  TWO plane products per 16 samples instead of three, four VALU
// instructions per pair to form the planes instead of eleven.  Accuracy: the planes carry 22 bits of every term within
// 2^-17 of its column's bound; smaller terms keep an absolute error of 2^-39 of the bound (fp16's subnormal spacing) --
// of the column's sum that is far below what the fp32 accumulation over 2^23 samples itself leaves.  Against fp64:
// tests/test_mlp_split_gpu.py (same bars as the bf16 planes: error / max |dW2|).
template <int DIN, bool PAIR = false, bool BITS = false, bool F16 = false>
__global__ __launch_bounds__(kWsThreads, 1) void mlp_wgrad_gate_kernel(
    const float *__restrict__ h2, const float *__restrict__ x, const float *__restrict__ w1,
    const float *__restrict__ b1, int64_t m, float *__restrict__ slabs, WgradFusedArgs fused) {
  static_assert(DIN > 0, "compiled input widths only");
  static_assert(!F16 || BITS, "the fp16 planes go with the gate bits");
  constexpr int kIn = DIN, d_in = DIN;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const unsigned lds0 = lds_offset(smem);
  const int tid = threadIdx.x, lane = tid & 63, l32 = lane & 31, hh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wj = wave >> 1, wi = wave & 1;  // j-tiles {2wj, 2wj+1}, i-tiles {4wi .. 4wi+3}
  const int col = tid & 255;                // producer: column (j of the gate and i of h1) ...
  const int kh = wave >> 2;                 // ... and which eight samples of the chunk (wave-uniform)
  [[maybe_unused]] float col_scale = 1.0f;  // F16: this thread's column's power of two
  [[maybe_unused]] float *inv_scales = reinterpret_cast<float *>(smem + 4 * kWgStageBytes);  // F16: [256], for the epilogue

  const unsigned a_read = lds0 + (hh * kHidden + 64 * wj + l32) * 16;
  const unsigned b_read = lds0 + kWgOperandA + (hh * kHidden + 128 * wi + l32) * 16;
  const unsigned p_write = lds0 + (kh * kHidden + col) * 16;

  float w1r[kIn];
#pragma unroll
  for (int c = 0; c < kIn; ++c) w1r[c] = w1[col * d_in + c];
  const float b1r = b1[col];
  float gsum = 0.0f, dw3a = 0.0f;  // sum_s G dOut (db2 / W3) and dW3 of this thread's column and sample half
  if constexpr (F16) {
    float hb = __builtin_fabsf(b1r);
#pragma unroll
    for (int c = 0; c < kIn; ++c) hb = __builtin_fmaf(__uint_as_float(fused.bounds[4 + c]), __builtin_fabsf(w1r[c]), hb);
    // (fp32 rounding of the bound itself: a hair above)
    const int e = f16_bound_exponent(__uint_as_float(fused.bounds[0]) * hb * 1.0001f);
    col_scale = __builtin_amdgcn_ldexpf(1.0f, kF16Top - e);
    if (kh == 0) inv_scales[col] = __builtin_amdgcn_ldexpf(1.0f, e - kF16Top);
  }

  const int64_t chunks = (m + kWsChunk - 1) / kWsChunk;
  const int64_t stride = gridDim.x;
  const int64_t mine = (chunks - blockIdx.x + stride - 1) / stride;  // >= 1 (grid <= chunks)

  // h2 of chunk number n of this workgroup: this thread's column, its eight samples.
  // Chunks past the end (and samples past m) read as zero through the descriptor.
  auto load_h2 = [&](float (&dst)[8], int64_t n) {
    const int64_t chunk = blockIdx.x + n * stride;
    const int64_t left = m - chunk * kWsChunk;
    const int rows = left <= 0 ? 0 : left < kWsChunk ? (int)left : kWsChunk;
    if constexpr (BITS) {  // this column's gate word of each of the eight rows (two distinct words per wave and row)
      const uint32_t *g = fused.gate2;
      const __amdgpu_buffer_rsrc_t rsrc = buffer_rsrc(rows > 0 ? g + chunk * kWsChunk * 8 : g, rows * 32);
#pragma unroll
      for (int e = 0; e < 8; ++e) dst[e] = buffer_load_f32(rsrc, (col >> 5) * 4, (8 * kh + e) * 32);
      return;
    }
    const __amdgpu_buffer_rsrc_t rsrc = buffer_rsrc(rows > 0 ? h2 + chunk * kWsChunk * kHidden : h2, rows * kHidden * 4);
#pragma unroll
    for (int e = 0; e < 8; ++e) dst[e] = buffer_load_f32(rsrc, col * 4, (8 * kh + e) * (kHidden * 4));
  };
  auto open_of = [&](float v) { return BITS ? ((__float_as_uint(v) >> (col & 31)) & 1u) != 0 : v > 0.0f; };
  auto lds_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
  // Observations and dOut of the eight samples a wave produces: scalar registers, requested in
  // front of the PREVIOUS step's barrier (see mlp_wgrad_split_kernel); samples past m read as zero.
  f32x8 xq[kIn], dq;
  [[maybe_unused]] f32x8 dq_hi;  // PAIR: the samples' (g0, g1) pairs, sixteen dwords
  auto dout_of = [&](int e) { return PAIR ? (e < 4 ? dq[2 * e] : dq_hi[2 * e - 8]) : dq[e]; };
  auto request_scalars = [&](int64_t n) {
    const int64_t row0 = (blockIdx.x + n * stride) * kWsChunk + 8 * kh;
    const int64_t left = m - row0;
    const int rows = left <= 0 ? 0 : left < 8 ? (int)left : 8;
    const int64_t at = rows > 0 ? row0 : 0;
    const u32x4 rx = scalar_rsrc(x + at * kIn, rows * kIn * 4);
    xq[0] = scalar_buffer_load_x8<0>(rx);
    if constexpr (kIn > 1) xq[1] = scalar_buffer_load_x8<32>(rx);
    if constexpr (kIn > 2) xq[2] = scalar_buffer_load_x8<64>(rx);
    if constexpr (kIn > 3) xq[3] = scalar_buffer_load_x8<96>(rx);
    if constexpr (kIn > 4) xq[4] = scalar_buffer_load_x8<128>(rx);
    const u32x4 rd = scalar_rsrc(fused.dout + at * (PAIR ? 2 : 1), rows * (PAIR ? 8 : 4));
    dq = scalar_buffer_load_x8<0>(rd);
    if constexpr (PAIR) dq_hi = scalar_buffer_load_x8<32>(rd);
  };
  auto scalars_landed = [&]() {  // directly behind a barrier / lgkmcnt(0)
#pragma unroll
    for (int i = 0; i < kIn; ++i) scalar_tie(xq[i]);
    scalar_tie(dq);
    if constexpr (PAIR) scalar_tie(dq_hi);
  };
  // The chunk's operands -> stage `stage` (free from the previous step's barrier on): the
  // gate plane, then the three planes of dOut * h1, each written as soon as it is formed.
  auto produce = [&](const float (&h2v)[8], int stage) {
    const unsigned addr = p_write + stage * kWgStageBytes;
    {
      u32x4 g;
      if constexpr (BITS) {
        // the column's bit of each sample's gate word as a MASK (one signed bit-field extract: 0 or ~0): the bf16 plane
        // and the masked dOut are then one AND each -- no compare, no select (this kernel is bound by its VALU work)
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
          const uint32_t m0 = (uint32_t)__builtin_amdgcn_sbfe((int)__float_as_uint(h2v[e]), (unsigned)(col & 31), 1u);
          const uint32_t m1 = (uint32_t)__builtin_amdgcn_sbfe((int)__float_as_uint(h2v[e + 1]), (unsigned)(col & 31), 1u);
          g[e >> 1] = F16 ? (m0 & 0x00003c00u) | (m1 & 0x3c000000u)   // fp16 1.0 / 0.0
                          : (m0 & 0x00003f80u) | (m1 & 0x3f800000u);  // bf16 1.0 / 0.0
          gsum += __uint_as_float(m0 & __float_as_uint(dout_of(e)));
          gsum += __uint_as_float(m1 & __float_as_uint(dout_of(e + 1)));
        }
      } else {
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
          const bool o0 = open_of(h2v[e]), o1 = open_of(h2v[e + 1]);
          g[e >> 1] = (o0 ? 0x00003f80u : 0u) | (o1 ? 0x3f800000u : 0u);  // bf16 1.0 / 0.0
          gsum += (o0 ? dout_of(e) : 0.0f) + (o1 ? dout_of(e + 1) : 0.0f);
          dw3a = __builtin_fmaf(dout_of(e), h2v[e], dw3a);
          dw3a = __builtin_fmaf(dout_of(e + 1), h2v[e + 1], dw3a);
        }
      }
      lds_write_b128<0>(addr, g);
    }
    float b[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float v = b1r;
#pragma unroll
      for (int c = 0; c < kIn; ++c) v = __builtin_fmaf(xq[(e * kIn + c) >> 3][(e * kIn + c) & 7], w1r[c], v);
      b[e] = relu1(v) * dout_of(e);
    }
    if constexpr (F16) {
      u32x4 planes[2];
#pragma unroll
      for (int e = 0; e < 8; e += 2) {
        uint32_t hi, lo;
        f16_pair_scaled(b[e], b[e + 1], col_scale, hi, lo);
        planes[0][e >> 1] = hi;
        planes[1][e >> 1] = lo;
      }
      lds_write_b128<kWgOperandA>(addr, planes[0]);
      lds_write_b128<kWgOperandA + kWsPlane>(addr, planes[1]);
      return;
    }
    u32x4 planes[3];
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
      uint32_t hi, mid, lo;
      split_pair(b[e], b[e + 1], hi, mid, lo);
      planes[0][e >> 1] = hi;
      planes[1][e >> 1] = mid;
      planes[2][e >> 1] = lo;
    }
    lds_write_b128<kWgOperandA>(addr, planes[0]);
    lds_write_b128<kWgOperandA + kWsPlane>(addr, planes[1]);
    lds_write_b128<kWgOperandA + 2 * kWsPlane>(addr, planes[2]);
  };

  f32x16 acc[2][4];
  float hq[2][8];
  using T = std::true_type;
  using F = std::false_type;
  using P0 = std::integral_constant<int, 0>;
  using P1 = std::integral_constant<int, 1>;

  // Registers of a step: the gate fragments of the chunk in hand (G, two) and of the next one
  // (fetched behind the barrier into the other pair), and two sets of four B fragments that
  // trade roles: a step opens with its hi planes in X (even steps) or Y (odd), fetches the
  // mid planes into the other set, the lo planes into the first set once the hi products
  // are issued, and the next step's hi planes into the second behind the barrier.
  SplitFrags f;  // ah / am: gate fragments of even / odd steps; bh / bm: the B sets X / Y
  auto first_reads = [&](auto parity_tag, auto stage_tag) {  // ... of the chunk in stage S: gate -> role P's G, hi planes -> its first set
    constexpr int P = decltype(parity_tag)::value;
    constexpr int S = decltype(stage_tag)::value;
    const unsigned ar = a_read + S * kWgStageBytes, br = b_read + S * kWgStageBytes;
    u32x4(&G)[2] = *(P == 0 ? &f.ah : &f.am);
    u32x4(&B0)[4] = *((F16 || P == 0) ? &f.bh : &f.bm);  // (F16: the hi planes always in X, the lo planes in Y)
    G[0] = lds_read_b128<0>(ar);
    B0[0] = lds_read_b128<0>(br);
    B0[1] = lds_read_b128<512>(br);
    B0[2] = lds_read_b128<1024>(br);
    B0[3] = lds_read_b128<1536>(br);
    G[1] = lds_read_b128<512>(ar);
  };
  // (Round 3: a stagger of the SIMD's two waves -- waves 4..7 forming their operands one group of products later, beside
  // the mid planes' products -- measured no change: 87.1 against 87.4 ms per step of the headline bench.  Not kept.)
  // Round 3: FOUR stages and one barrier per TWO steps (SQ counters had the waves parked at waits and barriers for 27 %
  // of their cycles): chunks n, n + 1 are consumed while n + 2, n + 3 are produced; the register roles still alternate
  // per step (P = n % 2), the stage is S = n % 4, and only odd steps end in a barrier.
  auto do_step = [&](auto first_tag, auto parity_tag, auto stage_tag, int64_t n) {
    constexpr bool FIRST = decltype(first_tag)::value;
    constexpr int P = decltype(parity_tag)::value;
    constexpr int S = decltype(stage_tag)::value;
    const unsigned br = b_read + S * kWgStageBytes;
    u32x4(&G)[2] = *(P == 0 ? &f.ah : &f.am);
    if constexpr (F16) {
      // two planes: X = f.bh holds the hi planes (in from the previous step / prologue), Y = f.bm takes the lo planes;
      // the next chunk's gate and hi planes are fetched behind the barrier under the lo products
      u32x4(&X)[4] = f.bh;
      u32x4(&Y)[4] = f.bm;
      Y[0] = lds_read_b128<kWsPlane>(br);
      Y[1] = lds_read_b128<kWsPlane + 512>(br);
      Y[2] = lds_read_b128<kWsPlane + 1024>(br);
      Y[3] = lds_read_b128<kWsPlane + 1536>(br);
      __builtin_amdgcn_sched_barrier(0);
      f16_mma<FIRST>(G, X, acc);  // gate x hi
      load_h2(hq[P], n + 3);
      produce(hq[P ^ 1], (S + 2) & 3);
      request_scalars(n + 3);
      wait_lds<0>(Y[0], Y[1], Y[2], Y[3]);  // (with the scalar loads: nothing counts on order here)
      __builtin_amdgcn_sched_barrier(0);
      if constexpr ((S & 1) != 0) lds_barrier();
      else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      scalars_landed();
      if constexpr (P == 0) first_reads(P1{}, std::integral_constant<int, (S + 1) & 3>{});
      else first_reads(P0{}, std::integral_constant<int, (S + 1) & 3>{});
      __builtin_amdgcn_sched_barrier(0);
      f16_mma<false>(G, Y, acc);  // gate x lo
      __builtin_amdgcn_sched_barrier(0);
      {
        u32x4(&GN)[2] = *(P == 0 ? &f.am : &f.ah);
        wait_lds<0>(GN[0], GN[1], X[0], X[1], X[2], X[3]);
      }
      return;
    }
    u32x4(&B0)[4] = *(P == 0 ? &f.bh : &f.bm);
    u32x4(&B1)[4] = *(P == 0 ? &f.bm : &f.bh);
    // G and B0 (hi planes) are in (previous step / prologue); the mid planes:
    B1[0] = lds_read_b128<kWsPlane>(br);
    B1[1] = lds_read_b128<kWsPlane + 512>(br);
    B1[2] = lds_read_b128<kWsPlane + 1024>(br);
    B1[3] = lds_read_b128<kWsPlane + 1536>(br);
    __builtin_amdgcn_sched_barrier(0);
    split_mma<FIRST>(G, B0, acc);  // gate x hi: at once, nothing in front of the first products
    load_h2(hq[P], n + 3);
    produce(hq[P ^ 1], (S + 2) & 3);
    wait_lds<0>(B1[0], B1[1], B1[2], B1[3]);  // (only LDS operations are in flight)
    __builtin_amdgcn_sched_barrier(0);
    // the lo planes, into the hi planes' registers (their products are issued)
    B0[0] = lds_read_b128<2 * kWsPlane>(br);
    B0[1] = lds_read_b128<2 * kWsPlane + 512>(br);
    B0[2] = lds_read_b128<2 * kWsPlane + 1024>(br);
    B0[3] = lds_read_b128<2 * kWsPlane + 1536>(br);
    split_mma<false>(G, B1, acc);  // gate x mid: last use of B1
    __builtin_amdgcn_sched_barrier(0);
    request_scalars(n + 3);  // for the chunk the NEXT step produces
    wait_lds<0>(B0[0], B0[1], B0[2], B0[3]);  // (the scalar loads with them: nothing counts on order here)
    __builtin_amdgcn_sched_barrier(0);
    if constexpr ((S & 1) != 0) lds_barrier();  // end of a pair of steps: the next pair's chunks are complete, this pair's stages free
    else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    scalars_landed();
    // into the other gate pair and B1's registers (= the next step's first set)
    if constexpr (P == 0) first_reads(P1{}, std::integral_constant<int, (S + 1) & 3>{});
    else first_reads(P0{}, std::integral_constant<int, (S + 1) & 3>{});
    __builtin_amdgcn_sched_barrier(0);
    split_mma<false>(G, B0, acc);  // gate x lo
    __builtin_amdgcn_sched_barrier(0);
    // landed before anything can copy or carry these registers (loop back-edge)
    {
      u32x4(&GN)[2] = *(P == 0 ? &f.am : &f.ah);
      wait_lds<0>(GN[0], GN[1], B1[0], B1[1], B1[2], B1[3]);
    }
  };

  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  using S2 = std::integral_constant<int, 2>;
  using S3 = std::integral_constant<int, 3>;
  {
    load_h2(hq[0], 0);
    load_h2(hq[1], 1);
    request_scalars(0);
    lds_barrier();
    scalars_landed();
    produce(hq[0], 0);
    request_scalars(1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    scalars_landed();
    produce(hq[1], 1);
    load_h2(hq[1], 2);  // (step 0 produces chunk 2 from hq[1] and loads chunk 3 into hq[0])
    request_scalars(2);
    lds_barrier();
    scalars_landed();
    first_reads(P0{}, S0{});
    wait_lds<0>(f.ah[0], f.ah[1], f.bh[0], f.bh[1], f.bh[2], f.bh[3]);
  }
  {
    // a multiple of four steps, the last ones on all-zero chunks past the end if need be
    const int64_t steps = (mine + 3) & ~(int64_t)3;
    do_step(T{}, P0{}, S0{}, 0);
    do_step(F{}, P1{}, S1{}, 1);
    do_step(F{}, P0{}, S2{}, 2);
    do_step(F{}, P1{}, S3{}, 3);
#pragma unroll 1
    for (int64_t n = 4; n < steps; n += 4) {
      do_step(F{}, P0{}, S0{}, n);
      do_step(F{}, P1{}, S1{}, n + 1);
      do_step(F{}, P0{}, S2{}, n + 2);
      do_step(F{}, P1{}, S3{}, n + 3);
    }
  }

  // W3[j] (PAIR: W3[0][j] - W3[1][j]) multiplies the finished sums (row j of the slab)
  auto w3_of = [&](int j) { return PAIR ? fused.w3[j] - fused.w3[kHidden + j] : fused.w3[j]; };
  float *slab = slabs + (int64_t)blockIdx.x * kHidden * kHidden;
#pragma unroll
  for (int ja = 0; ja < 2; ++ja)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int j = 64 * wj + 32 * ja + (r & 3) + 8 * (r >> 2) + 4 * hh;
      const float w3j = BITS ? 1.0f : w3_of(j);  // (BITS: the slabs hold M; the reduction applies W3)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int i = 128 * wi + 32 * t + l32;
        slab[j * kHidden + i] = F16 ? acc[ja][t][r] * inv_scales[i] : acc[ja][t][r] * w3j;
      }
    }

  // Head gradients: fold the two sample halves (kh) of a column through LDS in a fixed
  // order and write this workgroup's partial row; zero the other kernel's segment where it
  // has no row of its own.
  float *red = reinterpret_cast<float *>(smem);  // [256][2]
  __syncthreads();
  if (kh == 1) {
    red[col * 2] = gsum;
    red[col * 2 + 1] = dw3a;
  }
  __syncthreads();
  float *row = fused.partials + (int64_t)blockIdx.x * fused.partial_stride;
  constexpr int kOut = PAIR ? 2 : 1;
  const int off_db2 = kHidden * d_in + kHidden, off_dw3 = off_db2 + kHidden, off_db3 = off_dw3 + kOut * kHidden;
  const bool more = fused.accumulate != 0;  // (a later segment: the first one wrote / zeroed the row)
  if (kh == 0) {
    const float sum_b2 = (gsum + red[col * 2]) * w3_of(col);
    row[off_db2 + col] = more ? row[off_db2 + col] + sum_b2 : sum_b2;
    // (BITS: the b2 part of dW3 = sum_i W2 M + b2 sum_s G dOut; the reduction adds the row dots to row 0)
    const float sum_w3 = BITS ? (gsum + red[col * 2]) * fused.b2[col] : dw3a + red[col * 2 + 1];
    row[off_dw3 + col] = more ? row[off_dw3 + col] + sum_w3 : sum_w3;
    if constexpr (PAIR) row[off_dw3 + kHidden + col] = more ? row[off_dw3 + kHidden + col] - sum_w3 : -sum_w3;
    // db3 = sum of dOut needs no matrix kernel: the caller forms it (the segment is zeroed)
    if (col < kOut && !more) row[off_db3 + col] = 0.0f;
  }
  if (!more && (int)blockIdx.x >= fused.other_rows)
    for (int idx = tid; idx < kHidden * d_in + kHidden; idx += kWsThreads) row[idx] = 0.0f;
}

// BITS mode of the gate-plane kernel: M = sum over slabs (slab order); dW2[j][i] (+)= w3e[j] M[j][i]; and the row
// dots sum_i W2[j][i] M[j][i] join dW3 in partial row 0 (PAIR: with opposite signs in the two rows of dW3).
// Workgroup = row j of the 256 x 256 output.
template <bool PAIR>
__global__ __launch_bounds__(kBlock) void mlp_wgrad_gate_reduce_kernel(const float *__restrict__ slabs, int rows,
                                                                      float *__restrict__ out, int accumulate,
                                                                      const float *__restrict__ w2,
                                                                      const float *__restrict__ w3,
                                                                      float *__restrict__ dw3_row0) {
  __shared__ float red[kBlock];
  const int j = blockIdx.x, i = threadIdx.x, idx = j * kBlock + i;
  const float *p = slabs + idx;
  float sum = 0.0f;
  int r = 0;
  for (; r + 16 <= rows; r += 16) {
    float v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = p[(int64_t)(r + u) * (kHidden * kHidden)];
#pragma unroll
    for (int u = 0; u < 16; ++u) sum += v[u];
  }
  for (; r < rows; ++r) sum += p[(int64_t)r * (kHidden * kHidden)];
  const float w3e = PAIR ? w3[j] - w3[kHidden + j] : w3[j];
  out[idx] = (accumulate ? out[idx] : 0.0f) + w3e * sum;
  red[i] = w2[idx] * sum;
  __syncthreads();
  for (int half = kBlock / 2; half > 0; half >>= 1) {  // fixed order
    if (i < half) red[i] += red[i + half];
    __syncthreads();
  }
  if (i == 0) {
    dw3_row0[j] += red[0];
    if constexpr (PAIR) dw3_row0[kHidden + j] -= red[0];
  }
}

// out[idx] (+)= sum over slabs, in slab order.
__global__ __launch_bounds__(kBlock) void mlp_wgrad_split_reduce_kernel(const float *__restrict__ slabs, int rows,
                                                                       float *__restrict__ out, int accumulate) {
  // One float per thread (a workgroup per CU), slabs added in slab order; the loads of
  // sixteen slabs are in flight together, only the additions are sequential.
  const int idx = blockIdx.x * kBlock + threadIdx.x;
  const float *p = slabs + idx;
  float sum = accumulate ? out[idx] : 0.0f;
  int r = 0;
  for (; r + 16 <= rows; r += 16) {
    float v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = p[(int64_t)(r + u) * (kHidden * kHidden)];
#pragma unroll
    for (int u = 0; u < 16; ++u) sum += v[u];
  }
  for (; r < rows; ++r) sum += p[(int64_t)r * (kHidden * kHidden)];
  out[idx] = sum;
}

template <int DIN>
static int launch_wgrad_split(int grid, hipStream_t s, const float *dz2, const float *x, const float *w1,
                              const float *b1, int64_t m, int d_in, float *slabs) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&mlp_wgrad_split_kernel<DIN>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipGetLastError();
    attr_set = true;
  }
  mlp_wgrad_split_kernel<DIN><<<grid, kWsThreads, 2 * kWsStageBytes, s>>>(dz2, x, w1, b1, m, d_in, slabs, WgradFusedArgs{},
                                                                           WgradOperands{});
  return launch_status();
}

template <int DIN, int NOUT, bool F16 = false>
static int launch_wgrad_fused(int grid, hipStream_t s, const float *h2, const float *x, const float *w1,
                              const float *b1, int64_t m, int d_in, float *slabs, WgradFusedArgs fused) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&mlp_wgrad_split_kernel<DIN, NOUT, false, F16>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipGetLastError();
    attr_set = true;
  }
  // (F16: + the inverse powers of two of the 256 columns of each operand behind the two stages)
  mlp_wgrad_split_kernel<DIN, NOUT, false, F16><<<grid, kWsThreads, 2 * kWsStageBytes + (F16 ? 2 * kHidden * 4 : 0), s>>>(
      h2, x, w1, b1, m, d_in, slabs, fused, WgradOperands{});
  return launch_status();
}

template <int DIN, bool PAIR = false, bool BITS = false, bool F16 = false>
static int launch_wgrad_gate(int grid, hipStream_t s, const float *h2, const float *x, const float *w1,
                             const float *b1, int64_t m, float *slabs, WgradFusedArgs fused) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&mlp_wgrad_gate_kernel<DIN, PAIR, BITS, F16>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipGetLastError();
    attr_set = true;
  }
  // (F16: + the columns' inverse powers of two behind the four stages)
  mlp_wgrad_gate_kernel<DIN, PAIR, BITS, F16><<<grid, kWsThreads, 4 * kWgStageBytes + (F16 ? kHidden * 4 : 0), s>>>(
      h2, x, w1, b1, m, slabs, fused);
  return launch_status();
}

// max |dOut[s][q]| per output q and max |x[s][c]| per column c over m rows -> bounds[q], bounds[4 + c] (bit patterns of
// non-negative floats, combined with atomic max on words the caller zeroed).  dOut rows have NOUT floats, x rows DIN;
// both are read as flat 16-byte vectors wherever aligned, the column of a flat index being its remainder.
template <int DIN, int NOUT>
__global__ __launch_bounds__(kBlock) void wgrad_bounds_kernel(const float *__restrict__ dout, const float *__restrict__ x,
                                                              int64_t m, uint32_t *__restrict__ bounds) {
  __shared__ float red[kBlock / kWave][NOUT + DIN];
  float mx[DIN], md[NOUT];
#pragma unroll
  for (int c = 0; c < DIN; ++c) mx[c] = 0.0f;
#pragma unroll
  for (int q = 0; q < NOUT; ++q) md[q] = 0.0f;
  const int64_t tid = (int64_t)blockIdx.x * kBlock + threadIdx.x, threads = (int64_t)gridDim.x * kBlock;
  auto scan = [&](const float *p, int64_t n, auto fold) {  // fold(flat index, value) over p[0 .. n)
    const int64_t vecs = ((uintptr_t)p & 15) == 0 ? n / 4 : 0;
    int64_t q = tid;
    for (; q + 3 * threads < vecs; q += 4 * threads) {  // four loads in the air per lane
      f32x4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = reinterpret_cast<const f32x4 *>(p)[q + u * threads];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i) fold(4 * (q + u * threads) + i, v[u][i]);
    }
    for (; q < vecs; q += threads) {
      const f32x4 v = reinterpret_cast<const f32x4 *>(p)[q];
#pragma unroll
      for (int i = 0; i < 4; ++i) fold(4 * q + i, v[i]);
    }
    for (int64_t idx = 4 * vecs + tid; idx < n; idx += threads) fold(idx, p[idx]);
  };
  scan(dout, m * NOUT, [&](int64_t idx, float v) {
    const int c = (int)(idx % NOUT);
#pragma unroll
    for (int k = 0; k < NOUT; ++k)
      if (k == c) md[k] = __builtin_fmaxf(md[k], __builtin_fabsf(v));
  });
  scan(x, m * DIN, [&](int64_t idx, float v) {
    const int c = (int)(idx % DIN);
#pragma unroll
    for (int k = 0; k < DIN; ++k)
      if (k == c) mx[k] = __builtin_fmaxf(mx[k], __builtin_fabsf(v));
  });
  auto wave_max = [](float v) {
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) v = __builtin_fmaxf(v, __shfl_down(v, off, kWave));
    return v;
  };
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
#pragma unroll
  for (int q = 0; q < NOUT; ++q) md[q] = wave_max(md[q]);
#pragma unroll
  for (int c = 0; c < DIN; ++c) mx[c] = wave_max(mx[c]);
  if (lane == 0) {
#pragma unroll
    for (int q = 0; q < NOUT; ++q) red[wave][q] = md[q];
#pragma unroll
    for (int c = 0; c < DIN; ++c) red[wave][NOUT + c] = mx[c];
  }
  __syncthreads();
  if (threadIdx.x < NOUT + DIN) {
    float v = red[0][threadIdx.x];
    for (int w = 1; w < kBlock / kWave; ++w) v = __builtin_fmaxf(v, red[w][threadIdx.x]);
    atomicMax(bounds + (threadIdx.x < NOUT ? threadIdx.x : 4 + threadIdx.x - NOUT), __float_as_uint(v));
  }
}

// the bounds behind the slabs of the caller's workspace (rl8_mlp_wgrad_workspace_bytes), filled for the whole call
template <int NOUT>
static uint32_t *launch_wgrad_bounds(hipStream_t s, const float *dout, const float *x, int64_t m, int d_in, float *workspace) {
  uint32_t *bounds = reinterpret_cast<uint32_t *>(workspace + (int64_t)kCUs * kHidden * kHidden);
  if (hipMemsetAsync(bounds, 0, 64, s) != hipSuccess) return nullptr;
  const int64_t want = m / (4 * kBlock);
  const int grid = (int)(want < 1 ? 1 : want > 4 * kCUs ? 4 * kCUs : want);  // (one atomic per block and column: 12 ns each)
  if (d_in == 1) wgrad_bounds_kernel<1, NOUT><<<grid, kBlock, 0, s>>>(dout, x, m, bounds);
  else if (d_in == 2) wgrad_bounds_kernel<2, NOUT><<<grid, kBlock, 0, s>>>(dout, x, m, bounds);
  else if (d_in == 3) wgrad_bounds_kernel<3, NOUT><<<grid, kBlock, 0, s>>>(dout, x, m, bounds);
  else if (d_in == 5) wgrad_bounds_kernel<5, NOUT><<<grid, kBlock, 0, s>>>(dout, x, m, bounds);
  else return nullptr;
  return bounds;
}

// flag[0] |= 1 if any row has dout[s][0] + dout[s][1] != 0 (bit patterns: g1 must be exactly -g0).
__global__ __launch_bounds__(kBlock) void dout_pair_check_kernel(const uint32_t *__restrict__ dout, int64_t m,
                                                               int *__restrict__ flag) {
  bool bad = false;
  for (int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x; r < m; r += (int64_t)gridDim.x * kBlock) {
    const uint2 v = reinterpret_cast<const uint2 *>(dout)[r];
    // exact negatives: equal magnitudes, opposite signs -- or both zeros of any sign
    bad |= !(((v.x ^ v.y) == 0x80000000u) || (((v.x | v.y) & 0x7fffffffu) == 0u));
  }
  if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

}  // namespace rl8

using namespace rl8;

#ifdef RL8_SPLIT_TRACE
RL8_API int rl8_debug_split_trace(unsigned long long *host_dst) {
  return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(g_split_trace), sizeof(g_split_trace));
}
RL8_API int rl8_debug_split_trace_epilogue(unsigned long long *host_dst) {
  return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(g_split_trace_epilogue), sizeof(g_split_trace_epilogue));
}
#endif

RL8_API int64_t rl8_mlp_split_packed_bytes(void) { return kSplitPackedBytes; }

RL8_API int rl8_mlp_pack_w2_split(const float *w2, int transposed, void *packed, void *stream) {
  if (!w2 || !packed) return RL8_ENULL;
  if (((uintptr_t)packed & 15) != 0) return RL8_EALIGN;
  mlp_pack_w2_split_kernel<<<(kSplitSteps * 8 * 64 + kBlock - 1) / kBlock, kBlock, 0, (hipStream_t)stream>>>(
      w2, transposed, reinterpret_cast<uint32_t *>(packed));
  return launch_status();
}

RL8_API int rl8_mlp_tower_forward_split_f32(const float *x, int64_t m, int d_in, const float *w1,
                                            const float *b1, const void *w2_split, const float *b2,
                                            const float *w3, const float *b3, int n_out, float *out,
                                            float *save_h1, float *save_h2, uint32_t *save_gate2, void *stream) {
  if (!x || !w1 || !b1 || !w2_split || !b2 || !w3 || !b3 || !out) return RL8_ENULL;
  if ((save_h1 != nullptr || save_gate2 != nullptr) && save_h2 == nullptr) return RL8_ENULL;  // h2 alone is allowed
  if (m <= 0 || d_in <= 0 || d_in > kMaxIn || n_out <= 0 || n_out > kMaxOut) return RL8_ESIZE;
  if (((uintptr_t)w2_split & 15) != 0 || !aligned16(w1) || !aligned16(b1) || (save_h1 && !aligned16(save_h1)) ||
      (save_h2 && !aligned16(save_h2)))
    return RL8_EALIGN;
  const int64_t tiles = (m + kSplitRows - 1) / kSplitRows;
  static const int cap = env_int("RL8_MLP_GRID_CAP");
  const int max_grid = cap > 0 ? cap : 2 * kCUs;
  const int grid = (int)(tiles < max_grid ? tiles : max_grid);
  hipStream_t s = (hipStream_t)stream;
  // Only the widths whose kernels are verified spill-free are compiled (see
  // rl8_mlp_forward_split_supports); the rest keep rl8_mlp_tower_forward_f32.
  switch (d_in) {
    case 1: return dispatch_forward_split_nout<1>(n_out, grid, s, x, m, d_in, w1, b1, w2_split, b2, w3, b3, out, save_h1, save_h2, save_gate2);
    case 2: return dispatch_forward_split_nout<2>(n_out, grid, s, x, m, d_in, w1, b1, w2_split, b2, w3, b3, out, save_h1, save_h2, save_gate2);
    case 3: return dispatch_forward_split_nout<3>(n_out, grid, s, x, m, d_in, w1, b1, w2_split, b2, w3, b3, out, save_h1, save_h2, save_gate2);
    case 5: return dispatch_forward_split_nout<5>(n_out, grid, s, x, m, d_in, w1, b1, w2_split, b2, w3, b3, out, save_h1, save_h2, save_gate2);
    default: return RL8_ESIZE;
  }
}

// These kernels read LDS through inline asm the compiler cannot see; a register
// spill placed between such a read and its wait could save a register whose data
// has not arrived.  So a width is offered only if its kernel compiles WITHOUT
// scratch (tests/test_kernel_resources.py holds the list to that); other widths
// use the fp32-MFMA kernels.
RL8_API int rl8_mlp_forward_split_supports(int d_in, int n_out) {
  return (d_in == 1 || d_in == 2 || d_in == 3 || d_in == 5) && n_out >= 1 && n_out <= 3;
}

static void fused_backward_grids(int64_t m, int *g1, int *g2);

RL8_API int rl8_mlp_tower_backward_split_f32(const float *x, const float *w1, const float *b1,
                                             const float *h2, const float *dout, int64_t m, int d_in,
                                             const void *w2t_split, const float *w3, int n_out,
                                             float *dz2_out, float *partials, int *partial_rows_out,
                                             const uint32_t *gate2, void *stream) {
  if (!x || !w1 || !b1 || (!h2 && (!gate2 || dz2_out)) || !dout || !w2t_split || !w3 || !partials || !partial_rows_out)
    return RL8_ENULL;
  if (m <= 0 || d_in <= 0 || d_in > kMaxIn || n_out <= 0 || n_out > kMaxOut) return RL8_ESIZE;
  if (((uintptr_t)w2t_split & 15) != 0 || !aligned16(h2) || !aligned16(dz2_out) || !aligned16(w3)) return RL8_EALIGN;
  int grid, g2;
  fused_backward_grids(m, &grid, &g2);
  // dz2_out == NULL: first half of the fused backward -- no dZ2 store, no head-gradient
  // launch; rl8_mlp_wgrad_fused_split_f32 fills the head segments of rows < g2.
  const int head_rows = dz2_out ? -1 : g2;
  *partial_rows_out = dz2_out ? grid : (grid > g2 ? grid : g2);
  const int stride = (int)rl8_mlp_backward_partial_floats(d_in, n_out);
  hipStream_t s = (hipStream_t)stream;
  switch (d_in) {
    case 1: return dispatch_backward_split_nout<1>(n_out, grid, s, x, w1, b1, h2, dout, m, d_in, w2t_split, w3, dz2_out, partials, stride, head_rows, gate2);
    case 2: return dispatch_backward_split_nout<2>(n_out, grid, s, x, w1, b1, h2, dout, m, d_in, w2t_split, w3, dz2_out, partials, stride, head_rows, gate2);
    case 3: return dispatch_backward_split_nout<3>(n_out, grid, s, x, w1, b1, h2, dout, m, d_in, w2t_split, w3, dz2_out, partials, stride, head_rows, gate2);
    case 5: return dispatch_backward_split_nout<5>(n_out, grid, s, x, w1, b1, h2, dout, m, d_in, w2t_split, w3, dz2_out, partials, stride, head_rows, gate2);
    default: return RL8_ESIZE;  // rl8_mlp_backward_split_supports(): other widths use rl8_mlp_tower_backward_f32
  }
}

RL8_API int rl8_mlp_backward_split_supports(int d_in, int n_out) {
  return rl8_mlp_forward_split_supports(d_in, n_out);  // spill-free widths only
}

/* dW2 (+)= dZ2^T h1 with h1 recomputed from the observations (see the kernel). */
RL8_API int rl8_mlp_wgrad_split_f32(const float *dz2, const float *x, const float *w1, const float *b1,
                                    int64_t m, int d_in, float *workspace, float *dw2_out, int accumulate,
                                    void *stream) {
  if (!dz2 || !x || !w1 || !b1 || !workspace || !dw2_out) return RL8_ENULL;
  if (m <= 0 || d_in <= 0 || d_in > kMaxIn) return RL8_ESIZE;
  if (!aligned16(dz2) || !aligned16(workspace) || !aligned16(dw2_out)) return RL8_EALIGN;
  hipStream_t s = (hipStream_t)stream;
  // Segments of kWgradSegmentRows samples, summed in order (see there).
  for (int64_t at = 0; at < m; at += kWgradSegmentRows) {
    const int64_t rows = m - at < kWgradSegmentRows ? m - at : kWgradSegmentRows;
    const int64_t chunks = (rows + kWsChunk - 1) / kWsChunk;
    const int grid = (int)(chunks < kCUs ? chunks : kCUs);
    const float *dz = dz2 + at * kHidden, *xs = x + at * d_in;
    int status;
    switch (d_in) {
      case 1: status = launch_wgrad_split<1>(grid, s, dz, xs, w1, b1, rows, d_in, workspace); break;
      case 2: status = launch_wgrad_split<2>(grid, s, dz, xs, w1, b1, rows, d_in, workspace); break;
      case 3: status = launch_wgrad_split<3>(grid, s, dz, xs, w1, b1, rows, d_in, workspace); break;
      case 5: status = launch_wgrad_split<5>(grid, s, dz, xs, w1, b1, rows, d_in, workspace); break;
      default: status = launch_wgrad_split<0>(grid, s, dz, xs, w1, b1, rows, d_in, workspace); break;
    }
    if (status != 0) return status;
    mlp_wgrad_split_reduce_kernel<<<kHidden * kHidden / kBlock, kBlock, 0, s>>>(workspace, grid, dw2_out,
                                                                                 accumulate || at > 0);
  }
  return launch_status();
}

template <int DIN, bool F16 = false>
static int launch_wgrad_loadh(int grid, hipStream_t s, const float *dz, const float *x, int64_t rows, float *workspace,
                              const WgradOperands &ops) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&mlp_wgrad_split_kernel<DIN, 0, true, F16>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipGetLastError();
    attr_set = true;
  }
  mlp_wgrad_split_kernel<DIN, 0, true, F16><<<grid, kWsThreads, 2 * kWsStageBytes + (F16 ? 2 * kHidden * 4 : 0), s>>>(
      dz, x, nullptr, nullptr, rows, DIN, workspace, WgradFusedArgs{}, ops);
  return launch_status();
}

/* dW (+)= dZ^T h with BOTH operands strided in memory (dZ rows at dz_pitch, h rows at
 * h_pitch floats, 256 columns each): the LSTM's recurrent weight gradient per gate.
 * x / d_in / colsums (all optional together): also the column sums
 *   colsums[wg][col * d_in + i] = sum_rows dZ[row][col] * x[row][i],  colsums[wg][256 * d_in + col] = sum_rows dZ[row][col]
 * per workgroup (rows of 256 * (d_in + 1) floats, *colsum_rows_out of them; the caller adds them
 * up in row order) -- the LSTM's dW_ih and bias gradients of that gate; x dense [m][d_in],
 * d_in in {1, 2, 3, 5}. */
static int wgrad_strided(const float *dz, int64_t dz_pitch, const float *h, int64_t h_pitch, int64_t m, float *workspace,
                         float *dw_out, int accumulate, const float *x, int d_in, float *colsums, int *colsum_rows_out,
                         const uint32_t *dz_bound, const uint32_t *h_bound, void *stream) {
  if (!dz || !h || !workspace || !dw_out) return RL8_ENULL;
  if (m <= 0 || dz_pitch < kHidden || h_pitch < kHidden) return RL8_ESIZE;
  if ((int64_t)kWsChunk * (dz_pitch > h_pitch ? dz_pitch : h_pitch) * 4 >= (int64_t)1 << 31) return RL8_ESIZE;
  if (!aligned16(workspace) || !aligned16(dw_out)) return RL8_EALIGN;
  if (colsums) {
    if (!x || !colsum_rows_out) return RL8_ENULL;
    if (!(d_in == 1 || d_in == 2 || d_in == 3 || d_in == 5)) return RL8_ESIZE;
  }
  hipStream_t s = (hipStream_t)stream;
  int first_grid = 0;
  for (int64_t at = 0; at < m; at += kWgradSegmentRows) {  // segments summed in order, as above
    const int64_t rows = m - at < kWgradSegmentRows ? m - at : kWgradSegmentRows;
    const int64_t chunks = (rows + kWsChunk - 1) / kWsChunk;
    int grid = (int)(chunks < kCUs ? chunks : kCUs);
    if (at == 0) first_grid = grid;
    if (grid > first_grid) grid = first_grid;  // (later segments add to the first one's rows)
    const WgradOperands ops{h + at * h_pitch, (int)dz_pitch, (int)h_pitch, colsums, at > 0, dz_bound, h_bound};
    const float *xs = colsums ? x + at * d_in : nullptr;
    int status;
    if (dz_bound) {
      switch (colsums ? d_in : 1) {
        case 1: status = launch_wgrad_loadh<1, true>(grid, s, dz + at * dz_pitch, xs, rows, workspace, ops); break;
        case 2: status = launch_wgrad_loadh<2, true>(grid, s, dz + at * dz_pitch, xs, rows, workspace, ops); break;
        case 3: status = launch_wgrad_loadh<3, true>(grid, s, dz + at * dz_pitch, xs, rows, workspace, ops); break;
        default: status = launch_wgrad_loadh<5, true>(grid, s, dz + at * dz_pitch, xs, rows, workspace, ops); break;
      }
    } else {
      switch (colsums ? d_in : 1) {
        case 1: status = launch_wgrad_loadh<1>(grid, s, dz + at * dz_pitch, xs, rows, workspace, ops); break;
        case 2: status = launch_wgrad_loadh<2>(grid, s, dz + at * dz_pitch, xs, rows, workspace, ops); break;
        case 3: status = launch_wgrad_loadh<3>(grid, s, dz + at * dz_pitch, xs, rows, workspace, ops); break;
        default: status = launch_wgrad_loadh<5>(grid, s, dz + at * dz_pitch, xs, rows, workspace, ops); break;
      }
    }
    if (status != 0) return status;
    mlp_wgrad_split_reduce_kernel<<<kHidden * kHidden / kBlock, kBlock, 0, s>>>(workspace, grid, dw_out,
                                                                                 accumulate || at > 0);
  }
  if (colsum_rows_out) *colsum_rows_out = first_grid;
  return launch_status();
}

RL8_API int rl8_mlp_wgrad_split_strided_f32(const float *dz, int64_t dz_pitch, const float *h, int64_t h_pitch,
                                            int64_t m, float *workspace, float *dw_out, int accumulate,
                                            const float *x, int d_in, float *colsums, int *colsum_rows_out,
                                            void *stream) {
  return wgrad_strided(dz, dz_pitch, h, h_pitch, m, workspace, dw_out, accumulate, x, d_in, colsums, colsum_rows_out, nullptr,
                       nullptr, stream);
}

/* The same product on fp16 planes (three plane products per 16 rows instead of six): dZ scaled by one power of two taken
 * from *dz_bound -- a device word holding the bit pattern of a bound on |dZ| over every row this call reads, e.g. what
 * rl8_lstm_rows_backward_f32 leaves in dg_bound_out -- and h by one from *h_bound (a device float >= max |h|; 1 for an
 * LSTM's outputs).  Entries far below the bound keep an absolute, not a relative, accuracy (2^-39 of the bound per
 * term). */
RL8_API int rl8_mlp_wgrad_f16_strided_f32(const float *dz, int64_t dz_pitch, const uint32_t *dz_bound, const float *h,
                                          int64_t h_pitch, const uint32_t *h_bound, int64_t m, float *workspace,
                                          float *dw_out, int accumulate, const float *x, int d_in, float *colsums,
                                          int *colsum_rows_out, void *stream) {
  if (!dz_bound || !h_bound) return RL8_ENULL;
  return wgrad_strided(dz, dz_pitch, h, h_pitch, m, workspace, dw_out, accumulate, x, d_in, colsums, colsum_rows_out, dz_bound,
                       h_bound, stream);
}

// Grids of the two halves of the fused backward (both derive them from m alone,
// so that each can zero the partial-row segments the other does not cover).
static void fused_backward_grids(int64_t m, int *g1, int *g2) {
  const int64_t tiles = (m + kSplitRows - 1) / kSplitRows;
  const int64_t chunks = (m + kWsChunk - 1) / kWsChunk;
  static const int cap = env_int("RL8_MLP_GRID_CAP");
  const int max_grid = cap > 0 ? cap : 2 * kCUs;
  *g1 = (int)(tiles < max_grid ? tiles : max_grid);
  *g2 = (int)(chunks < kCUs ? chunks : kCUs);
}

RL8_API int rl8_mlp_wgrad_fused_split_f32(const float *h2, const float *dout, const float *x, const float *w1,
                                          const float *b1, const float *w3, int64_t m, int d_in, int n_out,
                                          float *workspace, float *dw2_out, float *partials, void *stream) {
  if (!h2 || !dout || !x || !w1 || !b1 || !w3 || !workspace || !dw2_out || !partials) return RL8_ENULL;
  if (m <= 0 || !rl8_mlp_backward_split_supports(d_in, n_out)) return RL8_ESIZE;
  if (!aligned16(h2) || !aligned16(workspace) || !aligned16(dw2_out)) return RL8_EALIGN;
  int g1, g2;
  fused_backward_grids(m, &g1, &g2);
  const int stride = (int)rl8_mlp_backward_partial_floats(d_in, n_out);
  hipStream_t s = (hipStream_t)stream;
  // RL8_WGRAD_PLANES=bf16: the six-product kernel (A/B runs); default: two fp16 planes per operand, three products
  static const bool f16 = [] {
    const char *v = getenv("RL8_WGRAD_PLANES");
    return !(v && v[0] == 'b');
  }();
  uint32_t *bounds = nullptr;
  if (f16) {
    bounds = n_out == 1 ? launch_wgrad_bounds<1>(s, dout, x, m, d_in, workspace)
             : n_out == 2 ? launch_wgrad_bounds<2>(s, dout, x, m, d_in, workspace)
                          : launch_wgrad_bounds<3>(s, dout, x, m, d_in, workspace);
    if (!bounds) return launch_status() ? launch_status() : RL8_ESIZE;
  }
  // Segments of kWgradSegmentRows samples, summed in order (see there).  The first one
  // runs the grid the data-gradient kernel counted on (g2 rows of partials written, the
  // rest zeroed); later ones add to as many of those rows as they have workgroups.
  for (int64_t at = 0; at < m; at += kWgradSegmentRows) {
    const int64_t rows = m - at < kWgradSegmentRows ? m - at : kWgradSegmentRows;
    const int64_t chunks = (rows + kWsChunk - 1) / kWsChunk;
    const int grid = at == 0 ? g2 : (int)(chunks < g2 ? chunks : g2);
    const WgradFusedArgs fused{dout + at * n_out, w3, partials, stride, g1, at > 0, nullptr, nullptr, bounds};
    const float *h2s = h2 + at * kHidden, *xs = x + at * d_in;
    int status = RL8_ESIZE;
    // one output: the gate-plane kernel (three plane products per 16 samples instead of six)
    static const bool gate_kernel = env_int("RL8_WGRAD_GATE_OFF") == 0;
    if (n_out == 1 && gate_kernel) {
      switch (d_in) {
        case 1: status = launch_wgrad_gate<1>(grid, s, h2s, xs, w1, b1, rows, workspace, fused); break;
        case 2: status = launch_wgrad_gate<2>(grid, s, h2s, xs, w1, b1, rows, workspace, fused); break;
        case 3: status = launch_wgrad_gate<3>(grid, s, h2s, xs, w1, b1, rows, workspace, fused); break;
        default: status = launch_wgrad_gate<5>(grid, s, h2s, xs, w1, b1, rows, workspace, fused); break;
      }
      if (status != 0) return status;
      mlp_wgrad_split_reduce_kernel<<<kHidden * kHidden / kBlock, kBlock, 0, s>>>(workspace, grid, dw2_out, at > 0);
      continue;
    }
#define RL8_WGRAD_FUSED(D, N) \
  if (d_in == D && n_out == N) \
    status = f16 ? launch_wgrad_fused<D, N, true>(grid, s, h2s, xs, w1, b1, rows, d_in, workspace, fused) \
                 : launch_wgrad_fused<D, N>(grid, s, h2s, xs, w1, b1, rows, d_in, workspace, fused);
    RL8_WGRAD_FUSED(1, 1) RL8_WGRAD_FUSED(1, 2) RL8_WGRAD_FUSED(1, 3)
    RL8_WGRAD_FUSED(2, 1) RL8_WGRAD_FUSED(2, 2) RL8_WGRAD_FUSED(2, 3)
    RL8_WGRAD_FUSED(3, 1) RL8_WGRAD_FUSED(3, 2) RL8_WGRAD_FUSED(3, 3)
    RL8_WGRAD_FUSED(5, 1) RL8_WGRAD_FUSED(5, 2) RL8_WGRAD_FUSED(5, 3)
#undef RL8_WGRAD_FUSED
    if (status != 0) return status;
    mlp_wgrad_split_reduce_kernel<<<kHidden * kHidden / kBlock, kBlock, 0, s>>>(workspace, grid, dw2_out, at > 0);
  }
  return launch_status();
}

/* dout [m][2]: *flag_out (device int, zeroed here) becomes 1 unless dout[s][1] == -dout[s][0] bit for bit in every row. */
RL8_API int rl8_mlp_dout_pair_check(const float *dout, int64_t m, int *flag_out, void *stream) {
  if (!dout || !flag_out) return RL8_ENULL;
  if (m <= 0) return RL8_ESIZE;
  if (((uintptr_t)dout & 7) != 0) return RL8_EALIGN;
  hipStream_t s = (hipStream_t)stream;
  const hipError_t err = hipMemsetAsync(flag_out, 0, sizeof(int), s);
  if (err != hipSuccess) return (int)err;
  const int64_t blocks = (m + kBlock - 1) / kBlock;
  dout_pair_check_kernel<<<(int)(blocks < 8 * kCUs ? blocks : 8 * kCUs), kBlock, 0, s>>>(
      reinterpret_cast<const uint32_t *>(dout), m, flag_out);
  return launch_status();
}

/* rl8_mlp_wgrad_fused_split_f32 for a head of two outputs whose gradients are exact negatives
 * (dout[s][1] == -dout[s][0]: rl8_mlp_dout_pair_check): the gate-plane kernel, three plane
 * products per 16 samples instead of six. */
RL8_API int rl8_mlp_wgrad_fused_pair_f32(const float *h2, const float *dout, const float *x, const float *w1,
                                         const float *b1, const float *w3, int64_t m, int d_in,
                                         float *workspace, float *dw2_out, float *partials, void *stream) {
  if (!h2 || !dout || !x || !w1 || !b1 || !w3 || !workspace || !dw2_out || !partials) return RL8_ENULL;
  if (m <= 0 || !rl8_mlp_backward_split_supports(d_in, 2)) return RL8_ESIZE;
  if (!aligned16(h2) || !aligned16(workspace) || !aligned16(dw2_out) || ((uintptr_t)dout & 7) != 0) return RL8_EALIGN;
  int g1, g2;
  fused_backward_grids(m, &g1, &g2);
  const int stride = (int)rl8_mlp_backward_partial_floats(d_in, 2);
  hipStream_t s = (hipStream_t)stream;
  for (int64_t at = 0; at < m; at += kWgradSegmentRows) {  // segments summed in order, as rl8_mlp_wgrad_fused_split_f32
    const int64_t rows = m - at < kWgradSegmentRows ? m - at : kWgradSegmentRows;
    const int64_t chunks = (rows + kWsChunk - 1) / kWsChunk;
    const int grid = at == 0 ? g2 : (int)(chunks < g2 ? chunks : g2);
    const WgradFusedArgs fused{dout + at * 2, w3, partials, stride, g1, at > 0};
    const float *h2s = h2 + at * kHidden, *xs = x + at * d_in;
    int status;
    switch (d_in) {
      case 1: status = launch_wgrad_gate<1, true>(grid, s, h2s, xs, w1, b1, rows, workspace, fused); break;
      case 2: status = launch_wgrad_gate<2, true>(grid, s, h2s, xs, w1, b1, rows, workspace, fused); break;
      case 3: status = launch_wgrad_gate<3, true>(grid, s, h2s, xs, w1, b1, rows, workspace, fused); break;
      default: status = launch_wgrad_gate<5, true>(grid, s, h2s, xs, w1, b1, rows, workspace, fused); break;
    }
    if (status != 0) return status;
    mlp_wgrad_split_reduce_kernel<<<kHidden * kHidden / kBlock, kBlock, 0, s>>>(workspace, grid, dw2_out, at > 0);
  }
  return launch_status();
}

/* The weight gradient of a rank-one head from the gate BITS alone (no h2): n_out = 1, or n_out = 2 with
 * dout[s][1] == -dout[s][0] in every row (rl8_mlp_dout_pair_check).  Same outputs as
 * rl8_mlp_wgrad_fused_split_f32 / _pair_f32 -- dW2, and the head segments [db2 | dW3] of the partial rows --
 * with dW3 = sum_i W2[.][i] M[.][i] + b2 * sum_s gate * dOut taken from the sums M the kernel forms anyway
 * (dW2 = W3 M), so that neither this call nor the forward pass (rl8_mlp_tower_forward_f16_f32 with save_h2 =
 * NULL, save_gate2 given) touches the 1 KiB per row of h2.  w2: the layer's [256][256] weight, b2 its bias. */
RL8_API int rl8_mlp_wgrad_gate_bits_f32(const uint32_t *gate2, const float *dout, const float *x, const float *w1,
                                        const float *b1, const float *w2, const float *b2, const float *w3,
                                        int64_t m, int d_in, int n_out, float *workspace, float *dw2_out,
                                        float *partials, void *stream) {
  if (!gate2 || !dout || !x || !w1 || !b1 || !w2 || !b2 || !w3 || !workspace || !dw2_out || !partials) return RL8_ENULL;
  if (m <= 0 || (n_out != 1 && n_out != 2) || !rl8_mlp_backward_split_supports(d_in, n_out)) return RL8_ESIZE;
  if (!aligned16(gate2) || !aligned16(workspace) || !aligned16(dw2_out) || ((uintptr_t)dout & 7) != 0) return RL8_EALIGN;
  int g1, g2;
  fused_backward_grids(m, &g1, &g2);
  const int stride = (int)rl8_mlp_backward_partial_floats(d_in, n_out);
  const int off_dw3 = kHidden * d_in + 2 * kHidden;
  hipStream_t s = (hipStream_t)stream;
  // RL8_WGRAD_GATE_PLANES=bf16: the three-plane kernel (A/B runs); default: two fp16 planes
  static const bool f16 = [] {
    const char *v = getenv("RL8_WGRAD_GATE_PLANES");
    return !(v && v[0] == 'b');
  }();
  uint32_t *bounds = nullptr;
  if (f16) {
    bounds = n_out == 2 ? launch_wgrad_bounds<2>(s, dout, x, m, d_in, workspace) : launch_wgrad_bounds<1>(s, dout, x, m, d_in, workspace);
    if (!bounds) return launch_status() ? launch_status() : RL8_ESIZE;
  }
  for (int64_t at = 0; at < m; at += kWgradSegmentRows) {  // segments summed in order, as rl8_mlp_wgrad_fused_split_f32
    const int64_t rows = m - at < kWgradSegmentRows ? m - at : kWgradSegmentRows;
    const int64_t chunks = (rows + kWsChunk - 1) / kWsChunk;
    const int grid = at == 0 ? g2 : (int)(chunks < g2 ? chunks : g2);
    const WgradFusedArgs fused{dout + at * n_out, w3, partials, stride, g1, at > 0, gate2 + at * 8, b2, bounds};
    const float *xs = x + at * d_in;
    int status = RL8_ESIZE;
#define RL8_WGRAD_BITS(D) \
  if (d_in == D) \
    status = f16 ? (n_out == 2 ? launch_wgrad_gate<D, true, true, true>(grid, s, nullptr, xs, w1, b1, rows, workspace, fused) \
                               : launch_wgrad_gate<D, false, true, true>(grid, s, nullptr, xs, w1, b1, rows, workspace, fused)) \
           : n_out == 2 ? launch_wgrad_gate<D, true, true>(grid, s, nullptr, xs, w1, b1, rows, workspace, fused) \
                        : launch_wgrad_gate<D, false, true>(grid, s, nullptr, xs, w1, b1, rows, workspace, fused);
    RL8_WGRAD_BITS(1) RL8_WGRAD_BITS(2) RL8_WGRAD_BITS(3) RL8_WGRAD_BITS(5)
#undef RL8_WGRAD_BITS
    if (status != 0) return status;
    if (n_out == 2)
      mlp_wgrad_gate_reduce_kernel<true><<<kHidden, kBlock, 0, s>>>(workspace, grid, dw2_out, at > 0, w2, w3, partials + off_dw3);
    else
      mlp_wgrad_gate_reduce_kernel<false><<<kHidden, kBlock, 0, s>>>(workspace, grid, dw2_out, at > 0, w2, w3, partials + off_dw3);
  }
  return launch_status();
}
