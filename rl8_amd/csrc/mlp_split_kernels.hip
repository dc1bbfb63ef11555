// N1: the weight gradients of the default towers' 256x256 layer (and, strided, of the LSTM's recurrent
// weights) as fp32-ACCURATE products on the 16-bit matrix pipes.
//
// An fp32 value is exactly the sum of three bf16 values (8 + 8 + 8 significand bits, by truncation: hi = top16(x),
// mid = top16(x - hi), lo = x - hi - mid), bf16 x bf16 products are exact in fp32 and the MFMA accumulates in fp32, so
//   a*b = ah*bh + (ah*bm + am*bh) + (ah*bl + am*bm + al*bh) + O(2^-24 |a*b|)
// -- six bf16 MFMAs per 16 k instead of eight fp32 MFMAs of 1/16 the rate (the fp32 MFMAs share the VALU's multipliers:
// 155 TFLOP/s and nothing overlaps with them, tools/probes/mfma_valu_probe.hip; the 16-bit pipe sustains 1.5-1.7 PFLOP/s
// with VALU work beside it).  Round 3 moved the default paths to two fp16 planes per operand, each scaled by a power of
// two per column of the OUTPUT it indexes (three plane products; two when one operand is the ReLU gate itself): the
// sixteen-wave kernels below (round 4; the eight-wave fp16 variants they replaced were removed in round 5).  The
// eight-wave kernels keep the exact bf16 forms: what the guard of the fp16 planes falls back to, what
// RL8_WGRAD_PLANES / RL8_WGRAD_GATE_PLANES=bf16 select, and what serves operands without a bound (rl8_mlp_wgrad_split_f32,
// rl8_mlp_wgrad_split_strided_f32).
// The forward and data-gradient kernels of this scheme (round 1-2: mlp_tower_{forward,backward}_split_kernel) were
// removed in round 3: every width they served runs the rows-per-wave kernels of mlp_rows_kernels.hip /
// mlp_f16_kernels.hip, everything else the fp32-MFMA generation of mlp_kernels.hip.
#include <type_traits>
#include "split_tile.hip.h"

namespace rl8 {

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
  // The data-driven choice of planes (round 4): bounds[kGuardFlag] is what wgrad_tail_kernel decided for this call
  // (0: fp16 planes, 1: the exact bf16 planes); a kernel launched with `guard` set leaves at once unless the flag
  // equals `guard_want` -- both generations are launched, one of them runs, no host round trip.
  const uint32_t *guard = nullptr;
  int guard_want = 0;
};

// Words of the 64 behind the slabs of the weight-gradient workspace: [0 .. 3] max |dOut column|, [4 .. 11] max |x column|
// (wgrad_bounds_kernel; round 6: eight inputs, five until then), then the guard's sample sums, its ticket, its decision,
// the entries it found at the maximum and the NaNs it saw (words 12 .. 18) -- the first 32 words are zeroed per call -- and two
// counters that only ever grow (calls that consulted the guard, calls it sent to the bf16 planes), zeroed by whoever
// allocates the workspace if they are to be read.
constexpr int kGuardSum = 12 /* two words: a uint64, 8-byte aligned */, kGuardNonzero = 14, kGuardTicket = 15, kGuardFlag = 16;
constexpr int kGuardCalls = 32, kGuardFires = 33;
// What the guard measures ("bound / mean |term|"): how far below the largest |dOut| of the call the
// non-zero entries sit ON AVERAGE.  The planes are scaled by the column's bound, which carries max |dOut| as a factor;
// a term 2^-a below the bound keeps min(22, 39 - a) bits (fp16 planes; 50 - a with the wide low plane of the gate
// kernel).  When max / mean is moderate -- PPO's gradients at 2^25 rows: 2^1 .. 2^5, tools/diag/wgrad_planes_real_ppo.py
// -- an entry of the sum is dominated by terms that keep all their bits, however long the tail of small rows is (those
// add an absolute error of 2^-39 of the bound each: with R = max / mean the sum is off by at most 2^-39 R of its own
// size).  One row 10^6 above the rest turns that around: every other term is "small", R = 2^20 and more.  The call goes
// to the exact bf16 planes when R > 2^kGuardRatio.  (Integer arithmetic on 24-bit fractions of the maximum: the decision
// does not depend on the order of the atomics.)
constexpr int kGuardRatio = 12;

__device__ __forceinline__ bool guard_says_leave(const WgradFusedArgs &fused) {
  return fused.guard != nullptr && (int)(fused.guard[kGuardFlag] != 0u) != fused.guard_want;
}

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
  // groups = 4: FOUR products over the same h rows in one launch -- the LSTM's four gates, dZ of gate q = columns
  // [256 q, 256 q + 256) of the rows at `dz` (group_dz_offset = 256 floats) -- so that h is read from HBM once: workgroup
  // b works for gate (b >> 3) & 3 as number (b & 7) | (b >> 5 << 3) of its gate's gridDim.x / 4 workgroups; the four that
  // share a number sit on one XCD (b mod 8) and walk the same rows at the same time, three of them out of L2.  Slabs and
  // column-sum rows are numbered gate-major (gate q: [q gridDim.x / 4, (q + 1) gridDim.x / 4)).  gridDim.x a multiple of 32.
  int groups = 1, group_dz_offset = 0;
};

// (The fp16-plane form of this product -- both operands two planes, each scaled by a power of two per column of the
// OUTPUT it indexes so that the factors leave the sum over samples: three plane products instead of six -- is
// mlp_wgrad_fused16_kernel / mlp_wgrad_loadh16_kernel below.)
template <int DIN, int FUSED = 0, bool LOADH = false>
__global__ __launch_bounds__(kWsThreads, 1) void mlp_wgrad_split_kernel(
    const float *__restrict__ dz2, const float *__restrict__ x, const float *__restrict__ w1,
    const float *__restrict__ b1, int64_t m, int d_in_rt, float *__restrict__ slabs, WgradFusedArgs fused,
    WgradOperands ops) {
  static_assert(!LOADH || (DIN > 0 && FUSED == 0), "the two-operand mode: compiled input widths, no head fusion");
  if (guard_says_leave(fused)) return;  // (uniform: the other generation of planes forms this call's sums)
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

  const int64_t chunks = (m + kWsChunk - 1) / kWsChunk;
  // (two-operand mode with gate groups: see WgradOperands::groups)
  const bool grouped = LOADH && ops.groups == 4;
  const int gate_q = grouped ? ((int)blockIdx.x >> 3) & 3 : 0;
  const int64_t bid = grouped ? ((blockIdx.x & 7) | ((blockIdx.x >> 5) << 3)) : blockIdx.x;  // this workgroup's number in its group
  const int64_t stride = grouped ? gridDim.x / 4 : gridDim.x;
  const int64_t out_id = grouped ? gate_q * stride + bid : blockIdx.x;                          // its slab / column-sum row
  const float *dzp = dz2 + (grouped ? gate_q * ops.group_dz_offset : 0);
  const int64_t mine = (chunks - bid + stride - 1) / stride;  // >= 1 (workgroups per group <= chunks)

  // dZ2 of chunk number n of this workgroup: this thread's column, its eight samples.
  // Chunks past the end (and samples past m) read as zero through the descriptor.
  auto load_dz = [&](float (&dst)[8], int64_t n) {
    const int64_t chunk = bid + n * stride;
    const int64_t left = m - chunk * kWsChunk;
    const int rows = left <= 0 ? 0 : left < kWsChunk ? (int)left : kWsChunk;
    const __amdgpu_buffer_rsrc_t rsrc = buffer_rsrc(rows > 0 ? dzp + chunk * kWsChunk * dz_pitch : dzp,
                                                    rows > 0 ? ((rows - 1) * dz_pitch + kHidden) * 4 : 0);
#pragma unroll
    for (int e = 0; e < 8; ++e) dst[e] = buffer_load_f32(rsrc, col * 4, (8 * kh + e) * (dz_pitch * 4));
  };
  [[maybe_unused]] auto load_h = [&](float (&dst)[8], int64_t n) {  // LOADH: the same for the h1 operand
    const int64_t chunk = bid + n * stride;
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
  auto lds_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
  // Observations and dOut of the eight samples a wave produces are wave-uniform: with
  // DIN known they sit in scalar registers, fetched (kIn + kOut loads of eight floats)
  // in front of the PREVIOUS step's barrier, so that inside a step nothing but
  // ds_reads is in flight on lgkmcnt and the first MFMAs can go as soon as THEIR
  // fragments are in (wait_lds<N>), not after all twelve.
  constexpr bool kScalars = DIN > 0;  // (two-operand mode: the observations feed the column sums only)
  constexpr int kXq = kScalars ? kIn : 1, kDq = (kScalars && FUSED > 0) ? kOut : 1;
  [[maybe_unused]] f32x8 xq[kXq], dq[kDq];
  auto row0_of = [&](int64_t n) { return (bid + n * stride) * kWsChunk + 8 * kh; };
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
      if constexpr (kIn > 5) xq[5] = scalar_buffer_load_x8<160>(rx);
      if constexpr (kIn > 6) xq[6] = scalar_buffer_load_x8<192>(rx);
      if constexpr (FUSED > 0) {
        const u32x4 rd = scalar_rsrc(fused.dout + row0 * kOut, rows * kOut * 4);
        dq[0] = scalar_buffer_load_x8<0>(rd);
        if constexpr (kOut > 1) dq[1] = scalar_buffer_load_x8<32>(rd);
        if constexpr (kOut > 2) dq[2] = scalar_buffer_load_x8<64>(rd);
        if constexpr (kOut > 3) dq[3] = scalar_buffer_load_x8<96>(rd);
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
      split8(dzv, pa);
      split8(hv, pb);
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
      split8(dz, pa);
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
    split8(h, pb);
  };
  auto write_planes = [&](int stage, const u32x4 (&pa)[3], const u32x4 (&pb)[3]) {
    const unsigned addr = p_write + stage * kWsStageBytes;
    lds_write_b128<0>(addr, pa[0]);
    lds_write_b128<kWsPlane>(addr, pa[1]);
    lds_write_b128<2 * kWsPlane>(addr, pa[2]);
    lds_write_b128<kWsOperandBytes>(addr, pb[0]);
    lds_write_b128<kWsOperandBytes + kWsPlane>(addr, pb[1]);
    lds_write_b128<kWsOperandBytes + 2 * kWsPlane>(addr, pb[2]);
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

  float *slab = slabs + out_id * kHidden * kHidden;
#pragma unroll
  for (int ja = 0; ja < 2; ++ja)
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int j = 64 * wj + 32 * ja + (r & 3) + 8 * (r >> 2) + 4 * hh;
        const int i = 128 * wi + 32 * t + l32;
        slab[j * kHidden + i] = acc[ja][t][r];
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
        float *row = ops.colsums + out_id * (kHidden * (kIn + 1));
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

// (The fp16-plane form -- the second operand dOut[s] * h1[s][i] as TWO fp16 planes of its value times a power of two per
// COLUMN i, chosen from a bound on the column: max_s |dOut[s]| * (|b1[i]| + sum_c max_s |x[s][c]| |w1[i][c]|), placed below
// 2^14; two plane products per 16 samples instead of three -- is mlp_wgrad_gate16_kernel below.)
template <int DIN, bool PAIR = false, bool BITS = false>
__global__ __launch_bounds__(kWsThreads, 1) void mlp_wgrad_gate_kernel(
    const float *__restrict__ h2, const float *__restrict__ x, const float *__restrict__ w1,
    const float *__restrict__ b1, int64_t m, float *__restrict__ slabs, WgradFusedArgs fused) {
  static_assert(DIN > 0, "compiled input widths only");
  if (guard_says_leave(fused)) return;  // (uniform: the other generation of planes forms this call's sums)
  constexpr int kIn = DIN, d_in = DIN;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const unsigned lds0 = lds_offset(smem);
  const int tid = threadIdx.x, lane = tid & 63, l32 = lane & 31, hh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wj = wave >> 1, wi = wave & 1;  // j-tiles {2wj, 2wj+1}, i-tiles {4wi .. 4wi+3}
  const int col = tid & 255;                // producer: column (j of the gate and i of h1) ...
  const int kh = wave >> 2;                 // ... and which eight samples of the chunk (wave-uniform)

  const unsigned a_read = lds0 + (hh * kHidden + 64 * wj + l32) * 16;
  const unsigned b_read = lds0 + kWgOperandA + (hh * kHidden + 128 * wi + l32) * 16;
  const unsigned p_write = lds0 + (kh * kHidden + col) * 16;

  float w1r[kIn];
#pragma unroll
  for (int c = 0; c < kIn; ++c) w1r[c] = w1[col * d_in + c];
  const float b1r = b1[col];
  float gsum = 0.0f, dw3a = 0.0f;  // sum_s G dOut (db2 / W3) and dW3 of this thread's column and sample half

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
    if constexpr (kIn > 5) xq[5] = scalar_buffer_load_x8<160>(rx);
    if constexpr (kIn > 6) xq[6] = scalar_buffer_load_x8<192>(rx);
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
          g[e >> 1] = (m0 & 0x00003f80u) | (m1 & 0x3f800000u);  // bf16 1.0 / 0.0
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
    u32x4(&B0)[4] = *(P == 0 ? &f.bh : &f.bm);
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
        slab[j * kHidden + i] = acc[ja][t][r] * w3j;
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

// ---- the gate-bits weight gradient on SIXTEEN waves (round 4) -----------------------------------------------------------
// The kernel above spends four fifths of its time on the operand side (the build without matrix work: 224 of 279 us per
// 2^20 rows) with no unit busy -- VALU 37 %, LDS 46 %, matrix pipe 59 %: a latency chain (load, produce, write, barrier,
// read, multiply) behind two waves per SIMD.  Same arithmetic, same slabs, same partial rows here, with 1024 threads:
// four waves per SIMD, a wave owns 2 x 2 tiles of 32 x 32 (64 accumulator registers instead of 128), a thread produces
// FOUR samples of its column per 16-sample step instead of eight, one barrier per step over a four-stage ring, and the
// waits are the compiler's.  BITS + F16 planes (wide low plane) only: what rl8_mlp_wgrad_gate_bits_f32 runs.
constexpr int kW16Threads = 1024;
constexpr int kW16StageBytes = 3 * 2 * kHidden * 16;  // gate plane | hi plane | lo plane, each [sample half][column] x 16 B
constexpr int kW16Stages = 4;

template <int DIN, bool PAIR>
__global__ __launch_bounds__(kW16Threads, 1) void mlp_wgrad_gate16_kernel(
    const float *__restrict__ x, const float *__restrict__ w1, const float *__restrict__ b1, int64_t m,
    float *__restrict__ slabs, WgradFusedArgs fused) {
  if (guard_says_leave(fused)) return;
  constexpr int kIn = DIN;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float *inv_scales = reinterpret_cast<float *>(smem + kW16Stages * kW16StageBytes);  // [256]
  const int tid = threadIdx.x, lane = tid & 63, l32 = lane & 31, hh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wj = wave >> 2, wi = wave & 3;       // consumer: j-tiles {2 wj, 2 wj + 1}, i-tiles {2 wi, 2 wi + 1}
  const int col = tid & 255, q = wave >> 2;      // producer: column, and samples 4 q .. 4 q + 3 of a chunk (wave-uniform)

  float w1r[kIn];
#pragma unroll
  for (int c = 0; c < kIn; ++c) w1r[c] = w1[col * kIn + c];
  float b1r = b1[col];
  {
    float hb = __builtin_fabsf(b1r);
#pragma unroll
    for (int c = 0; c < kIn; ++c) hb = __builtin_fmaf(__uint_as_float(fused.bounds[4 + c]), __builtin_fabsf(w1r[c]), hb);
    const int e = f16_bound_exponent(__uint_as_float(fused.bounds[0]) * hb * 1.0001f);
    const float col_scale = __builtin_amdgcn_ldexpf(1.0f, kF16Top - e);
    if (q == 0) inv_scales[col] = __builtin_amdgcn_ldexpf(1.0f, e - kF16Top);
#pragma unroll
    for (int c = 0; c < kIn; ++c) w1r[c] *= col_scale;  // relu(x . (s w1) + s b1) = s h1 exactly
    b1r *= col_scale;
  }
  const float k2048 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(0x45000000));
  float gsum = 0.0f;

  const int64_t chunks = (m + kWsChunk - 1) / kWsChunk;
  const int64_t stride = gridDim.x;
  const int64_t mine = (chunks - blockIdx.x + stride - 1) / stride;  // >= 1

  // what a thread holds of a chunk before it produces it: its column's gate word of its four rows, and (wave-uniform)
  // the rows' observations and dOut
  struct Raw {
    uint32_t g[4];
    f32x8 dv, xv[(4 * kIn + 7) / 8];  // scalar registers: requested by issue(), usable behind land()
  };
  // (32-bit offsets through buffer descriptors that end at row m: rows past the end read as zero, no branches.  No wait
  // here: the loads fly under the step's production and products; land() is the wait and the compiler's fence)
  // d_in >= 6 (round 6): the scalar loads are made in land(), directly in front of their wait.  Requested a chunk ahead
  // like the narrower widths' -- two sets of up to 40 scalar registers in flight across a whole consume() + produce() --
  // the register allocator ran out and parked destinations of loads still in flight in vector lanes (v_writelane of stale
  // data: tools/check_inflight_regs.py finds it).  One set, alive from its wait to the end of produce(); the wait costs a
  // scalar-cache round trip per chunk, behind which the SIMD's other three waves work.
  constexpr bool kLateScalars = kIn >= 6;
  auto issue = [&](Raw &r, int64_t n) {
    const int64_t row0 = (blockIdx.x + n * stride) * kWsChunk + 4 * q;
    const int64_t left = m - row0;
    const int rows = left <= 0 ? 0 : left < 4 ? (int)left : 4;
    const int64_t at = rows > 0 ? row0 : 0;
    const __amdgpu_buffer_rsrc_t g = buffer_rsrc(fused.gate2 + at * 8, rows * 32);
    if constexpr (kLateScalars) {
#pragma unroll
      for (int e = 0; e < 4; ++e) r.g[e] = __float_as_uint(buffer_load_f32(g, (col >> 5) * 4, e * 32));
    } else {  // (the narrower widths: text and code as in rounds 4-5)
      const u32x4 rx = scalar_rsrc(x + at * kIn, rows * kIn * 4);
      const u32x4 rd = scalar_rsrc(fused.dout + at * (PAIR ? 2 : 1), rows * (PAIR ? 8 : 4));
#pragma unroll
      for (int e = 0; e < 4; ++e) r.g[e] = __float_as_uint(buffer_load_f32(g, (col >> 5) * 4, e * 32));
      r.dv = scalar_buffer_load_x8<0>(rd);
      r.xv[0] = scalar_buffer_load_x8<0>(rx);
      if constexpr (4 * kIn > 8) r.xv[1] = scalar_buffer_load_x8<32>(rx);
      if constexpr (4 * kIn > 16) r.xv[2] = scalar_buffer_load_x8<64>(rx);
    }
  };
  auto land = [&](Raw &r, [[maybe_unused]] int64_t n) {  // n: the chunk issue(r, n) was called for
    if constexpr (kLateScalars) {
      const int64_t row0 = (blockIdx.x + n * stride) * kWsChunk + 4 * q;
      const int64_t left = m - row0;
      const int rows = left <= 0 ? 0 : left < 4 ? (int)left : 4;
      const int64_t at = rows > 0 ? row0 : 0;
      const u32x4 rx = scalar_rsrc(x + at * kIn, rows * kIn * 4);
      const u32x4 rd = scalar_rsrc(fused.dout + at * (PAIR ? 2 : 1), rows * (PAIR ? 8 : 4));
      r.dv = scalar_buffer_load_x8<0>(rd);
      r.xv[0] = scalar_buffer_load_x8<0>(rx);
      r.xv[1] = scalar_buffer_load_x8<32>(rx);
      r.xv[2] = scalar_buffer_load_x8<64>(rx);
      if constexpr (4 * kIn > 24) r.xv[3] = scalar_buffer_load_x8<96>(rx);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    scalar_tie(r.dv);
#pragma unroll
    for (int i = 0; i < (4 * kIn + 7) / 8; ++i) scalar_tie(r.xv[i]);
  };
  auto d_of = [&](const Raw &r, int e) { return r.dv[PAIR ? 2 * e : e]; };
  auto x_of = [&](const Raw &r, int e, int c) { return r.xv[(e * kIn + c) >> 3][(e * kIn + c) & 7]; };
  auto produce = [&](const Raw &r, int stage) {
    unsigned char *base = smem + stage * kW16StageBytes + ((q >> 1) * kHidden + col) * 16 + (q & 1) * 8;
    uint32_t gw[2], hi[2], lo[2];
    float t[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float v = b1r;
#pragma unroll
      for (int c = 0; c < kIn; ++c) v = __builtin_fmaf(x_of(r, e, c), w1r[c], v);
      t[e] = relu1(v);
    }
#pragma unroll
    for (int e = 0; e < 4; e += 2) {
      const uint32_t m0 = (uint32_t)__builtin_amdgcn_sbfe((int)r.g[e], (unsigned)(col & 31), 1u);
      const uint32_t m1 = (uint32_t)__builtin_amdgcn_sbfe((int)r.g[e + 1], (unsigned)(col & 31), 1u);
      gw[e >> 1] = (m0 & 0x00003c00u) | (m1 & 0x3c000000u);
      gsum += __uint_as_float(m0 & __float_as_uint(d_of(r, e)));
      gsum += __uint_as_float(m1 & __float_as_uint(d_of(r, e + 1)));
      f16_pair_product_wide(t[e], t[e + 1], d_of(r, e), d_of(r, e + 1), k2048, hi[e >> 1], lo[e >> 1]);
    }
    *reinterpret_cast<u32x2 *>(base) = u32x2{gw[0], gw[1]};
    *reinterpret_cast<u32x2 *>(base + 2 * kHidden * 16) = u32x2{hi[0], hi[1]};
    *reinterpret_cast<u32x2 *>(base + 4 * kHidden * 16) = u32x2{lo[0], lo[1]};
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
  auto consume = [&](int stage) {
    const unsigned char *base = smem + stage * kW16StageBytes;
    const unsigned char *ap = base + (hh * kHidden + 64 * wj + l32) * 16;
    const unsigned char *bp = base + 2 * kHidden * 16 + (hh * kHidden + 64 * wi + l32) * 16;
    u32x4 g[2], gl[2], bh[2], bl[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      g[t] = *reinterpret_cast<const u32x4 *>(ap + t * 512);
      bh[t] = *reinterpret_cast<const u32x4 *>(bp + t * 512);
      bl[t] = *reinterpret_cast<const u32x4 *>(bp + 2 * kHidden * 16 + t * 512);
    }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) gl[t][r] = g[t][r] & kF16GateLowMask;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, g[a]), __builtin_bit_cast(half8, bh[b]),
                                                          acc[a][b], 0, 0, 0);
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, gl[a]), __builtin_bit_cast(half8, bl[b]),
                                                          acc[a][b], 0, 0, 0);
      }
  };

  // Ring: chunk n lives in stage n % 4.  Step n: load chunk n + 3, produce chunk n + 2 (loaded a step ago), consume chunk n
  // (produced two steps ago), barrier.  Stage (n + 2) % 4 was last read in step n - 2: two barriers back.
  Raw ra, rb;
  issue(ra, 0);
  land(ra, 0);
  produce(ra, 0);
  issue(ra, 1);
  land(ra, 1);
  produce(ra, 1);
  issue(ra, 2);
  land(ra, 2);
  __syncthreads();
  const int64_t steps = (mine + 3) & ~(int64_t)3;  // (a multiple of four: chunks past the end are all zero)
#pragma unroll 1
  for (int64_t n = 0; n < steps; n += 4) {
    // (one barrier per TWO steps: stages 2, 3 are written while 0, 1 are read, and the other way round)
    issue(rb, n + 3);
    consume(0);
    produce(ra, 2);
    land(rb, n + 3);
    issue(ra, n + 4);
    consume(1);
    produce(rb, 3);
    land(ra, n + 4);
    __syncthreads();
    issue(rb, n + 5);
    consume(2);
    produce(ra, 0);
    land(rb, n + 5);
    issue(ra, n + 6);
    consume(3);
    produce(rb, 1);
    land(ra, n + 6);
    __syncthreads();
  }

  float *slab = slabs + (int64_t)blockIdx.x * kHidden * kHidden;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int j = 64 * wj + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * hh;
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int i = 64 * wi + 32 * b + l32;
        slab[j * kHidden + i] = acc[a][b][r] * inv_scales[i];
      }
    }

  // head sums: the four sample quarters of a column folded in a fixed order; the partial row as the eight-wave kernel writes it
  float *red = reinterpret_cast<float *>(smem);  // [4][256]
  __syncthreads();
  red[q * kHidden + col] = gsum;
  __syncthreads();
  float *row = fused.partials + (int64_t)blockIdx.x * fused.partial_stride;
  constexpr int kOut = PAIR ? 2 : 1;
  constexpr int off_db2 = kHidden * kIn + kHidden, off_dw3 = off_db2 + kHidden, off_db3 = off_dw3 + kOut * kHidden;
  const bool more = fused.accumulate != 0;
  if (q == 0) {
    const float total = ((red[col] + red[kHidden + col]) + red[2 * kHidden + col]) + red[3 * kHidden + col];
    const float w3e = PAIR ? fused.w3[col] - fused.w3[kHidden + col] : fused.w3[col];
    const float sum_b2 = total * w3e, sum_w3 = total * fused.b2[col];
    row[off_db2 + col] = more ? row[off_db2 + col] + sum_b2 : sum_b2;
    row[off_dw3 + col] = more ? row[off_dw3 + col] + sum_w3 : sum_w3;
    if constexpr (PAIR) row[off_dw3 + kHidden + col] = more ? row[off_dw3 + kHidden + col] - sum_w3 : -sum_w3;
    if (col < kOut && !more) row[off_db3 + col] = 0.0f;
  }
  if (!more && (int)blockIdx.x >= fused.other_rows)
    for (int idx = tid; idx < kHidden * kIn + kHidden; idx += kW16Threads) row[idx] = 0.0f;
}

template <int DIN, bool PAIR>
static int launch_wgrad_gate16(int grid, hipStream_t s, const float *x, const float *w1, const float *b1, int64_t m,
                               float *slabs, WgradFusedArgs fused) {
  static LdsOptIn opt;
  if (const int e = allow_dynamic_lds(opt, reinterpret_cast<const void *>(&mlp_wgrad_gate16_kernel<DIN, PAIR>), 160 * 1024)) return e;
  mlp_wgrad_gate16_kernel<DIN, PAIR><<<grid, kW16Threads, kW16Stages * kW16StageBytes + kHidden * 4, s>>>(x, w1, b1, m, slabs, fused);
  return launch_status();
}

// ---- the general fused weight gradient on sixteen waves (round 4), same recipe ---------------------------------------------
// The fused product of mlp_wgrad_split_kernel (dZ2 formed from h2, dOut and W3; h1 recomputed) on fp16 planes: BOTH
// operands two planes, each scaled by a power of two per column of the OUTPUT it indexes -- dZ2[s][j] by 2^a(j) from
// sum_q max|dOut_q| |W3[q][j]|, h1[s][i] by 2^b(i) from |b1[i]| + sum_c max|x_c| |w1[i][c]| -- so the factors leave the sum
// over samples; dZ2's low plane wide; THREE plane products per 16 samples instead of six.  1024 threads: a wave owns
// 2 x 2 tiles, a thread produces four samples of its column as j of dZ2 AND as i of h1, four LDS stages of four planes,
// one barrier per two steps.  Same slabs and partial rows as the eight-wave kernel.
constexpr int kW16FusedStageBytes = 4 * 2 * kHidden * 16;  // dZ2 hi | dZ2 lo (wide) | h1 hi | h1 lo

template <int DIN, int NOUT>
__global__ __launch_bounds__(kW16Threads, 1) void mlp_wgrad_fused16_kernel(
    const float *__restrict__ h2, const float *__restrict__ x, const float *__restrict__ w1, const float *__restrict__ b1,
    int64_t m, float *__restrict__ slabs, WgradFusedArgs fused) {
  if (guard_says_leave(fused)) return;
  constexpr int kIn = DIN, kOut = NOUT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float *inv_a = reinterpret_cast<float *>(smem + kW16Stages * kW16FusedStageBytes), *inv_b = inv_a + kHidden;
  const int tid = threadIdx.x, lane = tid & 63, l32 = lane & 31, hh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wj = wave >> 2, wi = wave & 3;
  const int col = tid & 255, q = wave >> 2;

  float w1r[kIn], w3r[kOut], dw3a[kOut], db2a = 0.0f;
#pragma unroll
  for (int c = 0; c < kIn; ++c) w1r[c] = w1[col * kIn + c];
  const float b1r = b1[col];
#pragma unroll
  for (int o = 0; o < kOut; ++o) {
    w3r[o] = fused.w3[o * kHidden + col];
    dw3a[o] = 0.0f;
  }
  float scale_a, scale_b;
  {
    float za = 0.0f, hb = __builtin_fabsf(b1r);
#pragma unroll
    for (int o = 0; o < kOut; ++o) za = __builtin_fmaf(__uint_as_float(fused.bounds[o]), __builtin_fabsf(w3r[o]), za);
#pragma unroll
    for (int c = 0; c < kIn; ++c) hb = __builtin_fmaf(__uint_as_float(fused.bounds[4 + c]), __builtin_fabsf(w1r[c]), hb);
    const int ea = f16_bound_exponent(za * 1.0001f), eb = f16_bound_exponent(hb * 1.0001f);
    scale_a = __builtin_amdgcn_ldexpf(1.0f, kF16Top - ea);
    scale_b = __builtin_amdgcn_ldexpf(1.0f, kF16Top - eb);
    if (q == 0) {
      inv_a[col] = __builtin_amdgcn_ldexpf(1.0f, ea - kF16Top);
      inv_b[col] = __builtin_amdgcn_ldexpf(1.0f, eb - kF16Top);
    }
  }
  const float k2048 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(0x45000000));
  const uint32_t k_low = (uint32_t)__builtin_amdgcn_readfirstlane((int)kF16GateLowMask);

  const int64_t chunks = (m + kWsChunk - 1) / kWsChunk;
  const int64_t stride = gridDim.x;
  const int64_t mine = (chunks - blockIdx.x + stride - 1) / stride;

  constexpr int kDv = (4 * kOut + 7) / 8, kXv = (4 * kIn + 7) / 8;
  struct Raw {
    float h[4];               // h2 of this thread's column, its four rows
    f32x8 dv[kDv], xv[kXv];   // scalar registers: requested by issue(), usable behind land()
  };
  constexpr bool kLateScalars = kIn >= 6;  // (see mlp_wgrad_gate16_kernel: the scalar loads in land(), one set alive)
  auto issue = [&](Raw &r, int64_t n) {
    const int64_t row0 = (blockIdx.x + n * stride) * kWsChunk + 4 * q;
    const int64_t left = m - row0;
    const int rows = left <= 0 ? 0 : left < 4 ? (int)left : 4;
    const int64_t at = rows > 0 ? row0 : 0;
    const __amdgpu_buffer_rsrc_t hr = buffer_rsrc(h2 + at * kHidden, rows * kHidden * 4);
    if constexpr (kLateScalars) {
#pragma unroll
      for (int e = 0; e < 4; ++e) r.h[e] = buffer_load_f32(hr, col * 4, e * (kHidden * 4));
    } else {  // (the narrower widths: text and code as in rounds 4-5)
      const u32x4 rx = scalar_rsrc(x + at * kIn, rows * kIn * 4);
      const u32x4 rd = scalar_rsrc(fused.dout + at * kOut, rows * kOut * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) r.h[e] = buffer_load_f32(hr, col * 4, e * (kHidden * 4));
      r.dv[0] = scalar_buffer_load_x8<0>(rd);
      if constexpr (kDv > 1) r.dv[1] = scalar_buffer_load_x8<32>(rd);
      r.xv[0] = scalar_buffer_load_x8<0>(rx);
      if constexpr (kXv > 1) r.xv[1] = scalar_buffer_load_x8<32>(rx);
      if constexpr (kXv > 2) r.xv[2] = scalar_buffer_load_x8<64>(rx);
    }
  };
  auto land = [&](Raw &r, [[maybe_unused]] int64_t n) {  // n: the chunk issue(r, n) was called for
    if constexpr (kLateScalars) {
      const int64_t row0 = (blockIdx.x + n * stride) * kWsChunk + 4 * q;
      const int64_t left = m - row0;
      const int rows = left <= 0 ? 0 : left < 4 ? (int)left : 4;
      const int64_t at = rows > 0 ? row0 : 0;
      const u32x4 rx = scalar_rsrc(x + at * kIn, rows * kIn * 4);
      const u32x4 rd = scalar_rsrc(fused.dout + at * kOut, rows * kOut * 4);
      r.dv[0] = scalar_buffer_load_x8<0>(rd);
      if constexpr (kDv > 1) r.dv[1] = scalar_buffer_load_x8<32>(rd);
      r.xv[0] = scalar_buffer_load_x8<0>(rx);
      r.xv[1] = scalar_buffer_load_x8<32>(rx);
      r.xv[2] = scalar_buffer_load_x8<64>(rx);
      if constexpr (kXv > 3) r.xv[3] = scalar_buffer_load_x8<96>(rx);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < kDv; ++i) scalar_tie(r.dv[i]);
#pragma unroll
    for (int i = 0; i < kXv; ++i) scalar_tie(r.xv[i]);
  };
  auto produce = [&](const Raw &r, int stage) {
    unsigned char *base = smem + stage * kW16FusedStageBytes + ((q >> 1) * kHidden + col) * 16 + (q & 1) * 8;
    float dz[4], h[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float g = 0.0f;
#pragma unroll
      for (int o = 0; o < kOut; ++o) {
        const float d = r.dv[(e * kOut + o) >> 3][(e * kOut + o) & 7];
        g = __builtin_fmaf(d, w3r[o], g);
        dw3a[o] = __builtin_fmaf(d, r.h[e], dw3a[o]);
      }
      dz[e] = r.h[e] > 0.0f ? g : 0.0f;
      db2a += dz[e];
      float v = b1r;
#pragma unroll
      for (int c = 0; c < kIn; ++c) v = __builtin_fmaf(r.xv[(e * kIn + c) >> 3][(e * kIn + c) & 7], w1r[c], v);
      h[e] = relu1(v);
    }
    uint32_t ah[2], al[2], bh[2], bl[2];
#pragma unroll
    for (int e = 0; e < 4; e += 2) {
      f16_pair_scaled_wide(dz[e], dz[e + 1], scale_a, k2048, ah[e >> 1], al[e >> 1]);
      f16_pair_scaled(h[e], h[e + 1], scale_b, bh[e >> 1], bl[e >> 1]);
    }
    *reinterpret_cast<u32x2 *>(base) = u32x2{ah[0], ah[1]};
    *reinterpret_cast<u32x2 *>(base + 2 * kHidden * 16) = u32x2{al[0], al[1]};
    *reinterpret_cast<u32x2 *>(base + 4 * kHidden * 16) = u32x2{bh[0], bh[1]};
    *reinterpret_cast<u32x2 *>(base + 6 * kHidden * 16) = u32x2{bl[0], bl[1]};
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
  auto consume = [&](int stage) {
    const unsigned char *base = smem + stage * kW16FusedStageBytes;
    const unsigned char *ap = base + (hh * kHidden + 64 * wj + l32) * 16;
    const unsigned char *bp = base + 4 * kHidden * 16 + (hh * kHidden + 64 * wi + l32) * 16;
    // hi x hi and hi x lo first; then the h1 hi fragments become 2^-11 x themselves in place (the wide low plane's
    // partner) for lo x hi: no third set of B registers
    u32x4 ah[2], bh[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      ah[t] = *reinterpret_cast<const u32x4 *>(ap + t * 512);
      bh[t] = *reinterpret_cast<const u32x4 *>(bp + t * 512);
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, ah[a]), __builtin_bit_cast(half8, bh[b]), acc[a][b], 0, 0, 0);
    {
      u32x4 bl[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) bl[t] = *reinterpret_cast<const u32x4 *>(bp + 2 * kHidden * 16 + t * 512);
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, ah[a]), __builtin_bit_cast(half8, bl[b]), acc[a][b], 0, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      ah[t] = *reinterpret_cast<const u32x4 *>(ap + 2 * kHidden * 16 + t * 512);  // dZ2's wide low plane, into the hi plane's registers
#pragma unroll
      for (int r = 0; r < 4; ++r) bh[t][r] = f16_pair_times(bh[t][r], k_low);
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, ah[a]), __builtin_bit_cast(half8, bh[b]), acc[a][b], 0, 0, 0);
  };

  Raw ra, rb;
  issue(ra, 0);
  land(ra, 0);
  produce(ra, 0);
  issue(ra, 1);
  land(ra, 1);
  produce(ra, 1);
  issue(ra, 2);
  land(ra, 2);
  __syncthreads();
  const int64_t steps = (mine + 3) & ~(int64_t)3;
#pragma unroll 1
  for (int64_t n = 0; n < steps; n += 4) {
    issue(rb, n + 3);
    consume(0);
    produce(ra, 2);
    land(rb, n + 3);
    issue(ra, n + 4);
    consume(1);
    produce(rb, 3);
    land(ra, n + 4);
    __syncthreads();
    issue(rb, n + 5);
    consume(2);
    produce(ra, 0);
    land(rb, n + 5);
    issue(ra, n + 6);
    consume(3);
    produce(rb, 1);
    land(ra, n + 6);
    __syncthreads();
  }

  float *slab = slabs + (int64_t)blockIdx.x * kHidden * kHidden;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int j = 64 * wj + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * hh;
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int i = 64 * wi + 32 * b + l32;
        slab[j * kHidden + i] = acc[a][b][r] * (inv_a[j] * inv_b[i]);
      }
    }

  float *red = reinterpret_cast<float *>(smem);  // [4][256][1 + kOut]
  __syncthreads();
  red[(q * kHidden + col) * (1 + kOut)] = db2a;
#pragma unroll
  for (int o = 0; o < kOut; ++o) red[(q * kHidden + col) * (1 + kOut) + 1 + o] = dw3a[o];
  __syncthreads();
  float *row = fused.partials + (int64_t)blockIdx.x * fused.partial_stride;
  constexpr int off_db2 = kHidden * kIn + kHidden, off_dw3 = off_db2 + kHidden, off_db3 = off_dw3 + kOut * kHidden;
  const bool more = fused.accumulate != 0;
  if (q == 0) {
    auto total = [&](int k) {
      return ((red[col * (1 + kOut) + k] + red[(kHidden + col) * (1 + kOut) + k]) + red[(2 * kHidden + col) * (1 + kOut) + k]) +
             red[(3 * kHidden + col) * (1 + kOut) + k];
    };
    const float sum_b2 = total(0);
    row[off_db2 + col] = more ? row[off_db2 + col] + sum_b2 : sum_b2;
#pragma unroll
    for (int o = 0; o < kOut; ++o) {
      const float sum_w3 = total(1 + o);
      row[off_dw3 + o * kHidden + col] = more ? row[off_dw3 + o * kHidden + col] + sum_w3 : sum_w3;
    }
    if (col < kOut && !more) row[off_db3 + col] = 0.0f;
  }
  if (!more && (int)blockIdx.x >= fused.other_rows)
    for (int idx = tid; idx < kHidden * kIn + kHidden; idx += kW16Threads) row[idx] = 0.0f;
}

template <int DIN, int NOUT>
static int launch_wgrad_fused16(int grid, hipStream_t s, const float *h2, const float *x, const float *w1, const float *b1,
                                int64_t m, float *slabs, WgradFusedArgs fused) {
  static LdsOptIn opt;
  if (const int e = allow_dynamic_lds(opt, reinterpret_cast<const void *>(&mlp_wgrad_fused16_kernel<DIN, NOUT>), 160 * 1024)) return e;
  mlp_wgrad_fused16_kernel<DIN, NOUT><<<grid, kW16Threads, kW16Stages * kW16FusedStageBytes + 2 * kHidden * 4, s>>>(h2, x, w1, b1, m, slabs, fused);
  return launch_status();
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
  static LdsOptIn lds_attr_set_0;
  if (const int e_lds_attr_set_0 = allow_dynamic_lds(lds_attr_set_0, reinterpret_cast<const void *>(&mlp_wgrad_split_kernel<DIN>), 160 * 1024)) return e_lds_attr_set_0;
  mlp_wgrad_split_kernel<DIN><<<grid, kWsThreads, 2 * kWsStageBytes, s>>>(dz2, x, w1, b1, m, d_in, slabs, WgradFusedArgs{},
                                                                           WgradOperands{});
  return launch_status();
}

template <int DIN, int NOUT>
static int launch_wgrad_fused(int grid, hipStream_t s, const float *h2, const float *x, const float *w1,
                              const float *b1, int64_t m, int d_in, float *slabs, WgradFusedArgs fused) {
  static LdsOptIn lds_attr_set_0;
  if (const int e_lds_attr_set_0 = allow_dynamic_lds(lds_attr_set_0, reinterpret_cast<const void *>(&mlp_wgrad_split_kernel<DIN, NOUT, false>), 160 * 1024)) return e_lds_attr_set_0;
  mlp_wgrad_split_kernel<DIN, NOUT, false><<<grid, kWsThreads, 2 * kWsStageBytes, s>>>(
      h2, x, w1, b1, m, d_in, slabs, fused, WgradOperands{});
  return launch_status();
}

template <int DIN, bool PAIR = false, bool BITS = false>
static int launch_wgrad_gate(int grid, hipStream_t s, const float *h2, const float *x, const float *w1,
                             const float *b1, int64_t m, float *slabs, WgradFusedArgs fused) {
  static LdsOptIn lds_attr_set_0;
  if (const int e_lds_attr_set_0 = allow_dynamic_lds(lds_attr_set_0, reinterpret_cast<const void *>(&mlp_wgrad_gate_kernel<DIN, PAIR, BITS>), 160 * 1024)) return e_lds_attr_set_0;
  mlp_wgrad_gate_kernel<DIN, PAIR, BITS><<<grid, kWsThreads, 4 * kWgStageBytes, s>>>(
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
  if (hipMemsetAsync(bounds, 0, 128, s) != hipSuccess) return nullptr;  // words 0 .. 31; the lifetime counters sit at 32, 33
  const int64_t want = m / (4 * kBlock);
  const int grid = (int)(want < 1 ? 1 : want > 4 * kCUs ? 4 * kCUs : want);  // (one atomic per block and column: 12 ns each)
  if (d_in == 1) wgrad_bounds_kernel<1, NOUT><<<grid, kBlock, 0, s>>>(dout, x, m, bounds);
  else if (d_in == 2) wgrad_bounds_kernel<2, NOUT><<<grid, kBlock, 0, s>>>(dout, x, m, bounds);
  else if (d_in == 3) wgrad_bounds_kernel<3, NOUT><<<grid, kBlock, 0, s>>>(dout, x, m, bounds);
  else if (d_in == 4) wgrad_bounds_kernel<4, NOUT><<<grid, kBlock, 0, s>>>(dout, x, m, bounds);
  else if (d_in == 5) wgrad_bounds_kernel<5, NOUT><<<grid, kBlock, 0, s>>>(dout, x, m, bounds);
  else if (d_in == 6) wgrad_bounds_kernel<6, NOUT><<<grid, kBlock, 0, s>>>(dout, x, m, bounds);
  else if (d_in == 7) wgrad_bounds_kernel<7, NOUT><<<grid, kBlock, 0, s>>>(dout, x, m, bounds);
  else return nullptr;
  return bounds;
}

// The guard of the fp16 planes: over a sample of dOut (one KiB in every `every`, every entry of it; every = 1 -- all of it
// -- for calls of up to 2^20 floats, 16 beyond) count the non-zero entries and add up |entry| / max as 24-bit fractions
// (max: the call's largest, bounds[0 .. 3], complete when this kernel starts); the last workgroup to arrive turns the
// sums into the call's flag and bumps the two lifetime counters.  Entries AT the maximum are counted apart and left out
// of both sums (with the maximum inside a small sample its own 2^24 kept "count * 2^12 > sum" from ever
// holding: what is asked is how far the OTHER entries sit below it); a NaN anywhere in the sample sends the call to the
// exact planes.
constexpr int kGuardTop = 17, kGuardNan = 18;  // words of the workspace tail: entries at the maximum, NaNs seen
__global__ __launch_bounds__(kBlock) void wgrad_tail_kernel(const float *__restrict__ dout, int64_t floats,
                                                            uint32_t *__restrict__ bounds, int every) {
  float top = 0.0f;
#pragma unroll
  for (int q = 0; q < 4; ++q) top = __builtin_fmaxf(top, __uint_as_float(bounds[q]));
  // (max == 0: no entry counts)
  const float inv = top > 0.0f ? 1.0f / top : 0.0f;
  const int lane = threadIdx.x & (kWave - 1);
  const int64_t wave = ((int64_t)blockIdx.x * kBlock + threadIdx.x) / kWave, waves = (int64_t)gridDim.x * (kBlock / kWave);
  constexpr int64_t kPiece = 256;  // floats per sampled piece (a wave's 64 x 16 bytes)
  const int64_t kEvery = every;
  const int64_t pieces = (floats + kPiece * kEvery - 1) / (kPiece * kEvery);
  const bool vec = ((uintptr_t)dout & 15) == 0;
  unsigned long long sum = 0;
  unsigned nonzero = 0, at_top = 0, nans = 0;
  for (int64_t p = wave; p < pieces; p += waves) {
    const int64_t at = p * kPiece * kEvery + 4 * lane;
    float v[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (vec && at + 4 <= floats) {
      const f32x4 t = *reinterpret_cast<const f32x4 *>(dout + at);
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = t[i];
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (at + i < floats) v[i] = dout[at + i];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float a = __builtin_fabsf(v[i]);
      nans += a != a;
      const bool top_entry = a > 0.0f && a >= top;  // (NaN: every comparison false -- counted above, in no sum)
      at_top += top_entry;
      const bool counted = a > 0.0f && !top_entry;
      nonzero += counted;
      const float frac = counted ? __builtin_fminf(a * inv, 1.0f) * 16777216.0f : 0.0f;
      sum += (unsigned long long)(unsigned)frac;
    }
  }
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) {
    sum += __shfl_down(sum, off, kWave);
    nonzero += __shfl_down(nonzero, off, kWave);
    at_top += __shfl_down(at_top, off, kWave);
    nans += __shfl_down(nans, off, kWave);
  }
  __shared__ unsigned long long red_sum[kBlock / kWave];
  __shared__ unsigned red_nz[kBlock / kWave], red_top[kBlock / kWave], red_nan[kBlock / kWave];
  __shared__ bool last;
  if (lane == 0) {
    red_sum[threadIdx.x / kWave] = sum;
    red_nz[threadIdx.x / kWave] = nonzero;
    red_top[threadIdx.x / kWave] = at_top;
    red_nan[threadIdx.x / kWave] = nans;
  }
  __syncthreads();
  unsigned long long *total = reinterpret_cast<unsigned long long *>(bounds + kGuardSum);
  if (threadIdx.x == 0) {
    unsigned long long a = 0;
    unsigned b = 0, c = 0, d = 0;
    for (int w = 0; w < kBlock / kWave; ++w) {
      a += red_sum[w];
      b += red_nz[w];
      c += red_top[w];
      d += red_nan[w];
    }
    if (a) atomicAdd(total, a);
    if (b) atomicAdd(bounds + kGuardNonzero, b);
    if (c) atomicAdd(bounds + kGuardTop, c);
    if (d) atomicAdd(bounds + kGuardNan, d);
    __threadfence();
    last = atomicAdd(bounds + kGuardTicket, 1u) == gridDim.x - 1;
  }
  __syncthreads();
  if (last && threadIdx.x == 0) {
    __threadfence();
    const unsigned long long a = __hip_atomic_load(total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long b = __hip_atomic_load(bounds + kGuardNonzero, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned nan_seen = __hip_atomic_load(bounds + kGuardNan, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // max / mean of the entries below the maximum > 2^kGuardRatio  <=>  count * 2^24 > 2^kGuardRatio * sum of their
    // 24-bit fractions (no entry below the maximum in the sample: nothing to lose, fp16 planes); NaN: the exact planes
    const unsigned fire = (nan_seen != 0u || (b << (24 - kGuardRatio)) > a) ? 1u : 0u;
    bounds[kGuardFlag] = fire;
    bounds[kGuardCalls] += 1u;
    bounds[kGuardFires] += fire;
  }
}

// mode of the planes of a weight-gradient call: RL8_WGRAD_PLANES / RL8_WGRAD_GATE_PLANES = "f16" (default): fp16 planes
// under the guard; "f16!": fp16 planes whatever the data (diagnostics); "bf16": the exact planes always.  Read per call,
// so that a diagnostic can form the same gradient all three ways in one process.
enum PlaneMode { kPlanesGuarded = 0, kPlanesF16 = 1, kPlanesBf16 = 2 };
static PlaneMode plane_mode(const char *name) {
  const char *v = getenv(name);
  if (!v || !v[0]) return kPlanesGuarded;
  if (v[0] == 'b') return kPlanesBf16;
  return (v[0] == 'f' && v[1] == '1' && v[2] == '6' && v[3] == '!') ? kPlanesF16 : kPlanesGuarded;
}

// the guard's decision for the call whose bounds were just requested (same stream, behind wgrad_bounds_kernel)
static int launch_wgrad_tail(hipStream_t s, const float *dout, int64_t floats, uint32_t *bounds) {
  const int every = floats <= ((int64_t)1 << 20) ? 1 : 16;  // small calls (minibatches, tests): every entry
  const int64_t pieces = (floats + 256 * every - 1) / (256 * every);
  const int64_t want = (pieces + (kBlock / kWave) - 1) / (kBlock / kWave);
  const int grid = (int)(want < 1 ? 1 : want > 2 * kCUs ? 2 * kCUs : want);
  wgrad_tail_kernel<<<grid, kBlock, 0, s>>>(dout, floats, bounds, every);
  return launch_status();
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

// Grids of the two halves of the fused backward: see fused_backward_grids below.
static void fused_backward_grids(int64_t m, int *g1, int *g2);

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

template <int DIN>
static int launch_wgrad_loadh(int grid, hipStream_t s, const float *dz, const float *x, int64_t rows, float *workspace,
                              const WgradOperands &ops) {
  static LdsOptIn lds_attr_set_0;
  if (const int e_lds_attr_set_0 = allow_dynamic_lds(lds_attr_set_0, reinterpret_cast<const void *>(&mlp_wgrad_split_kernel<DIN, 0, true>), 160 * 1024)) return e_lds_attr_set_0;
  mlp_wgrad_split_kernel<DIN, 0, true><<<grid, kWsThreads, 2 * kWsStageBytes, s>>>(
      dz, x, nullptr, nullptr, rows, DIN, workspace, WgradFusedArgs{}, ops);
  return launch_status();
}

// ---- the two-operand weight gradient (the LSTM's dW_hh per gate) on sixteen waves (round 4) ---------------------------
// The two-operand mode of mlp_wgrad_split_kernel -- both operands read from memory -- on fp16 planes: each operand two
// planes on one power of two per launch, dZ's low plane wide, optional column sums for dW_ih / db.  1024 threads: twice
// the loads in flight per CU on a kernel that runs at the rate HBM delivers its 1-KiB row pieces.  Same grouping (four
// gates per launch, the four workgroups that walk the same rows on one XCD), slabs and column-sum rows.
template <int DIN>
__global__ __launch_bounds__(kW16Threads, 1) void mlp_wgrad_loadh16_kernel(
    const float *__restrict__ dz, const float *__restrict__ x, int64_t m, float *__restrict__ slabs, WgradOperands ops) {
  constexpr int kIn = DIN;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, l32 = lane & 31, hh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wj = wave >> 2, wi = wave & 3;
  const int col = tid & 255, q = wave >> 2;

  const int ea = f16_bound_exponent(__uint_as_float(*ops.dz_bound) * 1.0001f);
  const int eb = f16_bound_exponent(__uint_as_float(*ops.h_bound) * 1.0001f);
  const float scale_a = __builtin_amdgcn_ldexpf(1.0f, kF16Top - ea), scale_b = __builtin_amdgcn_ldexpf(1.0f, kF16Top - eb);
  const float inv_ab = __builtin_amdgcn_ldexpf(1.0f, ea + eb - 2 * kF16Top);
  const float k2048 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(0x45000000));
  const uint32_t k_low = (uint32_t)__builtin_amdgcn_readfirstlane((int)kF16GateLowMask);

  const int64_t chunks = (m + kWsChunk - 1) / kWsChunk;
  const bool grouped = ops.groups == 4;
  const int gate_q = grouped ? ((int)blockIdx.x >> 3) & 3 : 0;
  const int64_t bid = grouped ? ((blockIdx.x & 7) | ((blockIdx.x >> 5) << 3)) : blockIdx.x;
  const int64_t stride = grouped ? gridDim.x / 4 : gridDim.x;
  const int64_t out_id = grouped ? gate_q * stride + bid : blockIdx.x;
  const float *dzp = dz + (grouped ? gate_q * ops.group_dz_offset : 0);
  const int64_t mine = (chunks - bid + stride - 1) / stride;
  const bool want_colsums = ops.colsums != nullptr;
  float cs_b = 0.0f, cs_w[kIn];
#pragma unroll
  for (int c = 0; c < kIn; ++c) cs_w[c] = 0.0f;

  constexpr int kXv = (4 * kIn + 7) / 8;
  struct Raw {
    float a[4], b[4];
    f32x8 xv[kXv];
  };
  auto issue = [&](Raw &r, int64_t n) {
    const int64_t row0 = (bid + n * stride) * kWsChunk + 4 * q;
    const int64_t left = m - row0;
    const int rows = left <= 0 ? 0 : left < 4 ? (int)left : 4;
    const int64_t at = rows > 0 ? row0 : 0;
    const __amdgpu_buffer_rsrc_t ar = buffer_rsrc(dzp + at * ops.dz_pitch, rows > 0 ? ((rows - 1) * ops.dz_pitch + kHidden) * 4 : 0);
    const __amdgpu_buffer_rsrc_t br = buffer_rsrc(ops.h + at * ops.h_pitch, rows > 0 ? ((rows - 1) * ops.h_pitch + kHidden) * 4 : 0);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      r.a[e] = buffer_load_f32(ar, col * 4, e * (ops.dz_pitch * 4));
      r.b[e] = buffer_load_f32(br, col * 4, e * (ops.h_pitch * 4));
    }
    const u32x4 rx = scalar_rsrc(want_colsums ? x + at * kIn : reinterpret_cast<const float *>(slabs), want_colsums ? rows * kIn * 4 : 0);
    r.xv[0] = scalar_buffer_load_x8<0>(rx);
    if constexpr (kXv > 1) r.xv[1] = scalar_buffer_load_x8<32>(rx);
    if constexpr (kXv > 2) r.xv[2] = scalar_buffer_load_x8<64>(rx);
    if constexpr (kXv > 3) r.xv[3] = scalar_buffer_load_x8<96>(rx);
  };
  auto land = [&](Raw &r) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < kXv; ++i) scalar_tie(r.xv[i]);
  };
  auto produce = [&](const Raw &r, int stage) {
    unsigned char *base = smem + stage * kW16FusedStageBytes + ((q >> 1) * kHidden + col) * 16 + (q & 1) * 8;
    if (want_colsums) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        cs_b += r.a[e];
#pragma unroll
        for (int c = 0; c < kIn; ++c) cs_w[c] = __builtin_fmaf(r.a[e], r.xv[(e * kIn + c) >> 3][(e * kIn + c) & 7], cs_w[c]);
      }
    }
    uint32_t ah[2], al[2], bh[2], bl[2];
#pragma unroll
    for (int e = 0; e < 4; e += 2) {
      f16_pair_scaled_wide(r.a[e], r.a[e + 1], scale_a, k2048, ah[e >> 1], al[e >> 1]);
      f16_pair_scaled(r.b[e], r.b[e + 1], scale_b, bh[e >> 1], bl[e >> 1]);
    }
    *reinterpret_cast<u32x2 *>(base) = u32x2{ah[0], ah[1]};
    *reinterpret_cast<u32x2 *>(base + 2 * kHidden * 16) = u32x2{al[0], al[1]};
    *reinterpret_cast<u32x2 *>(base + 4 * kHidden * 16) = u32x2{bh[0], bh[1]};
    *reinterpret_cast<u32x2 *>(base + 6 * kHidden * 16) = u32x2{bl[0], bl[1]};
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
  auto consume = [&](int stage) {  // (as mlp_wgrad_fused16_kernel's)
    const unsigned char *base = smem + stage * kW16FusedStageBytes;
    const unsigned char *ap = base + (hh * kHidden + 64 * wj + l32) * 16;
    const unsigned char *bp = base + 4 * kHidden * 16 + (hh * kHidden + 64 * wi + l32) * 16;
    u32x4 ah[2], bh[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      ah[t] = *reinterpret_cast<const u32x4 *>(ap + t * 512);
      bh[t] = *reinterpret_cast<const u32x4 *>(bp + t * 512);
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, ah[a]), __builtin_bit_cast(half8, bh[b]), acc[a][b], 0, 0, 0);
    {
      u32x4 bl[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) bl[t] = *reinterpret_cast<const u32x4 *>(bp + 2 * kHidden * 16 + t * 512);
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, ah[a]), __builtin_bit_cast(half8, bl[b]), acc[a][b], 0, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      ah[t] = *reinterpret_cast<const u32x4 *>(ap + 2 * kHidden * 16 + t * 512);
#pragma unroll
      for (int r = 0; r < 4; ++r) bh[t][r] = f16_pair_times(bh[t][r], k_low);
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, ah[a]), __builtin_bit_cast(half8, bh[b]), acc[a][b], 0, 0, 0);
  };

  Raw ra, rb;
  issue(ra, 0);
  land(ra);
  produce(ra, 0);
  issue(ra, 1);
  land(ra);
  produce(ra, 1);
  issue(ra, 2);
  land(ra);
  __syncthreads();
  const int64_t steps = (mine + 3) & ~(int64_t)3;
#pragma unroll 1
  for (int64_t n = 0; n < steps; n += 4) {
    issue(rb, n + 3);
    consume(0);
    produce(ra, 2);
    land(rb);
    issue(ra, n + 4);
    consume(1);
    produce(rb, 3);
    land(ra);
    __syncthreads();
    issue(rb, n + 5);
    consume(2);
    produce(ra, 0);
    land(rb);
    issue(ra, n + 6);
    consume(3);
    produce(rb, 1);
    land(ra);
    __syncthreads();
  }

  float *slab = slabs + out_id * kHidden * kHidden;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int j = 64 * wj + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * hh;
#pragma unroll
      for (int b = 0; b < 2; ++b) slab[j * kHidden + 64 * wi + 32 * b + l32] = acc[a][b][r] * inv_ab;
    }

  if (want_colsums) {  // the four sample quarters of a column folded in a fixed order
    float *red = reinterpret_cast<float *>(smem);  // [4][256][1 + kIn]
    __syncthreads();
    red[(q * kHidden + col) * (1 + kIn)] = cs_b;
#pragma unroll
    for (int c = 0; c < kIn; ++c) red[(q * kHidden + col) * (1 + kIn) + 1 + c] = cs_w[c];
    __syncthreads();
    if (q == 0) {
      auto total = [&](int k) {
        return ((red[col * (1 + kIn) + k] + red[(kHidden + col) * (1 + kIn) + k]) + red[(2 * kHidden + col) * (1 + kIn) + k]) +
               red[(3 * kHidden + col) * (1 + kIn) + k];
      };
      float *row = ops.colsums + out_id * (kHidden * (kIn + 1));
      const bool more = ops.colsum_accumulate != 0;
      const float b = total(0);
      row[kHidden * kIn + col] = more ? row[kHidden * kIn + col] + b : b;
#pragma unroll
      for (int c = 0; c < kIn; ++c) {
        const float w = total(1 + c);
        row[col * kIn + c] = more ? row[col * kIn + c] + w : w;
      }
    }
  }
}

template <int DIN>
static int launch_wgrad_loadh16(int grid, hipStream_t s, const float *dz, const float *x, int64_t rows, float *workspace,
                                const WgradOperands &ops) {
  static LdsOptIn opt;
  if (const int e = allow_dynamic_lds(opt, reinterpret_cast<const void *>(&mlp_wgrad_loadh16_kernel<DIN>), 160 * 1024)) return e;
  mlp_wgrad_loadh16_kernel<DIN><<<grid, kW16Threads, kW16Stages * kW16FusedStageBytes, s>>>(dz, x, rows, workspace, ops);
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
                         const uint32_t *dz_bound, const uint32_t *h_bound, void *stream, int groups = 1) {
  if (!dz || !h || !workspace || !dw_out) return RL8_ENULL;
  if (m <= 0 || dz_pitch < kHidden || h_pitch < kHidden) return RL8_ESIZE;
  if ((int64_t)kWsChunk * (dz_pitch > h_pitch ? dz_pitch : h_pitch) * 4 >= (int64_t)1 << 31) return RL8_ESIZE;
  if (!aligned16(workspace) || !aligned16(dw_out)) return RL8_EALIGN;
  if (colsums) {
    if (!x || !colsum_rows_out) return RL8_ENULL;
    if (d_in < 1 || d_in > 7) return RL8_ESIZE;  // (round 6: 4, 6, 7 too)
  }
  hipStream_t s = (hipStream_t)stream;
  int first_grid = 0;
  for (int64_t at = 0; at < m; at += kWgradSegmentRows) {  // segments summed in order, as above
    const int64_t rows = m - at < kWgradSegmentRows ? m - at : kWgradSegmentRows;
    const int64_t chunks = (rows + kWsChunk - 1) / kWsChunk;
    int grid = (int)(chunks < kCUs ? chunks : kCUs);
    if (groups == 4) {  // a multiple of 32 workgroups, a quarter of them per gate (each gate's share <= chunks)
      const int64_t per_gate = chunks < kCUs / 4 ? chunks : kCUs / 4;
      grid = (int)((per_gate / 8) * 8) * 4;
      if (grid == 0) return RL8_ESIZE;  // (fewer than 128 rows: the caller uses the per-gate call)
    }
    if (at == 0) first_grid = grid;
    if (grid > first_grid) grid = first_grid;  // (later segments add to the first one's rows)
    const WgradOperands ops{h + at * h_pitch, (int)dz_pitch, (int)h_pitch, colsums, at > 0, dz_bound, h_bound,
                            groups, groups == 4 ? kHidden : 0};
    const float *xs = colsums ? x + at * d_in : nullptr;
    int status;
    if (dz_bound) {  // fp16 planes behind the caller's bound: the sixteen-wave kernel
      switch (colsums ? d_in : 1) {
        case 1: status = launch_wgrad_loadh16<1>(grid, s, dz + at * dz_pitch, xs, rows, workspace, ops); break;
        case 2: status = launch_wgrad_loadh16<2>(grid, s, dz + at * dz_pitch, xs, rows, workspace, ops); break;
        case 3: status = launch_wgrad_loadh16<3>(grid, s, dz + at * dz_pitch, xs, rows, workspace, ops); break;
        case 4: status = launch_wgrad_loadh16<4>(grid, s, dz + at * dz_pitch, xs, rows, workspace, ops); break;
        case 5: status = launch_wgrad_loadh16<5>(grid, s, dz + at * dz_pitch, xs, rows, workspace, ops); break;
        case 6: status = launch_wgrad_loadh16<6>(grid, s, dz + at * dz_pitch, xs, rows, workspace, ops); break;
        default: status = launch_wgrad_loadh16<7>(grid, s, dz + at * dz_pitch, xs, rows, workspace, ops); break;
      }
    } else {  // no bound: the exact bf16 planes
      switch (colsums ? d_in : 1) {
        case 1: status = launch_wgrad_loadh<1>(grid, s, dz + at * dz_pitch, xs, rows, workspace, ops); break;
        case 2: status = launch_wgrad_loadh<2>(grid, s, dz + at * dz_pitch, xs, rows, workspace, ops); break;
        case 3: status = launch_wgrad_loadh<3>(grid, s, dz + at * dz_pitch, xs, rows, workspace, ops); break;
        case 4: status = launch_wgrad_loadh<4>(grid, s, dz + at * dz_pitch, xs, rows, workspace, ops); break;
        case 5: status = launch_wgrad_loadh<5>(grid, s, dz + at * dz_pitch, xs, rows, workspace, ops); break;
        case 6: status = launch_wgrad_loadh<6>(grid, s, dz + at * dz_pitch, xs, rows, workspace, ops); break;
        default: status = launch_wgrad_loadh<7>(grid, s, dz + at * dz_pitch, xs, rows, workspace, ops); break;
      }
    }
    if (status != 0) return status;
    for (int q = 0; q < groups; ++q)  // (gate q: slabs [q grid / groups, ...), dW block q)
      mlp_wgrad_split_reduce_kernel<<<kHidden * kHidden / kBlock, kBlock, 0, s>>>(
          workspace + (int64_t)q * (grid / groups) * kHidden * kHidden, grid / groups, dw_out + (int64_t)q * kHidden * kHidden,
          accumulate || at > 0);
  }
  if (colsum_rows_out) *colsum_rows_out = first_grid / groups;
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

/* The four gates of an LSTM timestep in ONE launch: dz = the step's dG rows ([m] rows of dz_pitch floats, gate q at
 * columns [256 q, 256 q + 256)), dw_out [4][256][256] (+)= dG_q^T h per gate, colsums (optional, with x / d_in)
 * [4][*colsum_rows_out][256 (d_in + 1)] -- so that h is read from HBM once instead of four times (three of the four
 * workgroups that walk the same rows find them in their XCD's L2).  m >= 128; otherwise as
 * rl8_mlp_wgrad_f16_strided_f32. */
RL8_API int rl8_lstm_wgrad_f16_f32(const float *dz, int64_t dz_pitch, const uint32_t *dz_bound, const float *h,
                                   int64_t h_pitch, const uint32_t *h_bound, int64_t m, float *workspace, float *dw_out,
                                   int accumulate, const float *x, int d_in, float *colsums, int *colsum_rows_out,
                                   void *stream) {
  if (!dz_bound || !h_bound) return RL8_ENULL;
  if (dz_pitch < 4 * kHidden) return RL8_ESIZE;
  return wgrad_strided(dz, dz_pitch, h, h_pitch, m, workspace, dw_out, accumulate, x, d_in, colsums, colsum_rows_out, dz_bound,
                       h_bound, stream, 4);
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
  if (m <= 0 || !rl8_mlp_backward_f16_supports(d_in, n_out)) return RL8_ESIZE;
  if (!aligned16(h2) || !aligned16(workspace) || !aligned16(dw2_out)) return RL8_EALIGN;
  int g1, g2;
  fused_backward_grids(m, &g1, &g2);
  const int stride = (int)rl8_mlp_backward_partial_floats(d_in, n_out);
  hipStream_t s = (hipStream_t)stream;
  // Two fp16 planes per operand, three products -- unless the guard finds this call's dOut spread over too many
  // binades, or RL8_WGRAD_PLANES=bf16 asks for them always: the six-product kernel on exact bf16 planes.
  const PlaneMode planes = plane_mode("RL8_WGRAD_PLANES");
  static const bool gate_kernel = env_int("RL8_WGRAD_GATE_OFF") == 0;
  // (one output with h2 given: the gate-plane kernel on three exact bf16 planes -- no bounds, nothing to guard)
  const bool f16 = planes != kPlanesBf16 && !(n_out == 1 && gate_kernel);
  uint32_t *bounds = nullptr;
  if (f16) {
    bounds = n_out == 1 ? launch_wgrad_bounds<1>(s, dout, x, m, d_in, workspace)
             : n_out == 2 ? launch_wgrad_bounds<2>(s, dout, x, m, d_in, workspace)
             : n_out == 3 ? launch_wgrad_bounds<3>(s, dout, x, m, d_in, workspace)
                          : launch_wgrad_bounds<4>(s, dout, x, m, d_in, workspace);
    if (!bounds) return launch_status() ? launch_status() : RL8_ESIZE;
    if (planes == kPlanesGuarded && launch_wgrad_tail(s, dout, m * n_out, bounds) != 0) return launch_status();
  }
  const uint32_t *guard = planes == kPlanesGuarded && f16 ? bounds : nullptr;
  // Segments of kWgradSegmentRows samples, summed in order (see there).  The first one
  // runs the grid the data-gradient kernel counted on (g2 rows of partials written, the
  // rest zeroed); later ones add to as many of those rows as they have workgroups.
  for (int64_t at = 0; at < m; at += kWgradSegmentRows) {
    const int64_t rows = m - at < kWgradSegmentRows ? m - at : kWgradSegmentRows;
    const int64_t chunks = (rows + kWsChunk - 1) / kWsChunk;
    const int grid = at == 0 ? g2 : (int)(chunks < g2 ? chunks : g2);
    const WgradFusedArgs fused{dout + at * n_out, w3, partials, stride, g1, at > 0, nullptr, nullptr, bounds, guard, 0};
    WgradFusedArgs exact = fused;  // the same segment on the bf16 planes, should the guard have said so
    exact.guard_want = 1;
    const float *h2s = h2 + at * kHidden, *xs = x + at * d_in;
    int status = RL8_ESIZE;
    // one output: the gate-plane kernel (three plane products per 16 samples instead of six)
    if (n_out == 1 && gate_kernel) {
      switch (d_in) {
        case 1: status = launch_wgrad_gate<1>(grid, s, h2s, xs, w1, b1, rows, workspace, fused); break;
        case 2: status = launch_wgrad_gate<2>(grid, s, h2s, xs, w1, b1, rows, workspace, fused); break;
        case 3: status = launch_wgrad_gate<3>(grid, s, h2s, xs, w1, b1, rows, workspace, fused); break;
        case 4: status = launch_wgrad_gate<4>(grid, s, h2s, xs, w1, b1, rows, workspace, fused); break;
        case 5: status = launch_wgrad_gate<5>(grid, s, h2s, xs, w1, b1, rows, workspace, fused); break;
        case 6: status = launch_wgrad_gate<6>(grid, s, h2s, xs, w1, b1, rows, workspace, fused); break;
        default: status = launch_wgrad_gate<7>(grid, s, h2s, xs, w1, b1, rows, workspace, fused); break;
      }
      if (status != 0) return status;
      mlp_wgrad_split_reduce_kernel<<<kHidden * kHidden / kBlock, kBlock, 0, s>>>(workspace, grid, dw2_out, at > 0);
      continue;
    }
#define RL8_WGRAD_FUSED(D, N) \
  if (d_in == D && n_out == N) { \
    status = f16 ? launch_wgrad_fused16<D, N>(grid, s, h2s, xs, w1, b1, rows, workspace, fused) \
                 : launch_wgrad_fused<D, N>(grid, s, h2s, xs, w1, b1, rows, d_in, workspace, fused); \
    if (status == 0 && guard) status = launch_wgrad_fused<D, N>(grid, s, h2s, xs, w1, b1, rows, d_in, workspace, exact); \
  }
    RL8_WGRAD_FUSED(1, 1) RL8_WGRAD_FUSED(1, 2) RL8_WGRAD_FUSED(1, 3) RL8_WGRAD_FUSED(1, 4)
    RL8_WGRAD_FUSED(2, 1) RL8_WGRAD_FUSED(2, 2) RL8_WGRAD_FUSED(2, 3) RL8_WGRAD_FUSED(2, 4)
    RL8_WGRAD_FUSED(3, 1) RL8_WGRAD_FUSED(3, 2) RL8_WGRAD_FUSED(3, 3) RL8_WGRAD_FUSED(3, 4)
    RL8_WGRAD_FUSED(4, 1) RL8_WGRAD_FUSED(4, 2) RL8_WGRAD_FUSED(4, 3) RL8_WGRAD_FUSED(4, 4)
    RL8_WGRAD_FUSED(5, 1) RL8_WGRAD_FUSED(5, 2) RL8_WGRAD_FUSED(5, 3) RL8_WGRAD_FUSED(5, 4)
    RL8_WGRAD_FUSED(6, 1) RL8_WGRAD_FUSED(6, 2) RL8_WGRAD_FUSED(6, 3) RL8_WGRAD_FUSED(6, 4)
    RL8_WGRAD_FUSED(7, 1) RL8_WGRAD_FUSED(7, 2) RL8_WGRAD_FUSED(7, 3)  // (7 x 4: rl8_mlp_backward_f16_supports)
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
  if (m <= 0 || !rl8_mlp_backward_f16_supports(d_in, 2)) return RL8_ESIZE;
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
      case 4: status = launch_wgrad_gate<4, true>(grid, s, h2s, xs, w1, b1, rows, workspace, fused); break;
      case 5: status = launch_wgrad_gate<5, true>(grid, s, h2s, xs, w1, b1, rows, workspace, fused); break;
      case 6: status = launch_wgrad_gate<6, true>(grid, s, h2s, xs, w1, b1, rows, workspace, fused); break;
      default: status = launch_wgrad_gate<7, true>(grid, s, h2s, xs, w1, b1, rows, workspace, fused); break;
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
  if (m <= 0 || (n_out != 1 && n_out != 2) || !rl8_mlp_backward_f16_supports(d_in, n_out)) return RL8_ESIZE;
  if (!aligned16(gate2) || !aligned16(workspace) || !aligned16(dw2_out) || ((uintptr_t)dout & 7) != 0) return RL8_EALIGN;
  int g1, g2;
  fused_backward_grids(m, &g1, &g2);
  const int stride = (int)rl8_mlp_backward_partial_floats(d_in, n_out);
  const int off_dw3 = kHidden * d_in + 2 * kHidden;
  hipStream_t s = (hipStream_t)stream;
  // Two fp16 planes of dOut * h1 -- unless the guard finds this call's dOut spread over too many binades, or
  // RL8_WGRAD_GATE_PLANES=bf16 asks for them always: the three exact bf16 planes.
  const PlaneMode planes = plane_mode("RL8_WGRAD_GATE_PLANES");
  const bool f16 = planes != kPlanesBf16;
  uint32_t *bounds = nullptr;
  if (f16) {
    bounds = n_out == 2 ? launch_wgrad_bounds<2>(s, dout, x, m, d_in, workspace) : launch_wgrad_bounds<1>(s, dout, x, m, d_in, workspace);
    if (!bounds) return launch_status() ? launch_status() : RL8_ESIZE;
    if (planes == kPlanesGuarded && launch_wgrad_tail(s, dout, m * n_out, bounds) != 0) return launch_status();
  }
  const uint32_t *guard = planes == kPlanesGuarded ? bounds : nullptr;
  for (int64_t at = 0; at < m; at += kWgradSegmentRows) {  // segments summed in order, as rl8_mlp_wgrad_fused_split_f32
    const int64_t rows = m - at < kWgradSegmentRows ? m - at : kWgradSegmentRows;
    const int64_t chunks = (rows + kWsChunk - 1) / kWsChunk;
    const int grid = at == 0 ? g2 : (int)(chunks < g2 ? chunks : g2);
    const WgradFusedArgs fused{dout + at * n_out, w3, partials, stride, g1, at > 0, gate2 + at * 8, b2, bounds, guard, 0};
    WgradFusedArgs exact = fused;  // the same segment on the bf16 planes, should the guard have said so
    exact.guard_want = 1;
    const float *xs = x + at * d_in;
    int status = RL8_ESIZE;
#define RL8_WGRAD_BITS(D) \
  if (d_in == D) { \
    status = f16 ? (n_out == 2 ? launch_wgrad_gate16<D, true>(grid, s, xs, w1, b1, rows, workspace, fused) \
                               : launch_wgrad_gate16<D, false>(grid, s, xs, w1, b1, rows, workspace, fused)) \
           : n_out == 2 ? launch_wgrad_gate<D, true, true>(grid, s, nullptr, xs, w1, b1, rows, workspace, fused) \
                        : launch_wgrad_gate<D, false, true>(grid, s, nullptr, xs, w1, b1, rows, workspace, fused); \
    if (status == 0 && guard) \
      status = n_out == 2 ? launch_wgrad_gate<D, true, true>(grid, s, nullptr, xs, w1, b1, rows, workspace, exact) \
                          : launch_wgrad_gate<D, false, true>(grid, s, nullptr, xs, w1, b1, rows, workspace, exact); \
  }
    RL8_WGRAD_BITS(1) RL8_WGRAD_BITS(2) RL8_WGRAD_BITS(3) RL8_WGRAD_BITS(4) RL8_WGRAD_BITS(5) RL8_WGRAD_BITS(6) RL8_WGRAD_BITS(7)
#undef RL8_WGRAD_BITS
    if (status != 0) return status;
    if (n_out == 2)
      mlp_wgrad_gate_reduce_kernel<true><<<kHidden, kBlock, 0, s>>>(workspace, grid, dw2_out, at > 0, w2, w3, partials + off_dw3);
    else
      mlp_wgrad_gate_reduce_kernel<false><<<kHidden, kBlock, 0, s>>>(workspace, grid, dw2_out, at > 0, w2, w3, partials + off_dw3);
  }
  return launch_status();
}
