// a-9 (recurrent PPO), second generation: one LSTM timestep of the default
// recurrent models (src/rl8/models/_recurrent.py:201-321 -> torch.nn.LSTM(d, 256,
// batch_first)) with the recurrent product h_{t-1} x W_hh^T as an fp32-ACCURATE product
// on the 16-bit matrix pipe.  First on bf16 planes (three per operand, six plane products,
// the scheme of mlp_split_kernels.hip); now on fp16 planes (two per operand, THREE plane
// products: mlp_f16_kernels.hip), which costs nothing here: |h| < 1 by construction
// (h = o * tanh(c)), so the state's planes are those of h * 2^14 with no per-row factor, and
// W_hh gets one power of two for the matrix (behind the planes, like the towers' W2); the
// inverse of both meets the accumulators in the epilogue's first fma.
// lstm_kernels.hip runs the same step on fp32 MFMAs (64 cycles per 32x32x2 block; 108-129
// TFLOP/s); this one runs 5.3x fewer matrix-pipe cycles per product.
//
// Shape.  The time loop is OUTSIDE the kernel (one launch per timestep): with both
// operands as bf16 planes in LDS there is no room to keep a tile's h_t on chip between
// steps, and a launch over 2^19 sequences is long enough that it does not matter.
//   rl8_lstm_split_state      h_{t-1} [B][256] fp32 -> two fp16 planes in fragment order
//                             (1 KiB per row; HBM-bound, 2 KiB of traffic per row);
//   rl8_lstm_step_split_f32   per 128-row tile and block of 64 hidden units: gate
//                             pre-activations = h planes x W_hh planes, BOTH operands
//                             HBM/L2 -> LDS by direct-to-LDS loads (no registers, no VALU
//                             in the matrix loop); the input projection (d_in <= 7), the
//                             biases, the gate non-linearities and the cell update on the
//                             accumulators.
// A workgroup's 256 accumulator columns are [i | f | g | o] x 32 units per wave column
// half, so a lane holds all four gates of its (row, unit) pairs in the same register slot
// -- the cell update needs no exchange -- and, lane = unit, every h / c / gate store is two
// full 128-byte row segments.
#include "split_tile.hip.h"

namespace rl8 {

constexpr int kLsUnits = 64;                      // hidden units per workgroup pass
constexpr int kLsBlocks = kHidden / kLsUnits;     // unit blocks (grid dimension)
constexpr int kLsABytes = 2 * kSplitPlaneStride;  // LDS: [plane][k-half][row] x 16 B
constexpr int kLsBBytes = 2 * 8 * 1024;           // LDS: [column tile][plane] x 1 KiB
constexpr int kLsStageBytes = kLsABytes + kLsBBytes;
constexpr int kLsWBlockBytes = kSplitSteps * kLsBBytes;           // W_hh planes of one unit block: 256 KiB
constexpr int kLsPackedBytes = kLsBlocks * kLsWBlockBytes;        // 1 MiB (+ 16 bytes: scale, 1 / scale)
constexpr int kLsAChunkBytes = 8 * 1024;          // one k-step of a 128-row tile: [plane][k-half][row half] x 1 KiB
constexpr int kLsATileBytes = kSplitSteps * kLsAChunkBytes;
constexpr int kLsInCols = 8;                      // [w_ih (d_in <= 7) ... | b_ih + b_hh] per gate row
constexpr int kLsXBytes = 7 * kSplitRows * 4;     // x tile in LDS, [input][row]
#ifndef RL8_LS_DIAG
#define RL8_LS_DIAG 0   // tuning builds: 1 no h/c/gate stores (one checksum store per lane), 2 no transcendentals, 4 one plane product of three
#endif
constexpr int kLsDiag = RL8_LS_DIAG;

// Gate non-linearities as in lstm_kernels.hip (hardware exp2 / rcp).
__device__ __forceinline__ float ls_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float ls_tanh(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(__expf(2.0f * x) + 1.0f); }

// eight values times `scale` (a power of two) -> hi / lo fp16 planes
__device__ __forceinline__ void ls_planes(const float (&v)[8], float scale, u32x4 (&planes)[2]) {
#pragma unroll
  for (int e = 0; e < 8; e += 2) {
    uint32_t hi, lo;
    f16_pair(v[e] * scale, v[e + 1] * scale, hi, lo);
    planes[0][e >> 1] = hi;
    planes[1][e >> 1] = lo;
  }
}
constexpr float kLsStateScale = 16384.0f;  // 2^14: |h| < 1

// max |W_hh| -> {2^k, 2^-k} with max * 2^k < 2^14, behind the planes (one workgroup).
__global__ __launch_bounds__(1024) void lstm_whh_scale_kernel(const float *__restrict__ w_hh, float *__restrict__ tail) {
  __shared__ float red[1024];
  const int tid = threadIdx.x;
  float mx = 0.0f;
  // sixteen-byte loads, eight in flight per thread (one load at a time, the single workgroup took 73 us for the 1 MiB)
  const f32x4 *w4 = reinterpret_cast<const f32x4 *>(w_hh);
#pragma unroll 8
  for (int i = tid; i < kHidden * kHidden; i += 1024) {
    const f32x4 v = w4[i];
    mx = __builtin_fmaxf(__builtin_fmaxf(mx, __builtin_fmaxf(__builtin_fabsf(v[0]), __builtin_fabsf(v[1]))),
                         __builtin_fmaxf(__builtin_fabsf(v[2]), __builtin_fabsf(v[3])));
  }
  red[tid] = mx;
  __syncthreads();
  for (int half = 512; half > 0; half >>= 1) {
    if (tid < half) red[tid] = __builtin_fmaxf(red[tid], red[tid + half]);
    __syncthreads();
  }
  if (tid == 0) {
    const int e = f16_bound_exponent(red[0]);
    tail[0] = __builtin_amdgcn_ldexpf(1.0f, kF16Top - e);
    tail[1] = __builtin_amdgcn_ldexpf(1.0f, e - kF16Top);
    tail[2] = tail[3] = 0.0f;
  }
}

// W_hh [1024][256] (torch gate order i, f, g, o), times the matrix's power of two -> planes: 16-byte unit
// ((((ub*16 + s)*8 + ct)*2 + p)*64 + l) holds, for plane p, the eight values
//   W_hh[256 q + 64 ub + 32 wc + (l & 31)][16 s + 8 (l >> 5) + e],   ct = 4 wc + q,
// i.e. per unit block and k-step the 16 KiB the step's direct-to-LDS copy fetches, with
// column tile ct = (wave column half, gate).  wb [1024][8] = [w_ih row | 0.. | b_ih + b_hh].
__global__ __launch_bounds__(kBlock) void lstm_pack_split_kernel(const float *__restrict__ w_ih,
                                                                 const float *__restrict__ w_hh,
                                                                 const float *__restrict__ b_ih,
                                                                 const float *__restrict__ b_hh, int d_in,
                                                                 uint32_t *__restrict__ packed,
                                                                 float *__restrict__ wb) {
  const int unit = blockIdx.x * kBlock + threadIdx.x;  // (ub, s, ct, l)
  if (unit < 4 * kHidden * kLsInCols) {
    const int j = unit / kLsInCols, c = unit - j * kLsInCols;
    wb[unit] = c < d_in ? w_ih[j * d_in + c] : c == kLsInCols - 1 ? b_ih[j] + b_hh[j] : 0.0f;
  }
  if (unit >= kLsBlocks * kSplitSteps * 8 * 64) return;
  const int l = unit & 63, ct = (unit >> 6) & 7, s = (unit >> 9) & 15, ub = unit >> 13;
  const int j = kHidden * (ct & 3) + kLsUnits * ub + 32 * (ct >> 2) + (l & 31), k0 = 16 * s + 8 * (l >> 5);
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = w_hh[j * kHidden + k0 + e];
  const float scale = reinterpret_cast<const float *>(reinterpret_cast<const unsigned char *>(packed) + kLsPackedBytes)[0];
  u32x4 planes[2];
  ls_planes(v, scale, planes);
#pragma unroll
  for (int p = 0; p < 2; ++p)
    reinterpret_cast<u32x4 *>(packed)[(((ub * kSplitSteps + s) * 8 + ct) * 2 + p) * 64 + l] = planes[p];
}

// h [B][pitch] fp32 -> planes [tile][k-step][plane][k-half][row] x 16 B (rows past B are
// zero): the A operand of rl8_lstm_step_split_f32, one contiguous 8 KiB per tile and
// k-step.  Thread = (row, k-half); sixteen fragments each.  `bound_out` (or null): max |h| over the rows, as the bit
// pattern of the float (atomicMax of non-negative floats as integers), for the fp16-plane weight gradient that reads
// the same h_0 in the backward -- a separate abs + amax over 2^19 rows cost 0.33 ms per pass.
__global__ __launch_bounds__(kBlock) void lstm_split_state_kernel(const float *__restrict__ h, int64_t pitch,
                                                                  int64_t b, uint32_t *__restrict__ planes_out,
                                                                  uint32_t *__restrict__ bound_out) {
  __shared__ uint32_t block_max;
  if (threadIdx.x == 0) block_max = 0;
  __syncthreads();
  uint32_t mx = 0;
  const int rr = threadIdx.x & 127, kh = threadIdx.x >> 7;
  const int64_t tiles = (b + kSplitRows - 1) / kSplitRows;
  for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const int64_t row = tile * kSplitRows + rr;
    const float *src = h + row * pitch + 8 * kh;
    u32x4 *dst = reinterpret_cast<u32x4 *>(planes_out) + tile * (kLsATileBytes / 16) + kh * 128 + rr;
#pragma unroll 4
    for (int s = 0; s < kSplitSteps; ++s) {
      float v[8];
      if (row < b) {
        const f32x4 lo = *reinterpret_cast<const f32x4 *>(src + 16 * s), hi = *reinterpret_cast<const f32x4 *>(src + 16 * s + 4);
        v[0] = lo[0], v[1] = lo[1], v[2] = lo[2], v[3] = lo[3], v[4] = hi[0], v[5] = hi[1], v[6] = hi[2], v[7] = hi[3];
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = 0.0f;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) mx = max(mx, __float_as_uint(v[e]) & 0x7fffffffu);
      u32x4 planes[2];
      ls_planes(v, kLsStateScale, planes);
#pragma unroll
      for (int p = 0; p < 2; ++p) dst[s * (kLsAChunkBytes / 16) + p * 256] = planes[p];
    }
  }
  if (bound_out) {  // (uniform)
    atomicMax(&block_max, mx);
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(bound_out, block_max);
  }
}

struct LstmStepArgs {
  const float *x;       // [B][x_pitch]: this step's observations
  const float *c_prev;  // [B][c_prev_pitch]
  float *h_out, *c_out; // [B][*_pitch]
  float *gates;         // [B][gates_pitch] -> [4][256] post-activation i, f, g, o; or null
  void *planes_out;     // h_t as fp16 planes for the NEXT step (rl8_lstm_split_state's layout); or null
  int64_t x_pitch, c_prev_pitch, h_out_pitch, c_out_pitch, gates_pitch;
};

constexpr int kLsHPitch = 32 * 4 + 16;                                  // h scratch of the plane emission: [64 rows][144 B] per wave
constexpr int kLsScratchTail = 4 * 64 * kLsHPitch - kLsStageBytes;      // ... = chunk stage 1 + this tail
constexpr int lstm_step_lds_bytes() { return 2 * kLsStageBytes + kLsScratchTail + kLsXBytes; }
static_assert(lstm_step_lds_bytes() <= 80 * 1024, "two workgroups per CU");

// One timestep for rows [0, b): work item = (128-row tile, unit block ub); a workgroup
// keeps its unit block (gridDim.x is a multiple of 4), so W_hh's planes for it stay in L2
// and its gate rows of [w_ih | bias] are loaded once.
template <int DIN, bool SAVE>
__global__ __launch_bounds__(kBlock, 2) void lstm_step_split_kernel(
    const void *__restrict__ a_planes, const void *__restrict__ w_planes, const float *__restrict__ wb, int64_t b,
    LstmStepArgs args) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const unsigned lds0 = lds_offset(smem);
  const int tid = threadIdx.x, lane = tid & 63, l32 = lane & 31, hh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  // Work item = (tile, unit block); the four unit blocks of a tile read the same h planes (1 KiB per row), so they sit
  // on ONE XCD -- workgroup b runs on XCD b & 7 -- and next to each other in time: workgroups b, b + 8, b + 16, b + 24.
  // (With tile = b / 4 the four were on four XCDs, each L2 fetched the planes itself: 2.6x the algorithmic reads.)
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int ub = slot & (kLsBlocks - 1);
  const unsigned a_read = lds0 + hh * kSplitKhStride + (64 * wr + l32) * 16;
  const unsigned b_read = lds0 + kLsABytes + (4 * wc * 2) * 1024 + lane * 16;
  const __amdgpu_buffer_rsrc_t wrsrc =
      buffer_rsrc(static_cast<const unsigned char *>(w_planes) + ub * kLsWBlockBytes, kLsWBlockBytes);
  // undoes the operand scaling: 2^-14 of the state planes times W_hh's inverse power of two
  const float unscale = reinterpret_cast<const float *>(static_cast<const unsigned char *>(w_planes) + kLsPackedBytes)[1] *
                        (1.0f / kLsStateScale);
  const int64_t tiles = (b + kSplitRows - 1) / kSplitRows;
  const int64_t tile_stride = gridDim.x / kLsBlocks;

  // this lane's unit: gate rows of [w_ih | bias]
  const int unit = kLsUnits * ub + 32 * wc + l32;
  float w_in[4][DIN], bias[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float *row = wb + (kHidden * q + unit) * kLsInCols;
#pragma unroll
    for (int i = 0; i < DIN; ++i) w_in[q][i] = row[i];
    bias[q] = row[kLsInCols - 1];
  }

  auto a_rsrc = [&](int64_t tile) {
    return buffer_rsrc(static_cast<const unsigned char *>(a_planes) + tile * kLsATileBytes, kLsATileBytes);
  };
  // chunk ks of `tile` -> stage: 16 one-KiB blocks of W planes, 8 of h planes
  auto request = [&](const __amdgpu_buffer_rsrc_t &arsrc, int ks, int stage) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int block = wave * 4 + u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, smem + stage * kLsStageBytes + kLsABytes + block * 1024, 16,
                                               lane * 16, (ks * 16 + block) * 1024, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int c = wave * 2 + u;  // ((plane*2 + kh)*2 + half)
      const int lds_at = (c >> 2) * kSplitPlaneStride + ((c >> 1) & 1) * kSplitKhStride + (c & 1) * 1024;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(arsrc, smem + stage * kLsStageBytes + lds_at, 16, lane * 16,
                                               (ks * 8 + c) * 1024, 0, 0);
    }
  };
  auto step_barrier = [&]() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

  f32x16 acc[2][4];
  auto do_step = [&](auto first_tag, auto parity_tag, const __amdgpu_buffer_rsrc_t &next_rsrc, int next_ks) {
    constexpr bool FIRST = decltype(first_tag)::value;
    constexpr int P = decltype(parity_tag)::value;
    request(next_rsrc, next_ks, P ^ 1);
    const unsigned ar = a_read + P * kLsStageBytes, br = b_read + P * kLsStageBytes;
    SplitFrags f;  // ah / bh: hi planes, am / bm: lo planes -- all twelve fragments of the step at once
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
    wait_lds_all(f);
    if constexpr ((kLsDiag & 4) != 0) {  // tuning builds: one plane product of three
      f16_mma<FIRST>(f.ah, f.bh, acc);
    } else {
      f16_mma<FIRST>(f.am, f.bh, acc);  // lo x hi
      f16_mma<false>(f.ah, f.bm, acc);  // hi x lo
      f16_mma<false>(f.ah, f.bh, acc);  // hi x hi
    }
    __builtin_amdgcn_sched_barrier(0);
    step_barrier();
  };
  using T = std::true_type;
  using F = std::false_type;
  using P0 = std::integral_constant<int, 0>;
  using P1 = std::integral_constant<int, 1>;

  int64_t tile = (slot >> 2) * 8 + xcd;  // (gridDim.x is a multiple of 32: gridDim.x / 4 tiles in flight)
  if (tile < tiles) {
    request(a_rsrc(tile), 0, 0);
    step_barrier();
  }
  for (; tile < tiles; tile += tile_stride) {
    const int64_t r0 = tile * kSplitRows;
    const int rows = (int)((b - r0) < kSplitRows ? (b - r0) : kSplitRows);
    // this row's observation (threads 0..127), parked in LDS behind the matrix loop
    float xr[DIN];
    {
      const int row = tid & 127;
      const float *xp = args.x + (r0 + row) * args.x_pitch;
#pragma unroll
      for (int i = 0; i < DIN; ++i) xr[i] = (tid < kSplitRows && row < rows) ? xp[i] : 0.0f;
    }
    const __amdgpu_buffer_rsrc_t arsrc = a_rsrc(tile);
    const int64_t next = tile + tile_stride < tiles ? tile + tile_stride : tile;  // (last item: a harmless re-fetch)
    const __amdgpu_buffer_rsrc_t nrsrc = a_rsrc(next);
    do_step(T{}, P0{}, arsrc, 1);
    do_step(F{}, P1{}, arsrc, 2);
#pragma unroll 1
    for (int s = 2; s < kSplitSteps - 2; s += 2) {
      do_step(F{}, P0{}, arsrc, s + 1);
      do_step(F{}, P1{}, arsrc, s + 2);
    }
    do_step(F{}, P0{}, arsrc, kSplitSteps - 1);
    do_step(F{}, P1{}, nrsrc, 0);  // the next item's chunk 0 lands during the epilogue

    // ---- epilogue: pre-activations -> gates -> cell update ------------------------
    const unsigned xs = lds0 + 2 * kLsStageBytes + kLsScratchTail;
    if (tid < kSplitRows) {
#pragma unroll
      for (int i = 0; i < DIN; ++i) lds_write_b32(xs + (i * kSplitRows + tid) * 4, xr[i]);
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    const int lane_now = lane_id();
    const int l32e = lane_now & 31, hhe = lane_now >> 5;
    const int col4 = (kLsUnits * ub + 32 * wc + l32e) * 4;  // byte offset of this lane's unit in a [..][256] row
    const __amdgpu_buffer_rsrc_t crsrc =
        buffer_rsrc(args.c_prev + r0 * args.c_prev_pitch, (uint32_t)(((rows - 1) * args.c_prev_pitch + kHidden) * 4));
    const __amdgpu_buffer_rsrc_t hrsrc =
        buffer_rsrc(args.h_out + r0 * args.h_out_pitch, (uint32_t)(((rows - 1) * args.h_out_pitch + kHidden) * 4));
    const __amdgpu_buffer_rsrc_t cors =
        buffer_rsrc(args.c_out + r0 * args.c_out_pitch, (uint32_t)(((rows - 1) * args.c_out_pitch + kHidden) * 4));
    const __amdgpu_buffer_rsrc_t grsrc = buffer_rsrc(
        SAVE ? args.gates + r0 * args.gates_pitch : nullptr, (uint32_t)(((rows - 1) * args.gates_pitch + 4 * kHidden) * 4));
    // row pitches in bytes (scalar); per array the lane part (unit column + this lane's
    // 64 wr + 4 hh rows) is ONE vector offset, the rest of the row index is scalar
    const int cp4 = (int)args.c_prev_pitch * 4, hp4 = (int)args.h_out_pitch * 4, co4 = (int)args.c_out_pitch * 4,
              gp4 = (int)args.gates_pitch * 4;
    const int lane_rows = 64 * wr + 4 * hhe;
    const int v_cp = col4 + lane_rows * cp4, v_h = col4 + lane_rows * hp4, v_co = col4 + lane_rows * co4,
              v_g = col4 + lane_rows * gp4;
    // c_{t-1} of all thirty-two (row, unit) pairs of this lane, requested up front: fetched
    // per group of four rows, their HBM round trip sat in front of every group (eight per
    // item, ~1.5 us each of a 50 us item)
    [[maybe_unused]] float diag_sum = 0.0f;
    // h_t of this wave's [64 rows][32 units] block also goes through LDS (chunk stage 1 is
    // dead from the barrier of step 15 to the barrier that ends this epilogue; row pitch 144 B)
    // (+ the tail behind it) when the caller wants it as fp16 planes for the next timestep: written as the
    // accumulators hold it (lane = unit), read back lane = row, eight units = one fragment.
    constexpr int kHPitch = kLsHPitch;
    static_assert(4 * 64 * kHPitch <= kLsStageBytes + kLsScratchTail, "h scratch = stage 1 + the tail");
    const bool want_planes = args.planes_out != nullptr;
    const unsigned h_scr = lds0 + kLsStageBytes + wave * (64 * kHPitch);
    const unsigned h_scr_w = h_scr + 4 * hhe * kHPitch + l32e * 4;
    float cprev[2][16];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) cprev[mt][r] = buffer_load_f32(crsrc, v_cp, (32 * mt + 8 * (r >> 2) + (r & 3)) * cp4);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {  // four rows at a time: r = 4 rg + e, row = 8 rg + e + 4 hh
        const int row0 = 64 * wr + 32 * mt + 8 * rg + 4 * hhe;
        // the rows' observations: one 16-byte read per input
        u32x4 xq[DIN];
#pragma unroll
        for (int i = 0; i < DIN; ++i)
          xq[i] = i == 0   ? lds_read_b128<0 * kSplitRows * 4>(xs + row0 * 4)
                  : i == 1 ? lds_read_b128<1 * kSplitRows * 4>(xs + row0 * 4)
                  : i == 2 ? lds_read_b128<2 * kSplitRows * 4>(xs + row0 * 4)
                  : i == 3 ? lds_read_b128<3 * kSplitRows * 4>(xs + row0 * 4)
                  : i == 4 ? lds_read_b128<4 * kSplitRows * 4>(xs + row0 * 4)
                  : i == 5 ? lds_read_b128<5 * kSplitRows * 4>(xs + row0 * 4)
                           : lds_read_b128<6 * kSplitRows * 4>(xs + row0 * 4);
        float cp[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) cp[e] = cprev[mt][4 * rg + e];
#pragma unroll
        for (int i = 0; i < DIN; ++i) wait_lds<0>(xq[i]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * rg + e;
          float pre[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            float v = __builtin_fmaf(acc[mt][q][r], unscale, bias[q]);
#pragma unroll
            for (int i = 0; i < DIN; ++i) v = __builtin_fmaf(__uint_as_float(xq[i][e]), w_in[q][i], v);
            pre[q] = v;
          }
          float gi, gf, gg, go;
          if constexpr ((kLsDiag & 2) != 0) {
            gi = pre[0], gf = pre[1], gg = pre[2], go = pre[3];
          } else {
            gi = ls_sigmoid(pre[0]), gf = ls_sigmoid(pre[1]), gg = ls_tanh(pre[2]), go = ls_sigmoid(pre[3]);
          }
          const float c = __builtin_fmaf(gf, cp[e], gi * gg);
          const float h = (kLsDiag & 2) ? go * c : go * ls_tanh(c);
          const int srow = 32 * mt + 8 * rg + e;  // scalar part of the row index
          if constexpr ((kLsDiag & 1) != 0) {
            diag_sum += h + c + gi + gf + gg + go;
            continue;
          }
          if (want_planes) lds_write_b32(h_scr_w + srow * kHPitch, h);
          buffer_store_f32(h, hrsrc, v_h, srow * hp4);
          buffer_store_f32(c, cors, v_co, srow * co4);
          if constexpr (SAVE) {
            // (the saved gates are read once, by the backward pass, gigabytes later: streaming stores -- sc1 | nt; 0.5-1 %)
            buffer_store_f32_streaming(gi, grsrc, v_g, srow * gp4);
            buffer_store_f32_streaming(gf, grsrc, v_g + kHidden * 4, srow * gp4);
            buffer_store_f32_streaming(gg, grsrc, v_g + 2 * kHidden * 4, srow * gp4);
            buffer_store_f32_streaming(go, grsrc, v_g + 3 * kHidden * 4, srow * gp4);
          }
        }
      }
    }
    if constexpr ((kLsDiag & 1) != 0) buffer_store_f32(diag_sum, hrsrc, v_h, 0);
    if (want_planes) {
      // lane = row 64 wr + lane; fragment column j = units 8 j .. 8 j + 7 of this wave's 32:
      // k-step s = (8 ub + 4 wc + j) / 2, k-half (j & 1) [4 wc is even], row half wr
      unsigned char *dst = static_cast<unsigned char *>(args.planes_out) + tile * (int64_t)kLsATileBytes;
      const unsigned rd = h_scr + lane_now * kHPitch;
      u32x4 lo[4], hi[4];
      lo[0] = lds_read_b128<0>(rd), hi[0] = lds_read_b128<16>(rd);
      lo[1] = lds_read_b128<32>(rd), hi[1] = lds_read_b128<48>(rd);
      lo[2] = lds_read_b128<64>(rd), hi[2] = lds_read_b128<80>(rd);
      lo[3] = lds_read_b128<96>(rd), hi[3] = lds_read_b128<112>(rd);
      wait_lds<0>(lo[0], lo[1], lo[2], lo[3]);
      wait_lds<0>(hi[0], hi[1], hi[2], hi[3]);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float v[8] = {__uint_as_float(lo[j][0]), __uint_as_float(lo[j][1]), __uint_as_float(lo[j][2]), __uint_as_float(lo[j][3]),
                            __uint_as_float(hi[j][0]), __uint_as_float(hi[j][1]), __uint_as_float(hi[j][2]), __uint_as_float(hi[j][3])};
        u32x4 planes[2];
        ls_planes(v, kLsStateScale, planes);
        const int u8 = 8 * ub + 4 * wc + j;
        const int step = u8 >> 1, khj = u8 & 1;
#pragma unroll
        for (int p = 0; p < 2; ++p)
          *reinterpret_cast<u32x4 *>(dst + (step * 8 + (p * 2 + khj) * 2 + wr) * 1024 + lane_now * 16) = planes[p];
      }
      // every wave is done with the scratch before any wave's next step 0 writes stage 1
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    // (the next item's x is parked behind its own sixteen step barriers: no barrier needed for it)
  }
}

}  // namespace rl8

using namespace rl8;

// (round 6: every width the [w_ih | bias] rows hold -- d_in <= 7; 1, 2, 3, 5 until then)
RL8_API int rl8_lstm_split_supports(int d_in) { return d_in >= 1 && d_in <= kLsInCols - 1; }
RL8_API int64_t rl8_lstm_split_packed_bytes(void) { return kLsPackedBytes + 16; }
RL8_API int64_t rl8_lstm_split_wb_floats(void) { return 4 * kHidden * kLsInCols; }
// bytes of the state planes for b rows (whole 128-row tiles)
RL8_API int64_t rl8_lstm_split_state_bytes(int64_t b) { return ((b + kSplitRows - 1) / kSplitRows) * (int64_t)kLsATileBytes; }

RL8_API int rl8_lstm_pack_split(const float *w_ih, const float *w_hh, const float *b_ih, const float *b_hh, int d_in,
                                void *packed, float *wb, void *stream) {
  if (!w_ih || !w_hh || !b_ih || !b_hh || !packed || !wb) return RL8_ENULL;
  if (d_in < 1 || d_in > kLsInCols - 1) return RL8_ESIZE;
  if (!aligned16(packed) || !aligned16(w_hh)) return RL8_EALIGN;
  const int units = kLsBlocks * kSplitSteps * 8 * 64;  // 32 768 >= 1024 * 8
  lstm_whh_scale_kernel<<<1, 1024, 0, (hipStream_t)stream>>>(
      w_hh, reinterpret_cast<float *>(static_cast<unsigned char *>(packed) + kLsPackedBytes));
  lstm_pack_split_kernel<<<(units + kBlock - 1) / kBlock, kBlock, 0, (hipStream_t)stream>>>(
      w_ih, w_hh, b_ih, b_hh, d_in, static_cast<uint32_t *>(packed), wb);
  return launch_status();
}

static int split_state(const float *h, int64_t pitch, int64_t b, void *planes, uint32_t *bound_out, void *stream) {
  if (!h || !planes) return RL8_ENULL;
  if (b <= 0 || pitch < kHidden) return RL8_ESIZE;
  if (!aligned16(h) || !aligned16(planes) || (pitch & 3)) return RL8_EALIGN;
  const int64_t tiles = (b + kSplitRows - 1) / kSplitRows;
  const int grid = (int)(tiles < kMaxGrid ? tiles : kMaxGrid);
  if (bound_out && hipMemsetAsync(bound_out, 0, 4, (hipStream_t)stream) != hipSuccess) return launch_status();
  lstm_split_state_kernel<<<grid, kBlock, 0, (hipStream_t)stream>>>(h, pitch, b, static_cast<uint32_t *>(planes), bound_out);
  return launch_status();
}

RL8_API int rl8_lstm_split_state(const float *h, int64_t pitch, int64_t b, void *planes, void *stream) {
  return split_state(h, pitch, b, planes, nullptr, stream);
}

RL8_API int rl8_lstm_split_state_bound(const float *h, int64_t pitch, int64_t b, void *planes, uint32_t *bound_out,
                                       void *stream) {
  if (!bound_out) return RL8_ENULL;
  return split_state(h, pitch, b, planes, bound_out, stream);
}

template <int DIN>
static int launch_lstm_step(int grid, hipStream_t s, const void *a_planes, const void *w_planes, const float *wb, int64_t b,
                            const LstmStepArgs &args) {
  static LdsOptIn lds_attr_set_0;
  if (const int e_lds_attr_set_0 = allow_dynamic_lds(lds_attr_set_0, reinterpret_cast<const void *>(&lstm_step_split_kernel<DIN, true>), 160 * 1024)) return e_lds_attr_set_0;
  static LdsOptIn lds_attr_set_1;
  if (const int e_lds_attr_set_1 = allow_dynamic_lds(lds_attr_set_1, reinterpret_cast<const void *>(&lstm_step_split_kernel<DIN, false>), 160 * 1024)) return e_lds_attr_set_1;
  if (args.gates)
    lstm_step_split_kernel<DIN, true><<<grid, kBlock, lstm_step_lds_bytes(), s>>>(a_planes, w_planes, wb, b, args);
  else
    lstm_step_split_kernel<DIN, false><<<grid, kBlock, lstm_step_lds_bytes(), s>>>(a_planes, w_planes, wb, b, args);
  return launch_status();
}

RL8_API int rl8_lstm_step_split_f32(const float *x, int64_t x_pitch, int d_in, const void *h_planes,
                                    const float *c_prev, int64_t c_prev_pitch, const void *w_planes,
                                    const float *wb, int64_t b, float *h_out, int64_t h_out_pitch, float *c_out,
                                    int64_t c_out_pitch, float *gates, int64_t gates_pitch, void *planes_out,
                                    void *stream) {
  if (!x || !h_planes || !c_prev || !w_planes || !wb || !h_out || !c_out) return RL8_ENULL;
  if (b <= 0 || !rl8_lstm_split_supports(d_in)) return RL8_ESIZE;
  if (x_pitch < d_in || c_prev_pitch < kHidden || h_out_pitch < kHidden || c_out_pitch < kHidden ||
      (gates && gates_pitch < 4 * kHidden))
    return RL8_ESIZE;
  // a tile's rows are addressed with 32-bit byte offsets
  const int64_t widest = gates ? gates_pitch : (h_out_pitch > c_prev_pitch ? h_out_pitch : c_prev_pitch);
  if ((int64_t)kSplitRows * widest * 4 >= (int64_t)1 << 31) return RL8_ESIZE;
  if (!aligned16(h_planes) || !aligned16(w_planes)) return RL8_EALIGN;
  const int64_t tiles = (b + kSplitRows - 1) / kSplitRows;
  const int64_t items = tiles * kLsBlocks;
  const int grid = (int)(items < 2 * kCUs ? (items + 31) / 32 * 32 : 2 * kCUs);  // a multiple of 32: see the kernel's tile mapping
  if (planes_out && (!aligned16(planes_out) || planes_out == h_planes)) return RL8_EALIGN;
  const LstmStepArgs args = {x, c_prev, h_out, c_out, gates, planes_out, x_pitch, c_prev_pitch, h_out_pitch, c_out_pitch,
                             gates_pitch};
  hipStream_t s = (hipStream_t)stream;
  switch (d_in) {
    case 1: return launch_lstm_step<1>(grid, s, h_planes, w_planes, wb, b, args);
    case 2: return launch_lstm_step<2>(grid, s, h_planes, w_planes, wb, b, args);
    case 3: return launch_lstm_step<3>(grid, s, h_planes, w_planes, wb, b, args);
    case 4: return launch_lstm_step<4>(grid, s, h_planes, w_planes, wb, b, args);
    case 5: return launch_lstm_step<5>(grid, s, h_planes, w_planes, wb, b, args);
    case 6: return launch_lstm_step<6>(grid, s, h_planes, w_planes, wb, b, args);
    default: return launch_lstm_step<7>(grid, s, h_planes, w_planes, wb, b, args);
  }
}
