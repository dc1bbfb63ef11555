// a-9 (recurrent PPO), backward through time on the 16-bit matrix pipe (round 3).
// What lstm_kernels.hip's lstm_backward_kernel computes -- per step, from dL/dh_t (the heads' gradient plus what flows
// back from t + 1) and the carried dL/dc_t (torch.nn.LSTM's backward; src/rl8/models/_recurrent.py:201-321 builds the
// module, src/rl8/algorithms/_recurrent.py runs loss.backward() through it),
//   do = dh * tanh(c_t) * o(1-o)            dc  = dh * o * (1 - tanh^2 c_t) + dc_carry
//   di = dc * g * i(1-i)     dg = dc * i * (1 - g^2)     df = dc * c_{t-1} * f(1-f)
//   dc_carry = dc * f        dh_carry = sum_q dgate_q x W_hh[q]
// and the pre-activation gate gradients dG [b][l][4][256] stored for the weight-gradient kernels -- with the recurrent
// product as an fp32-accurate product of bf16 planes (three per operand, six plane products: split_tile.hip.h; bf16
// because dG has fp32's range and no bound that is known before it is computed) instead of fp32 MFMAs on 32-row tiles
// (11.65 ms per 2^21 row-steps, 60 % of the fp32 matrix peak, W_hh^T re-read from L2 per 32 rows).
//
// Shape: rows per wave, as mlp_rows_kernels.hip.  A wave owns 32 sequences for all l steps; a workgroup is four waves
// (128 sequences), one per SIMD, one workgroup per CU.  The product is taken TRANSPOSED, dh_{t-1}^T = W_hh^T x dG^T:
//   A = W_hh^T planes, out unit x gate row, from a ring of 24-KiB chunks in LDS that the four waves fill together with
//       direct-to-LDS loads and read as fragments (the only operand that goes through LDS);
//   B = dG^T planes, gate row x sequence: a lane's B fragment is eight gate rows of ITS OWN sequence (lane & 31), which
//       it has just computed from its own loads -- no exchange, no LDS;
//   D = dh_{t-1}^T: lane = sequence again, registers = out units.  With the out units of a tile permuted (lr_out_unit)
//       the sixteen registers of tile mo are units 32 mo + 16 (r >> 3) + 8 (lane >> 5) + (r & 7): exactly the eight-unit
//       runs this lane needs as the NEXT step's dh_carry when it computes gate rows 16 c + 8 (lane >> 5) + e of chunk
//       c = 2 mo + (r >> 3).  So dh never leaves the registers between steps, and every global access of a lane is two
//       16-byte pieces of one row (lanes n and n + 32 together: one 64-byte sector).
// k order of a step: unit chunk c = 0..15 (sixteen hidden units), gates o, i, g, f within it: 64 gate-steps of 16 k,
// each 8 out tiles x 6 plane products = 48 MFMAs (32x32x16).  The slot of a gate-step in the four-slot ring is its
// gate position, so the chunk loop is a runtime loop over c with a fixed body.
// Carries: dh in registers (copied out of the accumulators once per step, read per chunk by a dynamic register index
// c); dc through a [b][256] scratch in HBM (2 KiB of traffic per row-step; in registers it would take the 128 the row
// loads in flight need: 512 per lane = 128 accumulators + 128 dh + loads, planes, fragments).
// HBM per row-step: gates 4 KiB, c_t, c_{t-1}, dh_t 1 KiB each, dc in/out 2 KiB, dG 4 KiB out = 13 KiB (the fp32 kernel
// moves 11); matrix pipe 3072 cycles per row-step.
#include "split_tile.hip.h"

namespace rl8 {

constexpr int kLrRows = 128;                 // sequences per workgroup (32 per wave)
constexpr int kLrChunks = 16;                // unit chunks per step
constexpr int kLrGateSteps = 4 * kLrChunks;  // k-chunks of 16 per step
constexpr int kLrSlotBytes = 3 * 8 * 1024;   // one gate-step of W_hh^T: [plane][out tile] x 1 KiB
constexpr int kLrRing = 4;
constexpr int kLrPackedBytes = kLrGateSteps * kLrSlotBytes;  // 1.5 MiB
constexpr int kLrLdsBytes = kLrRing * kLrSlotBytes;
constexpr int kLrRowLoads = 16;              // 16-byte loads a lane issues per chunk
constexpr int kLrStores = 10;                // 16-byte stores per chunk: dG 8, dc 2
constexpr int kLrDma = 6;                    // 1-KiB direct-to-LDS loads per wave and gate-step
#ifndef RL8_LR_DIAG
#define RL8_LR_DIAG 0  // tuning builds (tools/diag_mlp.sh lr<bits>): 1 no stores reach memory, 2 no row loads do, 4 one plane
#endif                 // product of six, 8 no W_hh^T traffic (wrong results, same instruction stream)
constexpr int kLrDiag = RL8_LR_DIAG;
#ifndef RL8_LR_SAFE_WAITS
#define RL8_LR_SAFE_WAITS 0                  // tuning builds: 1 = every barrier behind vmcnt(0)
#endif

// gate position within a unit chunk -> gate of torch's [i | f | g | o] layout
__host__ __device__ constexpr int lr_gate(int pos) { return pos == 0 ? 3 : pos == 1 ? 0 : pos == 2 ? 2 : 1; }
// row m of out tile mo of the transposed product -> hidden unit (see the header: makes a lane's registers runs of eight)
__host__ __device__ constexpr int lr_out_unit(int mo, int m) {
  return 32 * mo + 16 * (m >> 4) + 8 * ((m >> 2) & 1) + 4 * ((m >> 3) & 1) + (m & 3);
}

__device__ __forceinline__ float lr_tanh(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(__expf(2.0f * x) + 1.0f); }

// W_hh [1024][256] -> bf16 planes of W_hh^T in fragment order:
// packed[gs][plane][mo][lane][e] = plane of W_hh[256 q + 16 c + 8 (lane >> 5) + e][lr_out_unit(mo, lane & 31)],
// gs = 4 c + pos, q = lr_gate(pos).
__global__ __launch_bounds__(kBlock) void lstm_rows_pack_kernel(const float *__restrict__ w_hh, uint32_t *__restrict__ packed) {
  const int idx = blockIdx.x * kBlock + threadIdx.x;  // (gs, mo, lane)
  if (idx >= kLrGateSteps * 8 * 64) return;
  const int lane = idx & 63, mo = (idx >> 6) & 7, gs = idx >> 9;
  const int q = lr_gate(gs & 3), c = gs >> 2;
  const int out = lr_out_unit(mo, lane & 31);
  const float *src = w_hh + (int64_t)(kHidden * q + 16 * c + 8 * (lane >> 5)) * kHidden + out;
  u32x4 planes[3];
#pragma unroll
  for (int e = 0; e < 8; e += 2) {
    uint32_t hi, mid, lo;
    split_pair(src[e * kHidden], src[(e + 1) * kHidden], hi, mid, lo);
    planes[0][e >> 1] = hi;
    planes[1][e >> 1] = mid;
    planes[2][e >> 1] = lo;
  }
#pragma unroll
  for (int p = 0; p < 3; ++p)
    *reinterpret_cast<u32x4 *>(reinterpret_cast<unsigned char *>(packed) + (int64_t)gs * kLrSlotBytes + (p * 8 + mo) * 1024 + lane * 16) =
        planes[p];
}

struct LrArgs {
  const float *c0;     // [b][256]
  const float *gates;  // [b][l][4][256] post-activation i, f, g, o
  const float *cs;     // [b][l][256]
  const float *dhs;    // [b][l][256]
  float *dgates;       // [b][l][4][256]
  float *dc;           // [b][256] scratch: the carried dL/dc between steps
  int64_t b;
  int l;
};

typedef float f32x16v __attribute__((ext_vector_type(16)));

// what a lane loads for one unit chunk: eight consecutive units of its sequence from each array
struct LrLoads {
  u32x4 g[4][2], ct[2], cp[2], dh[2], dc[2];
};
struct LrLoadDesc {
  __amdgpu_buffer_rsrc_t gates, cs, cprev, dhs, dcin;
  int v_cp;  // this lane's offset into cprev (cs of t - 1: sequence pitch; c0: 1 KiB)
};
struct LrStoreDesc {
  __amdgpu_buffer_rsrc_t dgates, dcout;
};

__global__ __launch_bounds__(kBlock, 1) void lstm_rows_backward_kernel(LrArgs a, const void *__restrict__ w_planes) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const unsigned lds0 = lds_offset(smem);
  const int tid = threadIdx.x, lane = tid & 63, n = lane & 31, hh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l = a.l;
  const __amdgpu_buffer_rsrc_t wrsrc = buffer_rsrc(w_planes, (kLrDiag & 8) ? 0 : kLrPackedBytes);
  const unsigned a_read = lds0 + lane * 16;

  // per-lane byte offsets: sequence n of the wave's 32, units 8 hh .. 8 hh + 7 of a chunk
  const int v_gates = n * l * (4 * kHidden * 4) + hh * 32;
  const int v_seq = n * l * (kHidden * 4) + hh * 32;
  const int v_state = n * (kHidden * 4) + hh * 32;

  auto request = [&](int gs, int slot) {
#pragma unroll
    for (int u = 0; u < kLrDma; ++u) {
      const int block = wave * kLrDma + u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, smem + slot * kLrSlotBytes + block * 1024, 16, lane * 16,
                                               gs * kLrSlotBytes + block * 1024, 0, 0);
    }
  };

  const int64_t tiles = (a.b + kLrRows - 1) / kLrRows;
  // descriptors of the wave's 32 sequences at (tile, t); a wave past the end gets zero records: loads give 0.0, stores
  // are dropped, the instruction counts the barriers rely on stay the same
  auto wave_rows = [&](int64_t tile) {
    const int64_t left = tile < tiles ? a.b - (tile * kLrRows + 32 * wave) : 0;
    return (int)(left < 0 ? 0 : left > 32 ? 32 : left);
  };
  auto seq_rsrc = [&](const float *base, int64_t tile, int t, int per_row, int rows) {
    const int64_t r0 = tile * kLrRows + 32 * wave;
    return buffer_rsrc(rows > 0 ? base + (r0 * l + t) * (int64_t)per_row : base,
                       rows > 0 ? (uint32_t)(((rows - 1) * l + 1) * per_row * 4) : 0u);
  };
  auto state_rsrc = [&](const float *base, int64_t tile, int rows) {
    const int64_t r0 = tile * kLrRows + 32 * wave;
    return buffer_rsrc(rows > 0 ? base + r0 * kHidden : base, rows > 0 ? (uint32_t)(rows * kHidden * 4) : 0u);
  };
  auto load_desc = [&](int64_t tile, int t) {
    const int rows = (kLrDiag & 2) ? 0 : wave_rows(tile);
    LrLoadDesc d;
    d.gates = seq_rsrc(a.gates, tile, t, 4 * kHidden, rows);
    d.cs = seq_rsrc(a.cs, tile, t, kHidden, rows);
    d.dhs = seq_rsrc(a.dhs, tile, t, kHidden, rows);
    d.cprev = t > 0 ? seq_rsrc(a.cs, tile, t - 1, kHidden, rows) : state_rsrc(a.c0, tile, rows);
    d.v_cp = t > 0 ? v_seq : v_state;
    d.dcin = state_rsrc(a.dc, tile, t == l - 1 ? 0 : rows);  // the last step of a sequence starts from dc = 0
    return d;
  };
  auto store_desc = [&](int64_t tile, int t) {
    const int rows = (kLrDiag & 1) ? 0 : wave_rows(tile);
    LrStoreDesc d;
    d.dgates = seq_rsrc(a.dgates, tile, t, 4 * kHidden, rows);
    d.dcout = state_rsrc(a.dc, tile, rows);
    return d;
  };
  auto issue_loads = [&](LrLoads &ld, const LrLoadDesc &d, int c) {
    const int soff = c * 64;
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        ld.g[q][h2] = __builtin_amdgcn_raw_buffer_load_b128(d.gates, v_gates + h2 * 16, soff + q * (kHidden * 4), 0);
      ld.ct[h2] = __builtin_amdgcn_raw_buffer_load_b128(d.cs, v_seq + h2 * 16, soff, 0);
      ld.cp[h2] = __builtin_amdgcn_raw_buffer_load_b128(d.cprev, d.v_cp + h2 * 16, soff, 0);
      ld.dh[h2] = __builtin_amdgcn_raw_buffer_load_b128(d.dhs, v_seq + h2 * 16, soff, 0);
      ld.dc[h2] = __builtin_amdgcn_raw_buffer_load_b128(d.dcin, v_state + h2 * 16, soff, 0);
    }
  };

  // dh carried into the step being computed: dhv[e][c] = dL/dh of unit 16 c + 8 hh + e from the step after it
  f32x16v dhv[8];
  f32x16 acc[8];
  float dg[4][8];  // [gate position][e]: dG of the chunk whose matrix work comes next

  // the gate arithmetic of chunk c of the step `sd` stores for: loads -> dg, stores of dG and dc
  auto gate_math = [&](const LrLoads &ld, const LrStoreDesc &sd, int c) {
    float dc_out[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int h2 = e >> 2, j = e & 3;
      const float gi = __uint_as_float(ld.g[0][h2][j]), gf = __uint_as_float(ld.g[1][h2][j]),
                  gg = __uint_as_float(ld.g[2][h2][j]), go = __uint_as_float(ld.g[3][h2][j]);
      const float ct = __uint_as_float(ld.ct[h2][j]), cp = __uint_as_float(ld.cp[h2][j]);
      const float dh = __uint_as_float(ld.dh[h2][j]) + dhv[e][c];
      const float tc = lr_tanh(ct);
      const float d_o = dh * tc * (go * (1.0f - go));
      const float dc = __builtin_fmaf(dh * go, 1.0f - tc * tc, __uint_as_float(ld.dc[h2][j]));
      dg[0][e] = d_o;
      dg[1][e] = dc * gg * (gi * (1.0f - gi));
      dg[2][e] = dc * gi * (1.0f - gg * gg);
      dg[3][e] = dc * cp * (gf * (1.0f - gf));
      dc_out[e] = dc * gf;
    }
    const int soff = c * 64;
#pragma unroll
    for (int pos = 0; pos < 4; ++pos)
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2)
        __builtin_amdgcn_raw_buffer_store_b128(
            u32x4{__float_as_uint(dg[pos][4 * h2]), __float_as_uint(dg[pos][4 * h2 + 1]), __float_as_uint(dg[pos][4 * h2 + 2]),
                  __float_as_uint(dg[pos][4 * h2 + 3])},
            sd.dgates, v_gates + h2 * 16, soff + lr_gate(pos) * (kHidden * 4), 0);
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2)
      __builtin_amdgcn_raw_buffer_store_b128(
          u32x4{__float_as_uint(dc_out[4 * h2]), __float_as_uint(dc_out[4 * h2 + 1]), __float_as_uint(dc_out[4 * h2 + 2]),
                __float_as_uint(dc_out[4 * h2 + 3])},
          sd.dcout, v_state + h2 * 16, soff, 0);
  };

  // one gate-step: 16 k of the product, W_hh^T planes from ring slot POS, B planes from dg[POS].  Out tiles in pairs:
  // two accumulators alternate, so no MFMA waits for the one before it.
  auto matrix_step = [&](auto pos_tag) {
    constexpr int POS = decltype(pos_tag)::value;
    u32x4 bh, bm, bl;
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
      uint32_t hi, mid, lo;
      split_pair(dg[POS][e], dg[POS][e + 1], hi, mid, lo);
      bh[e >> 1] = hi;
      bm[e >> 1] = mid;
      bl[e >> 1] = lo;
    }
    const unsigned ar = a_read + POS * kLrSlotBytes;
    u32x4 ah[2][2], am[2][2], al[2][2];  // [buffer][tile of the pair]
    auto fetch = [&](int mp, int s) {
      ah[s][0] = mp == 0 ? lds_read_b128<0 * 1024>(ar) : mp == 1 ? lds_read_b128<2 * 1024>(ar) : mp == 2 ? lds_read_b128<4 * 1024>(ar) : lds_read_b128<6 * 1024>(ar);
      ah[s][1] = mp == 0 ? lds_read_b128<1 * 1024>(ar) : mp == 1 ? lds_read_b128<3 * 1024>(ar) : mp == 2 ? lds_read_b128<5 * 1024>(ar) : lds_read_b128<7 * 1024>(ar);
      am[s][0] = mp == 0 ? lds_read_b128<8 * 1024>(ar) : mp == 1 ? lds_read_b128<10 * 1024>(ar) : mp == 2 ? lds_read_b128<12 * 1024>(ar) : lds_read_b128<14 * 1024>(ar);
      am[s][1] = mp == 0 ? lds_read_b128<9 * 1024>(ar) : mp == 1 ? lds_read_b128<11 * 1024>(ar) : mp == 2 ? lds_read_b128<13 * 1024>(ar) : lds_read_b128<15 * 1024>(ar);
      al[s][0] = mp == 0 ? lds_read_b128<16 * 1024>(ar) : mp == 1 ? lds_read_b128<18 * 1024>(ar) : mp == 2 ? lds_read_b128<20 * 1024>(ar) : lds_read_b128<22 * 1024>(ar);
      al[s][1] = mp == 0 ? lds_read_b128<17 * 1024>(ar) : mp == 1 ? lds_read_b128<19 * 1024>(ar) : mp == 2 ? lds_read_b128<21 * 1024>(ar) : lds_read_b128<23 * 1024>(ar);
    };
    fetch(0, 0);
#pragma unroll
    for (int mp = 0; mp < 4; ++mp) {
      const int s = mp & 1;
      // everything outstanding is this pair's six fragments (lgkmcnt(0): a stray scalar load cannot spoil the count);
      // the next pair's are requested behind the wait and land under this pair's twelve MFMAs
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(ah[s][0]), "+v"(ah[s][1]), "+v"(am[s][0]), "+v"(am[s][1]), "+v"(al[s][0]), "+v"(al[s][1]));
      if (mp < 3) fetch(mp + 1, s ^ 1);
      f32x16 d0 = acc[2 * mp], d1 = acc[2 * mp + 1];
#define RL8_LR_MMA(A, B)                                                                                              \
  d0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A[s][0]), __builtin_bit_cast(bf16x8, B), d0, 0, 0, 0); \
  d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A[s][1]), __builtin_bit_cast(bf16x8, B), d1, 0, 0, 0)
      if constexpr ((kLrDiag & 4) == 0) {
        RL8_LR_MMA(al, bh);  // smallest terms first
        RL8_LR_MMA(ah, bl);
        RL8_LR_MMA(am, bm);
        RL8_LR_MMA(am, bh);
        RL8_LR_MMA(ah, bm);
      }
      RL8_LR_MMA(ah, bh);
#undef RL8_LR_MMA
      acc[2 * mp] = d0;
      acc[2 * mp + 1] = d1;
    }
  };

  // The barrier that opens gate-step gs: this wave's share of chunk gs has landed (vector-memory operations complete in
  // issue order: all but the N youngest are done) and every wave is through with the slot the next request overwrites.
  // N = what the wave has issued behind its request for chunk gs, three gate-steps ago: per position of gs in its unit
  // chunk, with the order of a chunk's operations [row loads (16) | request (6)] [request] [request] [request | stores (10)].
  auto open_step = [&](auto pos_tag) {
    constexpr int POS = decltype(pos_tag)::value;
    constexpr int N = RL8_LR_SAFE_WAITS ? 0
                      : POS == 0 ? 2 * kLrDma + kLrStores
                      : POS == 1 ? 2 * kLrDma + kLrStores + kLrRowLoads
                      : POS == 2 ? kLrStores + kLrRowLoads + 2 * kLrDma
                                 : 2 * kLrDma;
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
  };
  using P0 = std::integral_constant<int, 0>;
  using P1 = std::integral_constant<int, 1>;
  using P2 = std::integral_constant<int, 2>;
  using P3 = std::integral_constant<int, 3>;

  int64_t tile = blockIdx.x;
  if (tile >= tiles) return;  // (the host launches no more workgroups than tiles)
  int t = l - 1;
  LrLoads ld;
  LrStoreDesc sd = store_desc(tile, t);
#pragma unroll
  for (int e = 0; e < 8; ++e) dhv[e] = f32x16v{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  {
    const LrLoadDesc d0 = load_desc(tile, t);
    issue_loads(ld, d0, 0);
    asm volatile("" ::: "memory");
    request(0, 0);
    request(1, 1);
    request(2, 2);
    gate_math(ld, sd, 0);
  }
  LrLoadDesc nd = load_desc(tile, t);  // where the NEXT chunk's loads come from

  while (true) {
    // position behind this step: the same sequences one step earlier, or the next tile's last step, or nothing
    const bool last_step = t == 0;
    const int64_t ntile = last_step ? tile + gridDim.x : tile;
    const int nt = last_step ? l - 1 : t - 1;
    const LrLoadDesc after = load_desc(ntile, nt);
#pragma unroll 1
    for (int c = 0; c < kLrChunks; ++c) {
      const int gs = 4 * c;
      const bool wrap = c == kLrChunks - 1;
      open_step(P0{});
      issue_loads(ld, wrap ? after : nd, wrap ? 0 : c + 1);
      asm volatile("" ::: "memory");
      request((gs + 3) & (kLrGateSteps - 1), 3);
      if (c == 0) {
#pragma unroll
        for (int mo = 0; mo < 8; ++mo) acc[mo] = f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
      }
      matrix_step(P0{});
      open_step(P1{});
      request((gs + 4) & (kLrGateSteps - 1), 0);
      matrix_step(P1{});
      open_step(P2{});
      request((gs + 5) & (kLrGateSteps - 1), 1);
      matrix_step(P2{});
      open_step(P3{});
      request((gs + 6) & (kLrGateSteps - 1), 2);
      matrix_step(P3{});
      if (wrap) {
        // the step's dh is complete: it becomes the carry of the step the next arithmetic belongs to (zero for a new tile)
#pragma unroll
        for (int cc = 0; cc < kLrChunks; ++cc)
#pragma unroll
          for (int e = 0; e < 8; ++e) dhv[e][cc] = last_step ? 0.0f : acc[cc >> 1][8 * (cc & 1) + e];
        sd = store_desc(ntile, nt);
      }
      gate_math(ld, sd, wrap ? 0 : c + 1);
    }
    if (last_step && ntile >= tiles) break;
    tile = ntile;
    t = nt;
    nd = after;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no request may still be writing LDS when the workgroup ends
}

}  // namespace rl8

using namespace rl8;

RL8_API int64_t rl8_lstm_rows_backward_pack_bytes(void) { return kLrPackedBytes; }

RL8_API int rl8_lstm_rows_backward_pack(const float *w_hh, void *packed, void *stream) {
  if (!w_hh || !packed) return RL8_ENULL;
  if (!aligned16(packed)) return RL8_EALIGN;
  lstm_rows_pack_kernel<<<kLrGateSteps * 8 * 64 / kBlock, kBlock, 0, (hipStream_t)stream>>>(w_hh, static_cast<uint32_t *>(packed));
  return launch_status();
}

RL8_API int rl8_lstm_rows_backward_f32(int64_t b, int l, const float *c0, const float *gates, const float *cs, const float *dhs,
                                       const void *packed, float *dgates, float *dc_scratch, void *stream) {
  if (!c0 || !gates || !cs || !dhs || !packed || !dgates || !dc_scratch) return RL8_ENULL;
  if (b <= 0 || l <= 0) return RL8_ESIZE;
  // a wave addresses its 32 sequences with 32-bit offsets
  if ((int64_t)32 * l * 4 * kHidden * 4 >= (int64_t)1 << 31) return RL8_ESIZE;
  if (!aligned16(packed) || !aligned16(c0) || !aligned16(gates) || !aligned16(cs) || !aligned16(dhs) || !aligned16(dgates) ||
      !aligned16(dc_scratch))
    return RL8_EALIGN;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&lstm_rows_backward_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, kLrLdsBytes);
    (void)hipGetLastError();
    attr_set = true;
  }
  const int64_t tiles = (b + kLrRows - 1) / kLrRows;
  const int grid = (int)(tiles < kCUs ? tiles : kCUs);
  const LrArgs args = {c0, gates, cs, dhs, dgates, dc_scratch, b, l};
  lstm_rows_backward_kernel<<<grid, kBlock, kLrLdsBytes, (hipStream_t)stream>>>(args, packed);
  return launch_status();
}
