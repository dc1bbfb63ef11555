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
//   D = dh_{t-1}^T: lane = sequence again, registers = out units: register r of tile mo is unit 32 mo + 8 (r >> 2) +
//       4 (lane >> 5) + (r & 3).  Those sixteen units are exactly what this lane needs as the NEXT step's dh_carry when
//       it computes the gate rows of chunk mo (its B fragments are made of the same sixteen units, in the same register
//       order: lr_in_unit), so dh never leaves the registers between steps; and four consecutive registers are four
//       consecutive units: 16 bytes of a row, next to the 16 of lane + 32.
// k order of a step: unit chunk c = 0..7 (the 32 hidden units of out tile c: a lane's sixteen are the registers of
// accumulator c), gates o, i, g, f within it, two k-steps of 16 per gate: 64 gate-steps of 16 k, each 8 out tiles x 6
// plane products = 48 MFMAs (32x32x16).  The slot of a gate-step in the four-slot ring is its position in the chunk
// mod 4, so the chunk loop is a runtime loop over c with a fixed body of eight gate-steps.
// Row accesses are whole 128-byte lines: per array and chunk a lane reads its sixteen units with four 16-byte loads
// issued together, and lanes n, n + 32 cover the line between them.  (The first version worked on chunks of sixteen
// units -- half a line now, the other half a chunk-time later -- and ran at the 2.7 TB/s HBM gives 64-byte pieces:
// tools/probes/piece_size_probe.hip; 128-byte pieces get 4.2.)  To fit the registers the chunk's loads come in three
// phases, each requested five or more gate-steps before its arithmetic: A = {o, c_t, dh_t, dc} -> dG_o, dc;
// B = {i, g} -> dG_i, dG_g; C = {f, c_{t-1}} -> dG_f, dc out.  A waits in registers; B and C are direct-to-LDS loads into
// a per-wave park (each lane reads back exactly the 16 bytes it asked for) and cost no registers while in flight.
// Carries: dh in registers (copied out of the accumulators once per step, read per chunk by a dynamic register index
// c); dc through a [b][256] scratch in HBM (2 KiB of traffic per row-step; in registers it would take the 128 the row
// loads in flight need: 512 per lane = 128 accumulators + 128 dh + loads, dG, planes, fragments).
// HBM per row-step: gates 4 KiB, c_t, c_{t-1}, dh_t 1 KiB each, dc in/out 2 KiB, dG 4 KiB out = 13 KiB (the fp32 kernel
// moves 11); matrix pipe 3072 cycles per row-step.
#include "split_tile.hip.h"

namespace rl8 {

constexpr int kLrRows = 128;                 // sequences per workgroup (32 per wave)
constexpr int kLrChunks = 8;                 // unit chunks (out tiles) per step
constexpr int kLrGateSteps = 8 * kLrChunks;  // k-chunks of 16 per step
constexpr int kLrSlotBytes = 3 * 8 * 1024;   // one gate-step of W_hh^T: [plane][out tile] x 1 KiB
constexpr int kLrRing = 4;
constexpr int kLrPackedBytes = kLrGateSteps * kLrSlotBytes;  // 1.5 MiB
constexpr int kLrStageBytes = 4 * 8 * 1024;  // one load phase of two arrays, parked in LDS: [wave][array][piece] x 1 KiB
constexpr int kLrLdsBytes = kLrRing * kLrSlotBytes + 2 * kLrStageBytes;  // 160 KiB: the whole CU
constexpr int kLrDma = 6;                    // 1-KiB direct-to-LDS loads per wave and gate-step
// 16-byte row operations a lane issues in gate-step k of a chunk (k = 2 pos + half): loads in front of the step's
// request, stores behind its matrix work
constexpr int kLrLoadsAt[8] = {8, 0, 16, 0, 8, 0, 0, 0};   // C of this chunk | A of the next | B of the next
constexpr int kLrStoresAt[8] = {0, 8, 0, 0, 0, 8, 0, 4};   // dG_i, dG_g | dG_f, dc | dG_o of the next chunk
// operations a wave has issued behind its request for gate-step k's chunk of W_hh^T (made in step k - 3)
constexpr int lr_behind(int k) {
  return kLrStoresAt[(k + 5) & 7] + kLrLoadsAt[(k + 6) & 7] + kLrDma + kLrStoresAt[(k + 6) & 7] + kLrLoadsAt[(k + 7) & 7] +
         kLrDma + kLrStoresAt[(k + 7) & 7];
}
static_assert(lr_behind(0) == 24 && lr_behind(3) == 36 && lr_behind(6) == 28, "see the table in open_step");
// ... and behind the phase-B loads (made in step 4, read after step 1's matrix work) and the phase-C loads (step 0 -> 5);
// vmcnt has six bits: waiting for all but the 63 youngest covers anything further back
constexpr int lr_behind_loads(int from, int to) {  // from the loads of step `from` to the end of step `to`'s matrix work
  int n = kLrDma;
  for (int k = (from + 1) & 7;; k = (k + 1) & 7) {
    n += kLrStoresAt[(k + 7) & 7] + kLrLoadsAt[k] + kLrDma;
    if (k == to) break;
  }
  return n < 63 ? n : 63;
}
constexpr int kLrBehindB = lr_behind_loads(4, 1);
constexpr int kLrBehindC = lr_behind_loads(0, 5);
static_assert(kLrBehindB == 56 && kLrBehindC == 63, "see the table in open_step");
#ifndef RL8_LR_DIAG
#define RL8_LR_DIAG 0  // tuning builds (tools/diag_mlp.sh lr<bits>): 1 no stores reach memory, 2 no row loads do, 4 one plane
#endif                 // product of six, 8 no W_hh^T traffic (wrong results, same instruction stream)
constexpr int kLrDiag = RL8_LR_DIAG;
#ifndef RL8_LR_SAFE_WAITS
#define RL8_LR_SAFE_WAITS 0                  // tuning builds: 1 = every barrier behind vmcnt(0)
#endif

// gate position within a unit chunk -> gate of torch's [i | f | g | o] layout
__host__ __device__ constexpr int lr_gate(int pos) { return pos == 0 ? 3 : pos == 1 ? 0 : pos == 2 ? 2 : 1; }
// gate row (k) that slot e of k-half kb stands for in the 16-k step `c16` of a gate: the lane's B fragment of step
// c16 = 2 c + h is registers 8 h .. 8 h + 7 of its sixteen units of chunk c, and register r of a transposed 32 x 32
// accumulator is out unit 8 (r >> 2) + 4 (lane >> 5) + (r & 3) of the tile
__host__ __device__ constexpr int lr_in_unit(int c16, int kb, int e) { return 16 * c16 + 8 * (e >> 2) + 4 * kb + (e & 3); }

__device__ __forceinline__ float lr_tanh(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(__expf(2.0f * x) + 1.0f); }

// W_hh [1024][256] -> bf16 planes of W_hh^T in fragment order:
// packed[gs][plane][mo][lane][e] = plane of W_hh[256 q + lr_in_unit(c16, lane >> 5, e)][32 mo + (lane & 31)],
// gs = 4 c16 + pos, q = lr_gate(pos).
__global__ __launch_bounds__(kBlock) void lstm_rows_pack_kernel(const float *__restrict__ w_hh, uint32_t *__restrict__ packed) {
  const int idx = blockIdx.x * kBlock + threadIdx.x;  // (gs, mo, lane)
  if (idx >= kLrGateSteps * 8 * 64) return;
  const int lane = idx & 63, mo = (idx >> 6) & 7, gs = idx >> 9;
  const int q = lr_gate(gs & 3), c = gs >> 2;
  const int out = 32 * mo + (lane & 31);
  const float *src = w_hh + (int64_t)(kHidden * q) * kHidden + out;
  u32x4 planes[3];
#pragma unroll
  for (int e = 0; e < 8; e += 2) {
    uint32_t hi, mid, lo;
    split_pair(src[lr_in_unit(c, lane >> 5, e) * kHidden], src[lr_in_unit(c, lane >> 5, e + 1) * kHidden], hi, mid, lo);
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
  unsigned long long *stamps;  // tuning builds (RL8_LR_STAMP): cycles per wait site, summed over waves; else null
  uint32_t *dg_bound;          // max |dG| over everything written, as the bits of a non-negative float (atomic max); or null
};

typedef float f32x16v __attribute__((ext_vector_type(16)));

// what a lane loads for one unit chunk: its sixteen units (two runs of eight, 64 bytes apart) of its sequence
struct LrLoadDesc {
  __amdgpu_buffer_rsrc_t gates, cs, cprev, dhs, dcin;
  int cp_pitch;  // bytes between sequences in cprev (cs of t - 1: the sequence pitch; c0: 1 KiB)
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

  // per-lane byte offsets: sequence n of the wave's 32; piece j (registers 4 j .. 4 j + 3) of a lane's sixteen units of
  // a chunk is units 8 j + 4 hh ..+3: lanes n and n + 32 move 32 adjacent bytes per instruction, a 128-byte line in four
  const int v_gates = n * l * (4 * kHidden * 4) + hh * 16;
  const int v_seq = n * l * (kHidden * 4) + hh * 16;
  const int v_state = n * (kHidden * 4) + hh * 16;
  auto piece = [](int j) { return j * 32; };

  // gate-step k of chunk c -> ring slot: the packed order is [16-unit chunk 2 c + half][gate position]
  auto request = [&](int c, int k, int slot) {
    const int src = (4 * (2 * c + (k & 1)) + (k >> 1)) & (kLrGateSteps - 1);
#pragma unroll
    for (int u = 0; u < kLrDma; ++u) {
      const int block = wave * kLrDma + u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, smem + slot * kLrSlotBytes + block * 1024, 16, lane * 16,
                                               src * kLrSlotBytes + block * 1024, 0, 0);
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
    d.cp_pitch = t > 0 ? l * (kHidden * 4) : kHidden * 4;
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
  auto load4 = [&](u32x4 (&dst)[4], const __amdgpu_buffer_rsrc_t &r, int voff, int soff) {
#pragma unroll
    for (int j = 0; j < 4; ++j) dst[j] = __builtin_amdgcn_raw_buffer_load_b128(r, voff + piece(j), soff, 0);
  };
  auto store4 = [&](const float (&v)[16], const __amdgpu_buffer_rsrc_t &r, int voff, int soff) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(v[4 * j]), __float_as_uint(v[4 * j + 1]),
                                                   __float_as_uint(v[4 * j + 2]), __float_as_uint(v[4 * j + 3])},
                                             r, voff + piece(j), soff, 0);
  };
  auto at = [](const u32x4 (&v)[4], int e) { return __uint_as_float(v[e >> 2][e & 3]); };

#ifdef RL8_LR_STAMP
  unsigned long long tw[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};  // 0..7 the barriers' vmcnt, 8 B unpark, 9 C unpark, 10 barrier itself, 11 all,
  // 12 fragment waits, 13 issue of row operations and requests, 14 gate arithmetic (with its waits and stores), 15 plane split
  const unsigned long long t_begin = __builtin_readcyclecounter();
#define RL8_LR_T0 const unsigned long long t0_ = __builtin_readcyclecounter()
#define RL8_LR_T1(i) tw[i] += __builtin_readcyclecounter() - t0_
#else
#define RL8_LR_T0
#define RL8_LR_T1(i)
#endif
  // the three load phases of a chunk (see the header) and the arithmetic behind each
  u32x4 la_o[4], la_ct[4], la_dh[4], la_dc[4];
  auto issue_a = [&](const LrLoadDesc &d, int c) {
    load4(la_o, d.gates, v_gates, c * 128 + 3 * (kHidden * 4));
    load4(la_ct, d.cs, v_seq, c * 128);
    load4(la_dh, d.dhs, v_seq, c * 128);
    load4(la_dc, d.dcin, v_state, c * 128);
  };
  // park PH (0: phase B, 1: phase C), array AR of this wave: 32 rows x 128 bytes, fetched COALESCED -- instruction k
  // brings rows 8 k .. 8 k + 7 whole, eight lanes per row (8 tag look-ups instead of the 64 of a lane-per-row load) --
  // and laid out row-major in LDS, the eight 16-byte pieces of row r permuted by XOR with (r >> 1) & 7 so that the
  // lane = row reads below fall on sixteen different bank groups per quarter wave.  Lane i of instruction k: row
  // 8 k + (i >> 3), LDS slot i & 7 <- piece (i & 7) ^ ((4 k + (i >> 4)) & 7) of the row.
  const int park_piece = (lane & 7) ^ (lane >> 4);  // k even; k odd: ^ 4
  auto park4 = [&](int ph, int ar, const __amdgpu_buffer_rsrc_t &r, int pitch_bytes, int soff) {
    const int v_row = (lane >> 3) * pitch_bytes;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(
          r, smem + kLrRing * kLrSlotBytes + ph * kLrStageBytes + ((wave * 2 + ar) * 4 + k) * 1024, 16,
          v_row + ((k & 1) ? (park_piece ^ 4) : park_piece) * 16, soff + 8 * k * pitch_bytes, 0, 0);
  };
  // lane (n, hh) reads piece 2 j + hh of row n: slot (2 j + hh) ^ ((n >> 1) & 7)
  unsigned park_read[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
    park_read[j] = lds0 + kLrRing * kLrSlotBytes + wave * (8 * 1024) + n * 128 + (((2 * j + hh) ^ ((n >> 1) & 7)) * 16);
  const int pitch_gates = l * (4 * kHidden * 4), pitch_seq = l * (kHidden * 4);
  auto issue_b = [&](const LrLoadDesc &d, int c) {
    park4(0, 0, d.gates, pitch_gates, c * 128);
    park4(0, 1, d.gates, pitch_gates, c * 128 + 2 * (kHidden * 4));
  };
  auto issue_c = [&](const LrLoadDesc &d, int c) {
    park4(1, 0, d.gates, pitch_gates, c * 128 + 1 * (kHidden * 4));
    park4(1, 1, d.cprev, d.cp_pitch, c * 128);
  };
  // the parked phase PH back into registers; N = operations the wave has issued behind those loads
  auto unpark = [&](auto ph_tag, auto n_tag, u32x4 (&x)[4], u32x4 (&y)[4]) {
    constexpr int PH = decltype(ph_tag)::value;
    constexpr int N = RL8_LR_SAFE_WAITS ? 0 : decltype(n_tag)::value;
    {
      RL8_LR_T0;
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
      RL8_LR_T1(8 + PH);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      x[j] = lds_read_b128<PH * kLrStageBytes>(park_read[j]);
      y[j] = lds_read_b128<PH * kLrStageBytes + 4 * 1024>(park_read[j]);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(y[0]), "+v"(y[1]), "+v"(y[2]), "+v"(y[3]));
  };

  // dh carried into the step being computed: dhe[e][c] = dL/dh of unit 32 c + 8 (e >> 2) + 4 hh + (e & 3) from the step
  // after it (register e of accumulator c)
  typedef float f32x8v __attribute__((ext_vector_type(8)));
  f32x8v dhe[16];
  f32x16 acc[8];
  float dg[4][16];  // [gate position][e]: dG of the chunk whose matrix work is under way / comes next
  float dcv[16];    // dL/dc of that chunk (between its phases A and C)
  // max |dG| of everything this lane has stored (the weight-gradient kernels scale dG by a bound on it), parked in an
  // accumulator register between the three places per chunk that fold into it: the vector registers are all taken
  float dg_max;
  asm volatile("v_accvgpr_write_b32 %0, 0" : "=a"(dg_max));
  auto fold_max = [&](float m) {
    float t;
    asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(t) : "a"(dg_max));
    t = __builtin_fmaxf(t, m);
    asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(dg_max) : "v"(t));
  };

  auto math_a = [&](const LrStoreDesc &sd, int c) {  // -> dg[0] (o), dcv
    float lmax = 0.0f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const float go = at(la_o, e), ct = at(la_ct, e);
      const float dh = at(la_dh, e) + dhe[e][c];
      const float tc = lr_tanh(ct);
      dg[0][e] = dh * tc * (go * (1.0f - go));
      dcv[e] = __builtin_fmaf(dh * go, 1.0f - tc * tc, at(la_dc, e));
      lmax = __builtin_fmaxf(lmax, __builtin_fabsf(dg[0][e]));
    }
    fold_max(lmax);
    store4(dg[0], sd.dgates, v_gates, c * 128 + 3 * (kHidden * 4));
  };
  auto math_b = [&](const LrStoreDesc &sd, int c) {  // -> dg[1] (i), dg[2] (g)
    u32x4 lb_i[4], lb_g[4];
    unpark(std::integral_constant<int, 0>{}, std::integral_constant<int, kLrBehindB>{}, lb_i, lb_g);
    float lmax = 0.0f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const float gi = at(lb_i, e), gg = at(lb_g, e);
      dg[1][e] = dcv[e] * gg * (gi * (1.0f - gi));
      dg[2][e] = dcv[e] * gi * (1.0f - gg * gg);
      lmax = __builtin_fmaxf(lmax, __builtin_fmaxf(__builtin_fabsf(dg[1][e]), __builtin_fabsf(dg[2][e])));
    }
    fold_max(lmax);
    store4(dg[1], sd.dgates, v_gates, c * 128);
    store4(dg[2], sd.dgates, v_gates, c * 128 + 2 * (kHidden * 4));
  };
  auto math_c = [&](const LrStoreDesc &sd, int c) {  // -> dg[3] (f), dc out
    u32x4 lc_f[4], lc_cp[4];
    unpark(std::integral_constant<int, 1>{}, std::integral_constant<int, kLrBehindC>{}, lc_f, lc_cp);
    float dc_out[16], lmax = 0.0f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const float gf = at(lc_f, e), cp = at(lc_cp, e);
      dg[3][e] = dcv[e] * cp * (gf * (1.0f - gf));
      dc_out[e] = dcv[e] * gf;
      lmax = __builtin_fmaxf(lmax, __builtin_fabsf(dg[3][e]));
    }
    fold_max(lmax);
    store4(dg[3], sd.dgates, v_gates, c * 128 + 1 * (kHidden * 4));
    store4(dc_out, sd.dcout, v_state, c * 128);
  };

  // one gate-step: 16 k of the product, W_hh^T planes from ring slot K & 3, B planes from dg[K >> 1][8 (K & 1) ..+7].
  // Out tiles in pairs: two accumulators alternate, so no MFMA waits for the one before it.
  auto matrix_step = [&](auto k_tag) {
    constexpr int K = decltype(k_tag)::value;
    constexpr int POS = K >> 1, E0 = 8 * (K & 1);
    u32x4 bh, bm, bl;
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
      uint32_t hi, mid, lo;
      split_pair(dg[POS][E0 + e], dg[POS][E0 + e + 1], hi, mid, lo);
      bh[e >> 1] = hi;
      bm[e >> 1] = mid;
      bl[e >> 1] = lo;
    }
    const unsigned ar = a_read + (K & 3) * kLrSlotBytes;
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
      {
        RL8_LR_T0;
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(ah[s][0]), "+v"(ah[s][1]), "+v"(am[s][0]), "+v"(am[s][1]), "+v"(al[s][0]), "+v"(al[s][1]));
        RL8_LR_T1(12);
      }
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

  // The barrier that opens gate-step K of a chunk: this wave's share of its chunk of W_hh^T has landed (vector-memory
  // operations complete in issue order: all but the N youngest are done) and every wave is through with the slot the
  // next request overwrites.  N = lr_behind(K): what the wave has issued behind that request, made three gate-steps
  // earlier, with the order of a gate-step's operations [row loads | request | matrix work | stores]:
  //   K        0   1   2   3   4   5   6   7
  //   loads    8   .  16   .   8   .   .   .      (C | A of the next chunk | B of the next chunk)
  //   stores   .   8   .   .   .   8   .   4      (dG_i, dG_g | dG_f, dc | dG_o of the next chunk)
  //   N       24  24  32  36  36  20  28  20
  auto open_step = [&](auto k_tag) {
    constexpr int K = decltype(k_tag)::value;
    constexpr int N = RL8_LR_SAFE_WAITS ? 0 : lr_behind(K);
#ifdef RL8_LR_STAMP
    {
      RL8_LR_T0;
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory");
      RL8_LR_T1(K);
    }
    {
      RL8_LR_T0;
      asm volatile("s_barrier" ::: "memory");
      RL8_LR_T1(10);
    }
#else
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
#endif
  };
  auto K_ = [](auto i) { return std::integral_constant<int, decltype(i)::value>{}; };
  using K0 = std::integral_constant<int, 0>;
  using K1 = std::integral_constant<int, 1>;
  using K2 = std::integral_constant<int, 2>;
  using K3 = std::integral_constant<int, 3>;
  using K4 = std::integral_constant<int, 4>;
  using K5 = std::integral_constant<int, 5>;
  using K6 = std::integral_constant<int, 6>;
  using K7 = std::integral_constant<int, 7>;
  (void)K_;

  int64_t tile = blockIdx.x;
  if (tile >= tiles) return;  // (the host launches no more workgroups than tiles)
  int t = l - 1;
  LrStoreDesc sd = store_desc(tile, t);
#pragma unroll
  for (int e = 0; e < 16; ++e) dhe[e] = f32x8v{0, 0, 0, 0, 0, 0, 0, 0};
  LrLoadDesc nd = load_desc(tile, t);  // where the loads of the current step come from
  issue_a(nd, 0);
  issue_b(nd, 0);
  request(0, 0, 0);
  request(0, 1, 1);
  request(0, 2, 2);
  math_a(sd, 0);
  // the counted waits of the first chunk assume a full chunk of operations behind each request: drain instead
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  while (true) {
    // position behind this step: the same sequences one step earlier, or the next tile's last step, or nothing
    const bool last_step = t == 0;
    const int64_t ntile = last_step ? tile + gridDim.x : tile;
    const int nt = last_step ? l - 1 : t - 1;
    const LrLoadDesc after = load_desc(ntile, nt);
#pragma unroll 1
    for (int c = 0; c < kLrChunks; ++c) {
      const bool wrap = c == kLrChunks - 1;
      const int cn = wrap ? 0 : c + 1;
      open_step(K0{});
      { RL8_LR_T0; issue_c(nd, c); RL8_LR_T1(13); }
      asm volatile("" ::: "memory");
      { RL8_LR_T0; request(c, 3, 3); RL8_LR_T1(13); }
      if (c == 0) {
#pragma unroll
        for (int mo = 0; mo < 8; ++mo) acc[mo] = f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
      }
      matrix_step(K0{});
      open_step(K1{});
      { RL8_LR_T0; request(c, 4, 0); RL8_LR_T1(13); }
      matrix_step(K1{});
      { RL8_LR_T0; math_b(sd, c); RL8_LR_T1(14); }
      open_step(K2{});
      { RL8_LR_T0; issue_a(wrap ? after : nd, cn); RL8_LR_T1(13); }
      asm volatile("" ::: "memory");
      { RL8_LR_T0; request(c, 5, 1); RL8_LR_T1(13); }
      matrix_step(K2{});
      open_step(K3{});
      { RL8_LR_T0; request(c, 6, 2); RL8_LR_T1(13); }
      matrix_step(K3{});
      open_step(K4{});
      { RL8_LR_T0; issue_b(wrap ? after : nd, cn); RL8_LR_T1(13); }
      asm volatile("" ::: "memory");
      { RL8_LR_T0; request(c, 7, 3); RL8_LR_T1(13); }
      matrix_step(K4{});
      open_step(K5{});
      { RL8_LR_T0; request(cn, 0, 0); RL8_LR_T1(13); }
      matrix_step(K5{});
      { RL8_LR_T0; math_c(sd, c); RL8_LR_T1(14); }
      open_step(K6{});
      { RL8_LR_T0; request(cn, 1, 1); RL8_LR_T1(13); }
      matrix_step(K6{});
      open_step(K7{});
      { RL8_LR_T0; request(cn, 2, 2); RL8_LR_T1(13); }
      matrix_step(K7{});
      if (wrap) {
        // the step's dh is complete: it becomes the carry of the step the next arithmetic belongs to (zero for a new tile)
#pragma unroll
        for (int cc = 0; cc < kLrChunks; ++cc)
#pragma unroll
          for (int e = 0; e < 16; ++e) dhe[e][cc] = last_step ? 0.0f : acc[cc][e];
        sd = store_desc(ntile, nt);
      }
      { RL8_LR_T0; math_a(sd, cn); RL8_LR_T1(14); }
    }
    if (last_step && ntile >= tiles) break;
    tile = ntile;
    t = nt;
    nd = after;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no request may still be writing LDS when the workgroup ends
  if (a.dg_bound) {
    // (rows past the end contributed zeros: their loads returned 0.0)
    float t;
    asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(t) : "a"(dg_max));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) t = __builtin_fmaxf(t, __shfl_down(t, off, 64));
    if (lane == 0) atomicMax(a.dg_bound, __float_as_uint(t));
  }
#ifdef RL8_LR_STAMP
  tw[11] = __builtin_readcyclecounter() - t_begin;
  if (a.stamps && lane == 0)
    for (int i = 0; i < 16; ++i) atomicAdd(a.stamps + i, tw[i]);
#endif
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
                                       const void *packed, float *dgates, float *dc_scratch, uint32_t *dg_bound_out,
                                       void *stream) {
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
  unsigned long long *stamps = nullptr;
#ifdef RL8_LR_STAMP
  if (const char *v = getenv("RL8_LR_STAMP_PTR")) stamps = reinterpret_cast<unsigned long long *>(strtoull(v, nullptr, 0));
#endif
  if (dg_bound_out && hipMemsetAsync(dg_bound_out, 0, 4, (hipStream_t)stream) != hipSuccess) return launch_status();
  const LrArgs args = {c0, gates, cs, dhs, dgates, dc_scratch, b, l, stamps, dg_bound_out};
  lstm_rows_backward_kernel<<<grid, kBlock, kLrLdsBytes, (hipStream_t)stream>>>(args, packed);
  return launch_status();
}
