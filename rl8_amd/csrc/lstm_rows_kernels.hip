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
// (11.65 ms per 2^21 row-steps, 60 % of the fp32 matrix peak, W_hh^T re-read from L2 per 32 rows).  Round 6: the HEADS
// form -- where such a bound DOES exist, per sequence and step -- on two fp16 planes, three products (see lr_slot_bytes).
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
// Row accesses are whole 128-byte lines AND coalesced: lane = sequence would make every lane of a load or store address
// its own cache line (64 tag look-ups per instruction: with one wave per SIMD the wave stood at the ISSUE of its row
// operations for 28 % of its time).  So every row array goes through a per-wave park in LDS, 32 rows x 128 bytes per array
// and chunk: loads are direct-to-LDS, instruction k bringing rows 8 k .. 8 k + 7 whole (eight lanes per row), read back
// lane = sequence; stores are written lane = sequence into a park and read back eight lanes per row.  The eight 16-byte
// pieces of a row are XOR-swizzled by (row >> 1) & 7: both views are free of bank conflicts.  Two parks of two arrays
// per wave (the LDS is full: 96 KiB ring + 64 KiB parks), three load phases per chunk, each requested where its park
// falls free: B = {i, g} in gate-step 0 -> dG_i, dG_g behind step 1; C = {f, c_{t-1}} in 2 -> dG_f, dc behind 5;
// A of the NEXT chunk = {o, c_t | dh_t, dc} (both parks) in 6 -> dG_o, dc behind 7.
// (Rounds of this kernel: chunks of sixteen units, half a line per access: 10.4 ms per 2^21 row-steps; whole lines, lane
// per row, loads in registers: 9.0; phases B, C parked and coalesced: 8.6; everything through the parks: this one.)
// Carries: dh in registers (copied out of the accumulators once per step, read per chunk by a dynamic register index
// c); dc through a [b][256] scratch in HBM (2 KiB of traffic per row-step; in registers it would take the 128 the row
// loads in flight need: 512 per lane = 128 accumulators + 128 dh + loads, dG, planes, fragments).
// HBM per row-step: gates 4 KiB, c_t, c_{t-1}, dh_t 1 KiB each, dc in/out 2 KiB, dG 4 KiB out = 13 KiB (the fp32 kernel
// moves 11); matrix pipe 3072 cycles per row-step.  Measured (profiles/r03_lstm_traffic.txt): fabric traffic 1.00x / 1.03x
// of that, 28.4 GB per 2^21 row-steps at 3.7 TB/s -- the rate HBM gives 128-byte pieces (a chunk's 32 units of a row).
// Second form (lstm_rows_backward_heads_kernel): dL/dh_t of the models' output heads is not an array but
// dOut[row-step][0..3] x W_heads[4][256], formed in phase A from two direct-to-LDS loads per chunk (12 KiB per row-step).
#include "split_tile.hip.h"

namespace rl8 {

constexpr int kLrRows = 128;                 // sequences per workgroup (32 per wave)
constexpr int kLrChunks = 8;                 // unit chunks (out tiles) per step
constexpr int kLrGateSteps = 8 * kLrChunks;  // k-chunks of 16 per step
// Two plane schemes (round 6).  bf16: three exact planes of both operands, six products -- any dG.  F16 (the HEADS form
// only): two fp16 planes, THREE products.  dG has fp32's range, but lane = sequence in the B operand AND in the result, so
// a power of two per SEQUENCE and step commutes with the sum over gate rows: dG~ = dG 2^(kLrTop - e), e from a bound on the
// step's |dG| of that sequence that is known before its first chunk -- |dh| <= sum_q |dOut_q| max|W_q| + max|carried dh|,
// |dc| <= |dh| + max|carried dc|, |do|, |di| <= a quarter of those, |dg| <= |dc|, |df| <= |dc| |c_{t-1}| / 4 -- with the
// bound placed at 2^kLrTop = 2^6 instead of fp16's 2^14: |c_{t-1}| up to 4 094 before a plane overflows (to inf: loudly).
// Values down to 2^-9 of the bound keep hi + lo = 22 bits, below that the absolute resolution is 2^-31 of the bound.
constexpr int lr_slot_bytes(bool f16) { return (f16 ? 2 : 3) * 8 * 1024; }  // one gate-step of W_hh^T: [plane][out tile] x 1 KiB
constexpr int lr_dma(bool f16) { return f16 ? 4 : 6; }                       // 1-KiB direct-to-LDS loads per wave and gate-step
constexpr int kLrSlotBytes = lr_slot_bytes(false);
constexpr int kLrRing = 4;
constexpr int kLrPackedBytes = kLrGateSteps * kLrSlotBytes;          // 1.5 MiB: the bf16 planes
constexpr int kLrPacked16Bytes = kLrGateSteps * lr_slot_bytes(true);  // 1 MiB: the fp16 planes behind them, then {2^k, 2^-k, 0, 0}
constexpr int kLrPackTotalBytes = kLrPackedBytes + kLrPacked16Bytes + 16;
constexpr int kLrStageBytes = 4 * 8 * 1024;  // one park of two arrays: [wave][array][instruction] x 1 KiB; two parks
constexpr int lr_lds_bytes(bool f16) { return kLrRing * lr_slot_bytes(f16) + (f16 ? 3 : 2) * kLrStageBytes; }  // (F16: a third park)
constexpr int kLrLdsBytes = lr_lds_bytes(false);  // 160 KiB: the whole CU, either way
constexpr int kLrDma = lr_dma(false);
constexpr int kLrTop = 6;
// 16-byte row operations a lane issues in gate-step k of a chunk (k = 2 pos + half): loads in front of the step's
// request, stores behind its matrix work
// HEADS (the kernel's second form): dL/dh_t of the output heads is not read as a [b][l][256] array but formed from the
// heads' own gradient, four floats per row-step, and their weights: phase A then brings the chunk's rows of the heads'
// weights and the rows' four floats (one instruction each) instead of the four of dh_t.
constexpr int lr_loads_at(int k, bool heads, bool f16 = false) {  // B of this chunk | C of this chunk | A of the next
  // F16 (round 6, three parks: the ring's 16-KiB slots leave room for one more): every phase but the last third of A is
  // requested most of a chunk ahead of its arithmetic -- C of this chunk in step 0 (its park fell free with the arithmetic
  // that closed the chunk before), {o, c_t} of the NEXT chunk in step 1, {i, g} of the NEXT chunk in step 2 (the third
  // park: free since this chunk's dG_i / dG_g left it behind step 1), the heads' rows and the carried dc of the next chunk
  // in step 6 as before (their park is C's until step 5)
  if (f16) return k == 0 || k == 1 || k == 2 ? 8 : k == 6 ? 6 : 0;
  return k == 0 || k == 2 ? 8 : k == 6 ? (heads ? 14 : 16) : 0;
}
constexpr int kLrStoresAt[8] = {0, 8, 0, 0, 0, 8, 0, 4};   // dG_i, dG_g | dG_f, dc | dG_o of the next chunk
// operations a wave has issued behind its request for gate-step k's chunk of W_hh^T (made in step k - 3)
constexpr int lr_behind(int k, bool heads, bool f16 = false) {
  const int dma = lr_dma(f16);
  return kLrStoresAt[(k + 5) & 7] + lr_loads_at((k + 6) & 7, heads, f16) + dma + kLrStoresAt[(k + 6) & 7] +
         lr_loads_at((k + 7) & 7, heads, f16) + dma + kLrStoresAt[(k + 7) & 7];
}
static_assert(lr_behind(0, false) == 40 && lr_behind(3, false) == 28 && lr_behind(5, false) == 12 && lr_behind(7, false) == 36,
              "see the table in open_step");
static_assert(lr_behind(0, true) == 38 && lr_behind(1, true) == 24 && lr_behind(6, true) == 20 && lr_behind(7, true) == 34,
              "see the table in open_step");
// ... and behind the parked loads of a phase, from the step that makes them to the end of the matrix work of the step that
// reads them: B 0 -> 1, C 2 -> 5, A 6 -> 7 (vmcnt has six bits: all but the 63 youngest covers anything further back)
constexpr int lr_behind_loads(int from, int to, bool heads, bool f16 = false) {  // from the loads of step `from` to the end of step `to`'s matrix work
  const int dma = lr_dma(f16);
  int n = dma;
  for (int k = (from + 1) & 7;; k = (k + 1) & 7) {
    n += kLrStoresAt[(k + 7) & 7] + lr_loads_at(k, heads, f16) + dma;
    if (k == to) break;
  }
  return n < 63 ? n : 63;
}
// (none of the three spans contains gate-step 6: the same with and without HEADS)
constexpr int kLrBehindB = lr_behind_loads(0, 1, false);
constexpr int kLrBehindC = lr_behind_loads(2, 5, false);
constexpr int kLrBehindA = lr_behind_loads(6, 7, false);
static_assert(kLrBehindB == 12 && kLrBehindC == 24 && kLrBehindA == 12, "see the table in open_step");
static_assert(kLrBehindB == lr_behind_loads(0, 1, true) && kLrBehindC == lr_behind_loads(2, 5, true) &&
                  kLrBehindA == lr_behind_loads(6, 7, true),
              "the spans do not contain gate-step 6");
// F16 (four direct-to-LDS pieces per gate-step instead of six, the three-park schedule): K 0..7 -> 26 20 36 32 24 8 16 22;
// the phases' spans: {i, g} from step 2 to step 1 of the next chunk (66 operations: all but the 63 youngest covers it),
// {f, c_{t-1}} from step 0 to step 5 (48), the last part of A from step 6 to step 7 (8)
static_assert(lr_behind(0, true, true) == 26 && lr_behind(1, true, true) == 20 && lr_behind(2, true, true) == 36 &&
                  lr_behind(3, true, true) == 32 && lr_behind(4, true, true) == 24 && lr_behind(5, true, true) == 8 &&
                  lr_behind(6, true, true) == 16 && lr_behind(7, true, true) == 22,
              "see the table in open_step");
static_assert(lr_behind_loads(2, 1, true, true) == 63 && lr_behind_loads(0, 5, true, true) == 48 && lr_behind_loads(6, 7, true, true) == 8,
              "see the table in open_step");
constexpr int kLrHeads = 4;  // outputs of the heads the HEADS form takes (their gradient padded to four floats per row-step)
#ifndef RL8_LR_DIAG
#define RL8_LR_DIAG 0  // tuning builds (tools/diag_mlp.sh lr<bits>): 1 no stores reach memory, 2 no row loads do, 4 one plane
#endif                 // product of six, 8 no W_hh^T traffic (wrong results, same instruction stream)
constexpr int kLrDiag = RL8_LR_DIAG;
#ifndef RL8_LR_STORE_AUX
#define RL8_LR_STORE_AUX 0                   // cache policy of the kernel's stores (tuning builds: 18 = sc1 | nt, streaming)
#endif
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
  const float *dhs;    // [b][l][256]; HEADS: [b][l][4], the gradient of the heads' outputs (zero-padded)
  const float *heads_w;  // HEADS: [4][256], the heads' weights (rows past their number zero); else unused
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
  __amdgpu_buffer_rsrc_t gates, cs, cprev, dhs, dcin;  // (HEADS: dhs = the rows' four floats)
  int cp_pitch;  // bytes between sequences in cprev (cs of t - 1: the sequence pitch; c0: 1 KiB)
};
struct LrStoreDesc {
  __amdgpu_buffer_rsrc_t dgates, dcout;
};

template <bool HEADS, bool F16 = false>
__device__ __forceinline__ void lstm_rows_backward_body(const LrArgs &a, const void *__restrict__ w_planes) {
  static_assert(HEADS || !F16, "the fp16 planes need the step's bound on |dh| before its first chunk: the HEADS form only");
  constexpr int kSlot = lr_slot_bytes(F16), kDma = lr_dma(F16);  // (shadow the bf16 shape's names below)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const unsigned lds0 = lds_offset(smem);
  const int tid = threadIdx.x, lane = tid & 63, n = lane & 31, hh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l = a.l;
  const __amdgpu_buffer_rsrc_t wrsrc = buffer_rsrc(static_cast<const unsigned char *>(w_planes) + (F16 ? kLrPackedBytes : 0),
                                                   (kLrDiag & 8) ? 0 : (F16 ? kLrPacked16Bytes : kLrPackedBytes));
  const unsigned a_read = lds0 + lane * 16;
  // F16: 1 / (W_hh's power of two) from behind the planes; max |W_heads[q]| per head (wave-uniform, scalar registers)
  [[maybe_unused]] float inv_sw = 1.0f, wmax[kLrHeads] = {0.0f, 0.0f, 0.0f, 0.0f};
  if constexpr (F16) {
    inv_sw = reinterpret_cast<const float *>(static_cast<const unsigned char *>(w_planes) + kLrPackedBytes + kLrPacked16Bytes)[1];
#pragma unroll
    for (int q = 0; q < kLrHeads; ++q) {
      const f32x4 v = reinterpret_cast<const f32x4 *>(a.heads_w + q * kHidden)[lane];
      float m = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(v[0]), __builtin_fabsf(v[1])),
                                __builtin_fmaxf(__builtin_fabsf(v[2]), __builtin_fabsf(v[3])));
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) m = __builtin_fmaxf(m, __shfl_xor(m, off, 64));
      wmax[q] = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(m)));
    }
    // (all of it landed and reduced before the first counted wait: the prologue below drains with vmcnt(0))
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  }
  // F16: this sequence's power of two for the step being computed, its inverse times W_hh's; what bounds the next step's
  [[maybe_unused]] float s_n = 1.0f, inv_n = 1.0f, cmax = 0.0f, emax = 0.0f, emax_prev = 0.0f;

  // gate-step k of chunk c -> ring slot: the packed order is [16-unit chunk 2 c + half][gate position]
  auto request = [&](int c, int k, int slot, int u0 = 0, int u1 = lr_dma(F16)) {  // pieces u0 .. u1 - 1 of the wave's six (F16: four)
    const int src = (4 * (2 * c + (k & 1)) + (k >> 1)) & (kLrGateSteps - 1);
#pragma unroll
    for (int u = u0; u < u1; ++u) {
      const int block = wave * kDma + u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, smem + slot * kSlot + block * 1024, 16, lane * 16,
                                               src * kSlot + block * 1024, 0, 0);
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
    d.dhs = seq_rsrc(a.dhs, tile, t, HEADS ? kLrHeads : kHidden, rows);
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
  // Parks: PARK 0 / 1, array AR (0 / 1) of this wave = four 1-KiB blocks; instruction k moves rows 8 k .. 8 k + 7.
  // Lane i of instruction k: row 8 k + (i >> 3), LDS slot i & 7 <-> piece (i & 7) ^ ((4 k + (i >> 4)) & 7) of the row.
  const int park_piece = (lane & 7) ^ (lane >> 4);  // k even; k odd: ^ 4
  auto park_lds = [&](int park, int ar, int k) { return kLrRing * kSlot + park * kLrStageBytes + ((wave * 2 + ar) * 4 + k) * 1024; };
  auto park4 = [&](int park, int ar, const __amdgpu_buffer_rsrc_t &r, int pitch_bytes, int soff, int k0 = 0, int k1 = 4) {
    const int v_row = (lane >> 3) * pitch_bytes;
#pragma unroll
    for (int k = k0; k < k1; ++k)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(r, smem + park_lds(park, ar, k), 16,
                                               v_row + ((k & 1) ? (park_piece ^ 4) : park_piece) * 16, soff + 8 * k * pitch_bytes, 0, 0);
  };
  // lane (n, hh) holds piece 2 j + hh of row n: slot (2 j + hh) ^ ((n >> 1) & 7)
  unsigned park_at[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
    park_at[j] = lds0 + kLrRing * kSlot + wave * (8 * 1024) + n * 128 + (((2 * j + hh) ^ ((n >> 1) & 7)) * 16);
  const unsigned park_line = lds0 + kLrRing * kSlot + wave * (8 * 1024) + lane * 16;  // the instruction view
  // (a DS instruction's offset field has sixteen bits: the third park, 64 KiB up, gets address registers of its own)
  [[maybe_unused]] unsigned park2_at[4], park2_line = park_line + 2 * kLrStageBytes;
#pragma unroll
  for (int j = 0; j < 4; ++j) park2_at[j] = park_at[j] + 2 * kLrStageBytes;
  const int pitch_gates = l * (4 * kHidden * 4), pitch_seq = l * (kHidden * 4), pitch_state = kHidden * 4;
  // (each in parts, so that a gate-step can spread them between its MFMA groups: a direct-to-LDS load costs the wave
  // ~75 cycles at issue, and issued in one run at the head of a step those were 19 % of its time with the pipe idle)
  constexpr int kParkB = F16 ? 2 : 0;  // (F16: {i, g} and their dG live in the third park)
  auto issue_b = [&](const LrLoadDesc &d, int c, int part = -1) {
    if (part != 1) park4(kParkB, 0, d.gates, pitch_gates, c * 128);
    if (part != 0) park4(kParkB, 1, d.gates, pitch_gates, c * 128 + 2 * (kHidden * 4));
  };
  auto issue_c = [&](const LrLoadDesc &d, int c, int part = -1) {
    if (part != 1) park4(1, 0, d.gates, pitch_gates, c * 128 + 1 * (kHidden * 4));
    if (part != 0) park4(1, 1, d.cprev, d.cp_pitch, c * 128);
  };
  const __amdgpu_buffer_rsrc_t hwrsrc = buffer_rsrc(HEADS ? a.heads_w : nullptr, HEADS ? kLrHeads * kHidden * 4 : 0);
  auto issue_a = [&](const LrLoadDesc &d, int c, int part = -1) {  // parts 0, 1, 2: six, six (HEADS: four) and four of the sixteen
    if (part < 0 || part == 0) {
      park4(0, 0, d.gates, pitch_gates, c * 128 + 3 * (kHidden * 4));
      park4(0, 1, d.cs, pitch_seq, c * 128, 0, 2);
    }
    if (part < 0 || part == 1) {
      park4(0, 1, d.cs, pitch_seq, c * 128, 2, 4);
      if constexpr (HEADS) {
        // park 1, array 0: block 0 = [head q][this chunk's 32 units] (lanes 0..31: q = lane >> 3, 16 bytes each; the rest
        // of the instruction is out of range: zeros), block 1 = [sequence] x the four floats of its row-step
        const int far = 0x7fffff00;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(hwrsrc, smem + park_lds(1, 0, 0), 16,
                                                 lane < 32 ? (lane >> 3) * (kHidden * 4) + (lane & 7) * 16 : far, c * 128, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(d.dhs, smem + park_lds(1, 0, 1), 16, lane < 32 ? lane * (l * kLrHeads * 4) : far,
                                                 0, 0, 0);
      } else {
        park4(1, 0, d.dhs, pitch_seq, c * 128);
      }
    }
    if (part < 0 || part == 2) park4(1, 1, d.dcin, pitch_state, c * 128);
  };
  // array AR of park PARK into this lane's registers (its sixteen units); the wait for the loads is the caller's
  auto unpark = [&](auto park_tag, auto ar_tag, u32x4 (&x)[4]) {
    constexpr bool THIRD = decltype(park_tag)::value == 2;
    constexpr int OFF = (THIRD ? 0 : decltype(park_tag)::value * kLrStageBytes) + decltype(ar_tag)::value * 4096;
#pragma unroll
    for (int j = 0; j < 4; ++j) x[j] = lds_read_b128<OFF>(THIRD ? park2_at[j] : park_at[j]);
  };
  auto loads_landed = [&](auto n_tag, int stamp) {  // N = operations the wave has issued behind the phase's loads
    constexpr int N = RL8_LR_SAFE_WAITS ? 0 : decltype(n_tag)::value;
    RL8_LR_T0;
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
    RL8_LR_T1(stamp);
    (void)stamp;
  };
  // sixteen values per lane -> array AR of park PARK (lane = sequence view), then out of it eight lanes per row
  auto repark = [&](auto park_tag, auto ar_tag, const float (&v)[16]) {
    constexpr bool THIRD = decltype(park_tag)::value == 2;
    constexpr int OFF = (THIRD ? 0 : decltype(park_tag)::value * kLrStageBytes) + decltype(ar_tag)::value * 4096;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      lds_write_b128<OFF>(THIRD ? park2_at[j] : park_at[j], u32x4{__float_as_uint(v[4 * j]), __float_as_uint(v[4 * j + 1]), __float_as_uint(v[4 * j + 2]),
                                            __float_as_uint(v[4 * j + 3])});
  };
  auto store_park = [&](auto park_tag, auto ar_tag, const __amdgpu_buffer_rsrc_t &r, int pitch_bytes, int soff) {
    constexpr bool THIRD = decltype(park_tag)::value == 2;
    constexpr int OFF = (THIRD ? 0 : decltype(park_tag)::value * kLrStageBytes) + decltype(ar_tag)::value * 4096;
    const unsigned line = THIRD ? park2_line : park_line;
    u32x4 v[4];
    v[0] = lds_read_b128<OFF>(line), v[1] = lds_read_b128<OFF + 1024>(line);
    v[2] = lds_read_b128<OFF + 2048>(line), v[3] = lds_read_b128<OFF + 3072>(line);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
    const int v_row = (lane >> 3) * pitch_bytes;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      __builtin_amdgcn_raw_buffer_store_b128(v[k], r, v_row + ((k & 1) ? (park_piece ^ 4) : park_piece) * 16,
                                             soff + 8 * k * pitch_bytes, RL8_LR_STORE_AUX);
  };
  using Z0 = std::integral_constant<int, 0>;
  using Z1 = std::integral_constant<int, 1>;

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
    u32x4 la_o[4], la_ct[4], la_dh[4], la_dc[4];
    loads_landed(std::integral_constant<int, lr_behind_loads(6, 7, HEADS, F16)>{}, 8);
    if constexpr (HEADS) {
      // dL/dh_t of the heads for this lane's sixteen units: sum_q dOut[n][q] W[q][unit], the weights broadcast from the
      // park (two addresses per instruction: the lane halves), head by head
      const unsigned pk = lds0 + kLrRing * kSlot + kLrStageBytes + (wave * 2) * 4096;
      u32x4 dv = lds_read_b128<1024>(pk + n * 16);
      u32x4 wq[4];
#pragma unroll
      for (int q = 0; q < kLrHeads; ++q) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          wq[j] = q == 0   ? lds_read_b128<0>(pk + (8 * j + 4 * hh) * 4)
                  : q == 1 ? lds_read_b128<128>(pk + (8 * j + 4 * hh) * 4)
                  : q == 2 ? lds_read_b128<256>(pk + (8 * j + 4 * hh) * 4)
                           : lds_read_b128<384>(pk + (8 * j + 4 * hh) * 4);
        if (q == 0) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wq[0]), "+v"(wq[1]), "+v"(wq[2]), "+v"(wq[3]), "+v"(dv));
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wq[0]), "+v"(wq[1]), "+v"(wq[2]), "+v"(wq[3]));
        if constexpr (F16) {
          if (q == 0 && c == 0) {
            // a new step of this sequence: the power of two its dG planes carry (see the head of the file)
            float bound = cmax + emax_prev;
#pragma unroll
            for (int qq = 0; qq < kLrHeads; ++qq) bound = __builtin_fmaf(__builtin_fabsf(__uint_as_float(dv[qq])), wmax[qq], bound);
            // (down to 2^-118: the whole of fp32's normal range -- f16_bound_exponent stops at 2^-80, good for
            // operands that are weights or activations, not for the gradient of a mean over 2^25 samples times 1e-20)
            int e = __builtin_amdgcn_frexp_expf(bound * 1.0001f);
            e = e < -118 ? -118 : e;
            s_n = __builtin_amdgcn_ldexpf(1.0f, kLrTop - e);
            inv_n = __builtin_amdgcn_ldexpf(1.0f, e - kLrTop);  // (W_hh's 2^-k is applied apart: their product can underflow)
          }
        }
        const float d = __uint_as_float(dv[q]);
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const float term = d * at(wq, e);
          la_dh[e >> 2][e & 3] = __float_as_uint(q == 0 ? term : __uint_as_float(la_dh[e >> 2][e & 3]) + term);
        }
      }
    }
    unpark(Z0{}, Z0{}, la_o);
    unpark(Z0{}, Z1{}, la_ct);
    if constexpr (!HEADS) unpark(Z1{}, Z0{}, la_dh);
    unpark(Z1{}, Z1{}, la_dc);
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(la_o[0]), "+v"(la_o[1]), "+v"(la_o[2]), "+v"(la_o[3]), "+v"(la_ct[0]), "+v"(la_ct[1]), "+v"(la_ct[2]),
                   "+v"(la_ct[3]));
    asm volatile("" : "+v"(la_dh[0]), "+v"(la_dh[1]), "+v"(la_dh[2]), "+v"(la_dh[3]), "+v"(la_dc[0]), "+v"(la_dc[1]), "+v"(la_dc[2]),
                 "+v"(la_dc[3]));
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
    repark(Z0{}, Z0{}, dg[0]);
    store_park(Z0{}, Z0{}, sd.dgates, pitch_gates, c * 128 + 3 * (kHidden * 4));
  };
  auto math_b = [&](const LrStoreDesc &sd, int c) {  // -> dg[1] (i), dg[2] (g)
    u32x4 lb_i[4], lb_g[4];
    loads_landed(std::integral_constant<int, F16 ? lr_behind_loads(2, 1, HEADS, true) : lr_behind_loads(0, 1, HEADS, false)>{}, 8);
    using ZB = std::integral_constant<int, kParkB>;
    unpark(ZB{}, Z0{}, lb_i);
    unpark(ZB{}, Z1{}, lb_g);
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(lb_i[0]), "+v"(lb_i[1]), "+v"(lb_i[2]), "+v"(lb_i[3]), "+v"(lb_g[0]), "+v"(lb_g[1]), "+v"(lb_g[2]), "+v"(lb_g[3]));
    float lmax = 0.0f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const float gi = at(lb_i, e), gg = at(lb_g, e);
      dg[1][e] = dcv[e] * gg * (gi * (1.0f - gi));
      dg[2][e] = dcv[e] * gi * (1.0f - gg * gg);
      lmax = __builtin_fmaxf(lmax, __builtin_fmaxf(__builtin_fabsf(dg[1][e]), __builtin_fabsf(dg[2][e])));
    }
    fold_max(lmax);
    repark(ZB{}, Z0{}, dg[1]);
    repark(ZB{}, Z1{}, dg[2]);
    store_park(ZB{}, Z0{}, sd.dgates, pitch_gates, c * 128);
    store_park(ZB{}, Z1{}, sd.dgates, pitch_gates, c * 128 + 2 * (kHidden * 4));
  };
  auto math_c = [&](const LrStoreDesc &sd, int c) {  // -> dg[3] (f), dc out
    u32x4 lc_f[4], lc_cp[4];
    loads_landed(std::integral_constant<int, F16 ? lr_behind_loads(0, 5, HEADS, true) : lr_behind_loads(2, 5, HEADS, false)>{}, 9);
    unpark(Z1{}, Z0{}, lc_f);
    unpark(Z1{}, Z1{}, lc_cp);
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(lc_f[0]), "+v"(lc_f[1]), "+v"(lc_f[2]), "+v"(lc_f[3]), "+v"(lc_cp[0]), "+v"(lc_cp[1]), "+v"(lc_cp[2]),
                   "+v"(lc_cp[3]));
    float dc_out[16], lmax = 0.0f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const float gf = at(lc_f, e), cp = at(lc_cp, e);
      dg[3][e] = dcv[e] * cp * (gf * (1.0f - gf));
      dc_out[e] = dcv[e] * gf;
      lmax = __builtin_fmaxf(lmax, __builtin_fabsf(dg[3][e]));
      if constexpr (F16) emax = __builtin_fmaxf(emax, __builtin_fabsf(dc_out[e]));
    }
    fold_max(lmax);
    repark(Z1{}, Z0{}, dg[3]);
    repark(Z1{}, Z1{}, dc_out);
    store_park(Z1{}, Z0{}, sd.dgates, pitch_gates, c * 128 + 1 * (kHidden * 4));
    store_park(Z1{}, Z1{}, sd.dcout, pitch_state, c * 128);
  };

  // one gate-step: 16 k of the product, W_hh^T planes from ring slot K & 3, B planes from dg[K >> 1][8 (K & 1) ..+7].
  // Out tiles in pairs: two accumulators alternate, so no MFMA waits for the one before it.
  auto matrix_step = [&](auto k_tag, auto &&side) {  // side(mp): what the step issues behind the MFMAs of tile pair mp
    constexpr int K = decltype(k_tag)::value;
    constexpr int POS = K >> 1, E0 = 8 * (K & 1);
    if constexpr (F16) {
      // two fp16 planes of this sequence's dG times its power of two; W_hh^T's two planes from the slot: hi x lo, lo x hi,
      // hi x hi (smallest terms first), out tiles in pairs as below
      u32x4 bh, bl;
#pragma unroll
      for (int e = 0; e < 8; e += 2) {
        uint32_t hi, lo;
        f16_pair_scaled(dg[POS][E0 + e], dg[POS][E0 + e + 1], s_n, hi, lo);
        bh[e >> 1] = hi;
        bl[e >> 1] = lo;
      }
      const unsigned ar = a_read + (K & 3) * kSlot;
      u32x4 ah[2][2], al[2][2];  // [buffer][tile of the pair]
      auto fetch = [&](int mp, int s) {
        ah[s][0] = mp == 0 ? lds_read_b128<0 * 1024>(ar) : mp == 1 ? lds_read_b128<2 * 1024>(ar) : mp == 2 ? lds_read_b128<4 * 1024>(ar) : lds_read_b128<6 * 1024>(ar);
        ah[s][1] = mp == 0 ? lds_read_b128<1 * 1024>(ar) : mp == 1 ? lds_read_b128<3 * 1024>(ar) : mp == 2 ? lds_read_b128<5 * 1024>(ar) : lds_read_b128<7 * 1024>(ar);
        al[s][0] = mp == 0 ? lds_read_b128<8 * 1024>(ar) : mp == 1 ? lds_read_b128<10 * 1024>(ar) : mp == 2 ? lds_read_b128<12 * 1024>(ar) : lds_read_b128<14 * 1024>(ar);
        al[s][1] = mp == 0 ? lds_read_b128<9 * 1024>(ar) : mp == 1 ? lds_read_b128<11 * 1024>(ar) : mp == 2 ? lds_read_b128<13 * 1024>(ar) : lds_read_b128<15 * 1024>(ar);
      };
      fetch(0, 0);
#pragma unroll
      for (int mp = 0; mp < 4; ++mp) {
        const int s = mp & 1;
        {
          RL8_LR_T0;
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ah[s][0]), "+v"(ah[s][1]), "+v"(al[s][0]), "+v"(al[s][1]));
          RL8_LR_T1(12);
        }
        if (mp < 3) fetch(mp + 1, s ^ 1);
        f32x16 d0 = acc[2 * mp], d1 = acc[2 * mp + 1];
#define RL8_LR_MMA16(A, B)                                                                                              \
  d0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, A[s][0]), __builtin_bit_cast(half8, B), d0, 0, 0, 0); \
  d1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, A[s][1]), __builtin_bit_cast(half8, B), d1, 0, 0, 0)
        if constexpr ((kLrDiag & 4) == 0) {
          RL8_LR_MMA16(ah, bl);
          RL8_LR_MMA16(al, bh);
        }
        RL8_LR_MMA16(ah, bh);
#undef RL8_LR_MMA16
        acc[2 * mp] = d0;
        acc[2 * mp + 1] = d1;
        side(mp);
      }
      return;
    }
    u32x4 bh, bm, bl;
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
      uint32_t hi, mid, lo;
      split_pair(dg[POS][E0 + e], dg[POS][E0 + e + 1], hi, mid, lo);
      bh[e >> 1] = hi;
      bm[e >> 1] = mid;
      bl[e >> 1] = lo;
    }
    const unsigned ar = a_read + (K & 3) * kSlot;
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
      side(mp);
    }
  };

  // The barrier that opens gate-step K of a chunk: this wave's share of its chunk of W_hh^T has landed (vector-memory
  // operations complete in issue order: all but the N youngest are done) and every wave is through with the slot the
  // next request overwrites.  N = lr_behind(K): what the wave has issued behind that request, made three gate-steps
  // earlier, with the order of a gate-step's operations [row loads | request | matrix work | stores]:
  //   K        0   1   2   3   4   5   6   7
  //   loads    8   .   8   .   .   .  16   .      (B | C | A of the next chunk; all parked: direct-to-LDS)
  //   stores   .   8   .   .   .   8   .   4      (dG_i, dG_g | dG_f, dc | dG_o of the next chunk)
  //   N       40  24  32  28  28  12  20  36
  auto open_step = [&](auto k_tag) {
    constexpr int K = decltype(k_tag)::value;
    constexpr int N = RL8_LR_SAFE_WAITS ? 0 : lr_behind(K, HEADS, F16);
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
  if constexpr (F16) issue_b(nd, 0);  // (the three-park schedule requests {i, g} a chunk ahead: the first chunk's here)
  request(0, 0, 0);
  request(0, 1, 1);
  request(0, 2, 2);
  // (the counted waits of the first chunk assume a full chunk of operations behind each request: drain instead)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  math_a(sd, 0);
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
      // Within a step the order of the vector-memory operations is [parked row loads | request | stores], as the wait
      // table assumes; they go out behind the MFMA groups of the step, a few at a time.
      const LrLoadDesc &ad = wrap ? after : nd;
      auto only_request = [&](int cc, int k, int slot) {
        return [&, cc, k, slot](int mp) {
          if (2 * mp < kDma) request(cc, k, slot, 2 * mp, 2 * mp + 2);
        };
      };
      if constexpr (F16) {
        // The three-park schedule (lr_loads_at): [row loads | request | matrix work | stores] within a gate-step, as the
        // wait tables assume; a descriptor of the NEXT chunk is `ad` (the next step's, or tile's, at the wrap).
        auto loads_then_request = [&](auto &&first, auto &&second, int cc, int k, int slot) {
          return [&, cc, k, slot](int mp) {
            if (mp == 0) first();
            else if (mp == 1) second();
            else if (mp == 2) request(cc, k, slot, 0, kDma / 2);
            else request(cc, k, slot, kDma / 2, kDma);
          };
        };
        open_step(K0{});
        if (c == 0) {
#pragma unroll
          for (int mo = 0; mo < 8; ++mo) acc[mo] = f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        }
        matrix_step(K0{}, loads_then_request([&]() { issue_c(nd, c, 0); }, [&]() { issue_c(nd, c, 1); }, c, 3, 3));
        open_step(K1{});
        matrix_step(K1{}, loads_then_request([&]() { park4(0, 0, ad.gates, pitch_gates, cn * 128 + 3 * (kHidden * 4)); },
                                             [&]() { park4(0, 1, ad.cs, pitch_seq, cn * 128); }, c, 4, 0));
        { RL8_LR_T0; math_b(sd, c); RL8_LR_T1(14); }
        open_step(K2{});
        matrix_step(K2{}, loads_then_request([&]() { issue_b(ad, cn, 0); }, [&]() { issue_b(ad, cn, 1); }, c, 5, 1));
        open_step(K3{});
        matrix_step(K3{}, only_request(c, 6, 2));
        open_step(K4{});
        matrix_step(K4{}, only_request(c, 7, 3));
        open_step(K5{});
        matrix_step(K5{}, only_request(cn, 0, 0));
        { RL8_LR_T0; math_c(sd, c); RL8_LR_T1(14); }
        open_step(K6{});
        matrix_step(K6{}, [&](int mp) {  // the last third of A: the heads' rows and dOut (two loads), the carried dc (four)
          if (mp == 0) {
            const int far = 0x7fffff00;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(hwrsrc, smem + park_lds(1, 0, 0), 16,
                                                     lane < 32 ? (lane >> 3) * (kHidden * 4) + (lane & 7) * 16 : far, cn * 128, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ad.dhs, smem + park_lds(1, 0, 1), 16, lane < 32 ? lane * (l * kLrHeads * 4) : far,
                                                     0, 0, 0);
          } else if (mp == 1) {
            park4(1, 1, ad.dcin, pitch_state, cn * 128);
            asm volatile("" ::: "memory");
            request(cn, 1, 1, 0, 2);
          } else if (mp == 2) {
            request(cn, 1, 1, 2, kDma);
          }
        });
        open_step(K7{});
        matrix_step(K7{}, only_request(cn, 2, 2));
      } else {
      open_step(K0{});
      if (c == 0) {
#pragma unroll
        for (int mo = 0; mo < 8; ++mo) acc[mo] = f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
      }
      matrix_step(K0{}, [&](int mp) {
        if (mp == 0) issue_b(nd, c, 0);
        else if (mp == 1) issue_b(nd, c, 1);
        else if (mp == 2) request(c, 3, 3, 0, kDma / 2);
        else request(c, 3, 3, kDma / 2, kDma);
      });
      open_step(K1{});
      matrix_step(K1{}, only_request(c, 4, 0));
      { RL8_LR_T0; math_b(sd, c); RL8_LR_T1(14); }
      open_step(K2{});
      matrix_step(K2{}, [&](int mp) {
        if (mp == 0) issue_c(nd, c, 0);
        else if (mp == 1) issue_c(nd, c, 1);
        else if (mp == 2) request(c, 5, 1, 0, kDma / 2);
        else request(c, 5, 1, kDma / 2, kDma);
      });
      open_step(K3{});
      matrix_step(K3{}, only_request(c, 6, 2));
      open_step(K4{});
      matrix_step(K4{}, only_request(c, 7, 3));
      open_step(K5{});
      matrix_step(K5{}, only_request(cn, 0, 0));
      { RL8_LR_T0; math_c(sd, c); RL8_LR_T1(14); }
      open_step(K6{});
      matrix_step(K6{}, [&](int mp) {
        if (mp == 0) issue_a(ad, cn, 0);
        else if (mp == 1) issue_a(ad, cn, 1);
        else if (mp == 2) {
          issue_a(ad, cn, 2);
          asm volatile("" ::: "memory");
          request(cn, 1, 1, 0, 2);
        } else request(cn, 1, 1, 2, kDma);
      });
      open_step(K7{});
      matrix_step(K7{}, only_request(cn, 2, 2));
      }  // (the schedules)
      if (wrap) {
        // the step's dh is complete: it becomes the carry of the step the next arithmetic belongs to (zero for a new tile)
        if constexpr (F16) {
          // the accumulators carry this step's power of two times W_hh's; what the carries bound in the step that follows:
          // max |dh| over the sequence's 256 units (the other 128 sit in lane ^ 32), max |dc| of this step's chunks
          float m = 0.0f;
#pragma unroll
          for (int cc = 0; cc < kLrChunks; ++cc)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
              const float v = last_step ? 0.0f : (acc[cc][e] * inv_sw) * inv_n;
              dhe[e][cc] = v;
              m = __builtin_fmaxf(m, __builtin_fabsf(v));
            }
          const auto sm = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false);
          cmax = __builtin_fmaxf(__uint_as_float(sm[0]), __uint_as_float(sm[1]));
          const auto se = __builtin_amdgcn_permlane32_swap(__float_as_uint(emax), __float_as_uint(emax), false, false);
          emax_prev = last_step ? 0.0f : __builtin_fmaxf(__uint_as_float(se[0]), __uint_as_float(se[1]));
          emax = 0.0f;
        } else {
#pragma unroll
          for (int cc = 0; cc < kLrChunks; ++cc)
#pragma unroll
            for (int e = 0; e < 16; ++e) dhe[e][cc] = last_step ? 0.0f : acc[cc][e];
        }
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

// The two forms as plain kernels around the body (as a kernel TEMPLATE the host side of this body is not emitted by
// hipcc 7.2 -- the launches then link against nothing).
__global__ __launch_bounds__(kBlock, 1) void lstm_rows_backward_kernel(LrArgs a, const void *__restrict__ w_planes) {
  lstm_rows_backward_body<false>(a, w_planes);
}
__global__ __launch_bounds__(kBlock, 1) void lstm_rows_backward_heads_kernel(LrArgs a, const void *__restrict__ w_planes) {
  lstm_rows_backward_body<true>(a, w_planes);
}
__global__ __launch_bounds__(kBlock, 1) void lstm_rows_backward_heads16_kernel(LrArgs a, const void *__restrict__ w_planes) {
  lstm_rows_backward_body<true, true>(a, w_planes);
}

// max |W_hh| -> {2^k, 2^-k} with max * 2^k < 2^14 behind the fp16 planes (one workgroup; as lstm_whh_scale_kernel of
// lstm_split_kernels.hip)
__global__ __launch_bounds__(1024) void lstm_rows_scale_kernel(const float *__restrict__ w_hh, float *__restrict__ tail) {
  __shared__ float red[1024];
  const int tid = threadIdx.x;
  float mx = 0.0f;
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

// ... and the fp16 planes of W_hh^T times that power of two, same fragment order, two planes per gate-step
__global__ __launch_bounds__(kBlock) void lstm_rows_pack16_kernel(const float *__restrict__ w_hh, unsigned char *__restrict__ packed16,
                                                                  const float *__restrict__ tail) {
  const int idx = blockIdx.x * kBlock + threadIdx.x;  // (gs, mo, lane)
  if (idx >= kLrGateSteps * 8 * 64) return;
  const int lane = idx & 63, mo = (idx >> 6) & 7, gs = idx >> 9;
  const int q = lr_gate(gs & 3), c = gs >> 2;
  const int out = 32 * mo + (lane & 31);
  const float *src = w_hh + (int64_t)(kHidden * q) * kHidden + out;
  const float scale = tail[0];
  u32x4 planes[2];
#pragma unroll
  for (int e = 0; e < 8; e += 2) {
    uint32_t hi, lo;
    f16_pair_scaled(src[lr_in_unit(c, lane >> 5, e) * kHidden], src[lr_in_unit(c, lane >> 5, e + 1) * kHidden], scale, hi, lo);
    planes[0][e >> 1] = hi;
    planes[1][e >> 1] = lo;
  }
#pragma unroll
  for (int p = 0; p < 2; ++p)
    *reinterpret_cast<u32x4 *>(packed16 + (int64_t)gs * lr_slot_bytes(true) + (p * 8 + mo) * 1024 + lane * 16) = planes[p];
}

}  // namespace rl8

using namespace rl8;

RL8_API int64_t rl8_lstm_rows_backward_pack_bytes(void) { return kLrPackTotalBytes; }

RL8_API int rl8_lstm_rows_backward_pack(const float *w_hh, void *packed, void *stream) {
  if (!w_hh || !packed) return RL8_ENULL;
  if (!aligned16(packed)) return RL8_EALIGN;
  lstm_rows_pack_kernel<<<kLrGateSteps * 8 * 64 / kBlock, kBlock, 0, (hipStream_t)stream>>>(w_hh, static_cast<uint32_t *>(packed));
  // (round 6) behind the bf16 planes: the fp16 planes of the HEADS form and W_hh's power of two
  unsigned char *p16 = static_cast<unsigned char *>(packed) + kLrPackedBytes;
  float *tail = reinterpret_cast<float *>(p16 + kLrPacked16Bytes);
  lstm_rows_scale_kernel<<<1, 1024, 0, (hipStream_t)stream>>>(w_hh, tail);
  lstm_rows_pack16_kernel<<<kLrGateSteps * 8 * 64 / kBlock, kBlock, 0, (hipStream_t)stream>>>(w_hh, p16, tail);
  return launch_status();
}

static int rows_backward(int64_t b, int l, const float *c0, const float *gates, const float *cs, const float *dhs,
                         const float *heads_w, const void *packed, float *dgates, float *dc_scratch, uint32_t *dg_bound_out,
                         void *stream) {
  if (!c0 || !gates || !cs || !dhs || !packed || !dgates || !dc_scratch) return RL8_ENULL;
  if (b <= 0 || l <= 0) return RL8_ESIZE;
  // a wave addresses its 32 sequences with 32-bit offsets
  if ((int64_t)32 * l * 4 * kHidden * 4 >= (int64_t)1 << 31) return RL8_ESIZE;
  if (!aligned16(packed) || !aligned16(c0) || !aligned16(gates) || !aligned16(cs) || !aligned16(dhs) || !aligned16(dgates) ||
      !aligned16(dc_scratch) || !aligned16(heads_w))
    return RL8_EALIGN;
  static LdsOptIn lds_attr_set_0;
  if (const int e_lds_attr_set_0 = allow_dynamic_lds(lds_attr_set_0, reinterpret_cast<const void *>(&lstm_rows_backward_kernel), kLrLdsBytes)) return e_lds_attr_set_0;
  static LdsOptIn lds_attr_set_1;
  if (const int e_lds_attr_set_1 = allow_dynamic_lds(lds_attr_set_1, reinterpret_cast<const void *>(&lstm_rows_backward_heads_kernel), kLrLdsBytes)) return e_lds_attr_set_1;
  const int64_t tiles = (b + kLrRows - 1) / kLrRows;
  const int grid = (int)(tiles < kCUs ? tiles : kCUs);
  unsigned long long *stamps = nullptr;
#ifdef RL8_LR_STAMP
  if (const char *v = getenv("RL8_LR_STAMP_PTR")) stamps = reinterpret_cast<unsigned long long *>(strtoull(v, nullptr, 0));
#endif
  if (dg_bound_out && hipMemsetAsync(dg_bound_out, 0, 4, (hipStream_t)stream) != hipSuccess) return launch_status();
  const LrArgs args = {c0, gates, cs, dhs, heads_w, dgates, dc_scratch, b, l, stamps, dg_bound_out};
  // the HEADS form on fp16 planes (three plane products) unless RL8_AMD_LSTM_BACKWARD_PLANES=bf16 (six; read per call)
  const char *planes = getenv("RL8_AMD_LSTM_BACKWARD_PLANES");
  const bool f16 = heads_w != nullptr && !(planes && planes[0] == 'b');
  if (f16) {
    static LdsOptIn lds_attr_set_2;
    if (const int e = allow_dynamic_lds(lds_attr_set_2, reinterpret_cast<const void *>(&lstm_rows_backward_heads16_kernel), lr_lds_bytes(true))) return e;
    lstm_rows_backward_heads16_kernel<<<grid, kBlock, lr_lds_bytes(true), (hipStream_t)stream>>>(args, packed);
  } else if (heads_w) lstm_rows_backward_heads_kernel<<<grid, kBlock, kLrLdsBytes, (hipStream_t)stream>>>(args, packed);
  else lstm_rows_backward_kernel<<<grid, kBlock, kLrLdsBytes, (hipStream_t)stream>>>(args, packed);
  return launch_status();
}

RL8_API int rl8_lstm_rows_backward_f32(int64_t b, int l, const float *c0, const float *gates, const float *cs, const float *dhs,
                                       const void *packed, float *dgates, float *dc_scratch, uint32_t *dg_bound_out,
                                       void *stream) {
  return rows_backward(b, l, c0, gates, cs, dhs, nullptr, packed, dgates, dc_scratch, dg_bound_out, stream);
}

// The same with dL/dh_t = heads_dout[b][l][0..3] x heads_w [4][256] formed inside (both zero-padded to four heads): the
// output heads' data gradient is never written to memory nor read back.
RL8_API int rl8_lstm_rows_backward_heads_f32(int64_t b, int l, const float *c0, const float *gates, const float *cs,
                                             const float *heads_dout, const float *heads_w, const void *packed, float *dgates,
                                             float *dc_scratch, uint32_t *dg_bound_out, void *stream) {
  if (!heads_w) return RL8_ENULL;
  return rows_backward(b, l, c0, gates, cs, heads_dout, heads_w, packed, dgates, dc_scratch, dg_bound_out, stream);
}
