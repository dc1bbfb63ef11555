// N1: the default policy / value tower -- Linear(d_in,256) ReLU
// Linear(256,256) ReLU Linear(256,n_out) (src/rl8/models/_feedforward.py:336-362,
// src/rl8/nn/modules/mlp.py:12-52) -- as ONE fp32 kernel per direction that keeps
// the 256-wide activations on chip.
//
// Why: once the memory-bound kernels are fused, >99.9 % of collect()+step() is
// these towers.  Eager PyTorch runs them as GEMM + bias + ReLU launches whose
// [rows x 256] fp32 activations make an HBM round trip each (2 KB per sample per
// layer, forward and backward); at 2^20 x 32 samples that traffic costs as much
// as the GEMMs.  Here a workgroup walks 64-row tiles: layer 1 on the VALU
// (d_in is tiny: an outer product), layer 2 on the matrix cores with fp32
// MFMA (v_mfma_f32_32x32x2_f32: exact fp32 fma chains, no precision change),
// the head on the VALU, activations in LDS / accumulators throughout.  HBM sees
// the observations, the outputs and -- for training only -- one store of h1 / h2.
//
// MFMA operand maps (gfx950, 32x32x2 f32): lane l holds A[i = l&31][k = l>>5],
// B[k = l>>5][j = l&31]; D[row = (r&3) + 8*(r>>2) + 4*(l>>5)][col = l&31], r<16.
// The k index is permuted so that each lane's four consecutive k-steps use four
// consecutive floats: k = 8g + 4*(l>>5) + e, e = 0..3 (sums are order-free up to
// fp32 rounding).  W2 is pre-packed in that fragment order (rl8_mlp_pack_w2_f32)
// so a wave's B operand for one MFMA is one coalesced 256-byte load.
//
// LDS: one [64][257] fp32 tile (h1, overwritten by h2); the odd row stride makes
// both the row walks of the MFMA A operand and the column walks of the VALU
// phases bank-conflict-free.
#include "mfma_tile.hip.h"

namespace rl8 {

// Kernel-tuning builds only (tools/diag_mlp.sh): -DRL8_DIAG_SKIP=<bits> drops
// one memory stream of a tower kernel so its cost can be read off the microbench
// (forward: 8 h2 store, 32 h1 store; backward: 64 h2 loads, 128 dZ2 stores,
// 256 h1 loads).  The shipped library is built with 0.
#ifndef RL8_DIAG_SKIP
#define RL8_DIAG_SKIP 0
#endif
constexpr int kDiagSkip = RL8_DIAG_SKIP;

// w2 [256 out][256 in] row-major -> fragment order, one 256-byte run per
// (N-tile n, k-group g, element e) so that a wave's B operand for one MFMA is a
// single coalesced dword load:
// packed[((n*32 + g)*4 + e)*64 + l] = w2[32n + (l&31)][8g + 4(l>>5) + e]
__global__ __launch_bounds__(kBlock) void mlp_pack_w2_kernel(const float *__restrict__ w2,
                                                             float *__restrict__ packed) {
  const int idx = blockIdx.x * kBlock + threadIdx.x;
  if (idx >= kHidden * kHidden) return;
  const int l = idx & 63, e = (idx >> 6) & 3, g = (idx >> 8) & 31, n = idx >> 13;
  packed[idx] = w2[(32 * n + (l & 31)) * kHidden + 8 * g + 4 * (l >> 5) + e];
}

// Transposed packing for the backward data-gradient GEMM dH1 = dZ2 x W2:
// B[k = j][n = i] = w2[j][i]:  packedT[((n*32 + g)*4 + e)*64 + l] = w2[8g + 4(l>>5) + e][32n + (l&31)]
__global__ __launch_bounds__(kBlock) void mlp_pack_w2t_kernel(const float *__restrict__ w2,
                                                              float *__restrict__ packed) {
  const int idx = blockIdx.x * kBlock + threadIdx.x;
  if (idx >= kHidden * kHidden) return;
  const int l = idx & 63, e = (idx >> 6) & 3, g = (idx >> 8) & 31, n = idx >> 13;
  packed[idx] = w2[(8 * g + 4 * (l >> 5) + e) * kHidden + 32 * n + (l & 31)];
}

// Kernel-tuning builds only (-DRL8_PHASE_TRACE): shader-clock stamps at the phase
// boundaries of the tower kernels, tile iterations 4..7 of every workgroup, read
// back with rl8_debug_phase_trace().  Compiled out of the shipped library.
#ifdef RL8_PHASE_TRACE
constexpr int kTraceSlots = 12, kTraceTiles = 4;
__device__ unsigned long long g_phase_trace[512 * 4 * kTraceTiles * kTraceSlots];
__device__ __forceinline__ void trace_stamp(int iteration, int slot) {
  if (iteration >= 4 && iteration < 4 + kTraceTiles && (threadIdx.x & 63) == 0)
    g_phase_trace[((blockIdx.x * 4 + (threadIdx.x >> 6)) * kTraceTiles + (iteration - 4)) * kTraceSlots + slot] =
        __builtin_amdgcn_s_memtime();
}
#define RL8_TRACE(iteration, slot) trace_stamp(iteration, slot)
#else
#define RL8_TRACE(iteration, slot)
#endif

// Forward of one tower over m rows.  SAVE: also store h1 / h2 ([m][256] each) for
// the backward kernels.  LDS: ONE [64][257] tile (h1, then h2 in place once the
// matrix loop has consumed h1) + the observation tile + the head weights = 78 KB
// => two workgroups per CU (the second hides the first's barrier / memory waits;
// see the cost model above for what it cannot hide).
// DIN / NOUT: compile-time input / output widths (0 = run-time, up to
// kMaxIn / kMaxOut, zero-padded so the loops stay branch-free).
template <int DIN, int NOUT, bool SAVE>
__global__ __launch_bounds__(kBlock, 2) void mlp_tower_forward_kernel(
    const float *__restrict__ x, int64_t m, int d_in_rt, const float *__restrict__ w1,
    const float *__restrict__ b1, const float4 *__restrict__ w2p, const float *__restrict__ b2,
    const float *__restrict__ w3, const float *__restrict__ b3, int n_out_rt,
    float *__restrict__ out, float *__restrict__ save_h1, float *__restrict__ save_h2) {
  constexpr int kIn = DIN > 0 ? DIN : kMaxIn;
  constexpr int kOut = NOUT > 0 ? pad_out(NOUT) : kMaxOut;
  typedef float outvec __attribute__((ext_vector_type(kOut)));
  // Full unrolling turns every LDS offset into an immediate, but with wide
  // inputs / heads it also hoists hundreds of LDS reads into registers.
  constexpr int kLayer1Unroll = kIn <= 2 ? kTileRows : 8;
  constexpr int kHeadUnroll = kOut <= 2 ? 64 : 8;
  constexpr int kIlp = 8;  // independent rows per stage of a VALU phase
  const int d_in = DIN > 0 ? DIN : d_in_rt;
  const int n_out = NOUT > 0 ? NOUT : n_out_rt;
  extern __shared__ float lds[];
  // Small arrays first: their (uniform) addresses then fit the 16-bit immediate
  // offset of the LDS instructions and need no address register.
  float *xs = lds;                                  // [64][kIn], zero-padded columns
  float *w3s = xs + kTileRows * kMaxIn;             // [256][kOut]: head weights, unit-major
  float *ht = w3s + kMaxOut * kHidden;              // [64][257]: h1, later h2
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hh = lane >> 5;

  // Per-thread constants: column `tid` of layer 1 (zero-padded); head weights to
  // LDS as [unit][output] so one wide LDS read feeds one packed fma.
  float w1r[kIn];
#pragma unroll
  for (int i = 0; i < kIn; ++i) w1r[i] = i < d_in ? w1[tid * d_in + i] : 0.0f;
  const float b1r = b1[tid];
  for (int idx = tid; idx < kOut * kHidden; idx += kBlock) {
    const int j = idx / kOut, q = idx - j * kOut;
    w3s[idx] = q < n_out ? w3[q * kHidden + j] : 0.0f;
  }
  float b2r[2];
  b2r[0] = b2[64 * wave + (lane & 31)];
  b2r[1] = b2[64 * wave + 32 + (lane & 31)];
  outvec b3r;
#pragma unroll
  for (int q = 0; q < kOut; ++q) b3r[q] = q < n_out ? b3[q] : 0.0f;
  TileGemm gemm(buffer_rsrc(w2p, kHidden * kHidden * 4), wave, lane);
  __builtin_amdgcn_s_setprio(kValuPhasePriority);
  if constexpr (DIN == 0) {
    for (int idx = tid; idx < kTileRows * kMaxIn; idx += kBlock) xs[idx] = 0.0f;
  }

  const int64_t tiles = (m + kTileRows - 1) / kTileRows;
  // Observation tile: element e of the [64][d_in] tile is owned by thread e
  // (+256, ...); fetched one tile ahead so its HBM latency hides under the
  // matrix phase.
  constexpr int kXPerThread = (kTileRows * kIn + kBlock - 1) / kBlock;
  float xreg[kXPerThread];
  auto fetch_x = [&](int64_t tile) {
    const int64_t r0 = tile * kTileRows;
#pragma unroll
    for (int u = 0; u < kXPerThread; ++u) {
      const int idx = tid + u * kBlock;
      const int64_t g = r0 * d_in + idx;
      xreg[u] = (idx < kTileRows * d_in && g < m * d_in) ? x[g] : 0.0f;
    }
  };
  if ((int64_t)blockIdx.x < tiles) fetch_x(blockIdx.x);

  [[maybe_unused]] int iteration = -1;
  for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    ++iteration;
    RL8_TRACE(iteration, 0);
    const int64_t r0 = tile * kTileRows;
    const int rows = (int)((m - r0) < kTileRows ? (m - r0) : kTileRows);
    // Stores of this tile's h1 / h2 rows: descriptor based at the tile, sized to
    // its valid rows (stores past the end of a partial tile are dropped).
    const __amdgpu_buffer_rsrc_t h1rsrc = buffer_rsrc(SAVE ? save_h1 + r0 * kHidden : nullptr, rows * kHidden * 4);
    const __amdgpu_buffer_rsrc_t h2rsrc = buffer_rsrc(SAVE ? save_h2 + r0 * kHidden : nullptr, rows * kHidden * 4);
    __syncthreads();  // previous tile's readers of xs / ht are done
#pragma unroll
    for (int u = 0; u < kXPerThread; ++u) {
      const int idx = tid + u * kBlock;
      if (idx < kTileRows * d_in) {
        const int s = DIN > 0 ? idx / kIn : idx / d_in;
        xs[s * kIn + (idx - s * d_in)] = xreg[u];
      }
    }
    __syncthreads();
    RL8_TRACE(iteration, 1);
    gemm.prefetch();  // first weight fragments, requested ahead of this tile's h1 stores
    // Layer 1 (VALU): thread = output column; per row d_in fmas + one max, the
    // observation from a broadcast LDS read, offsets all immediates.
    // Blocks of kIlp rows, stage by stage: a wave issues in program order, and a
    // dependent pair back to back leaves a gap in which the SIMD's other wave
    // starts a 64-cycle MFMA -- measured, a chain-ordered VALU phase advanced one
    // instruction per MFMA of its neighbour (~80 cycles each).  Independent
    // instructions in a row issue as a burst instead.
#pragma unroll kLayer1Unroll / kIlp
    for (int s0 = 0; s0 < kTileRows; s0 += kIlp) {
      float v[kIlp];
#pragma unroll
      for (int u = 0; u < kIlp; ++u) v[u] = b1r;
#pragma unroll
      for (int i = 0; i < kIn; ++i) {
        float xv[kIlp];
#pragma unroll
        for (int u = 0; u < kIlp; ++u) xv[u] = xs[(s0 + u) * kIn + i];
#pragma unroll
        for (int u = 0; u < kIlp; ++u) v[u] = __builtin_fmaf(xv[u], w1r[i], v[u]);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < kIlp; ++u) v[u] = relu1(v[u]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < kIlp; ++u) {
        ht[(s0 + u) * kLdsStride + tid] = v[u];
        if constexpr (SAVE && !(kDiagSkip & 32)) buffer_store_f32(v[u], h1rsrc, tid * 4, (s0 + u) * (kHidden * 4));
      }
    }
    RL8_TRACE(iteration, 2);
    __syncthreads();
    RL8_TRACE(iteration, 3);
    if (tile + gridDim.x < tiles) fetch_x(tile + gridDim.x);  // lands during the matrix phase
    // Layer 2 (MFMA).
    f32x16 acc[2][2];
    gemm.run(ht, acc);
    RL8_TRACE(iteration, 4);
    __syncthreads();  // every wave has read all of h1: the tile may be overwritten
    RL8_TRACE(iteration, 5);
    // bias (packed adds over register pairs) + ReLU, accumulators -> h2 (in place
    // of h1) and, when saving, straight to HBM (each store covers two 128-byte
    // row segments).  Stage by stage over 16 values at a time (see layer 1).
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const int j = 64 * wave + 32 * nt + (lane & 31);
        f32x2 pre[8];
#pragma unroll
        for (int r = 0; r < 16; r += 2)
          pre[r / 2] = f32x2{acc[mt][nt][r], acc[mt][nt][r + 1]} + f32x2{b2r[nt], b2r[nt]};
        __builtin_amdgcn_sched_barrier(0);
        float v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = relu1(pre[r / 2][r & 1]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int sr = 32 * mt + (r & 3) + 8 * (r >> 2);  // + 4*hh
          ht[(sr + 4 * hh) * kLdsStride + j] = v[r];
          if constexpr (SAVE && !(kDiagSkip & 8))
            buffer_store_f32(v[r], h2rsrc, (4 * hh * kHidden + j) * 4, sr * (kHidden * 4));
        }
      }
    RL8_TRACE(iteration, 6);
    __syncthreads();
    RL8_TRACE(iteration, 7);
    // Head (VALU): 4 lanes per row; lane (a, q4) of wave w owns row 4a + w and the
    // units j = q4 (mod 4) -- LDS bank 4a + q4 + const, so the 64 lanes hit 64
    // different banks with immediate offsets only -- one packed fma per unit for
    // a pair of outputs, then two lane shuffles.
    {
      const int s = 4 * (lane >> 2) + wave, q4 = lane & 3;
      // four accumulator sets so that consecutive fmas are independent
      outvec oacc[4];
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int q = 0; q < kOut; ++q) oacc[c][q] = 0.0f;
      const float *row = ht + s * kLdsStride + q4;
      const outvec *wq = reinterpret_cast<const outvec *>(w3s) + q4;
#pragma unroll kHeadUnroll
      for (int jj = 0; jj < 64; ++jj) {
        const float hv = row[4 * jj];
        outvec hvv;
#pragma unroll
        for (int q = 0; q < kOut; ++q) hvv[q] = hv;
        if constexpr (kOut == 1)
          oacc[jj & 3][0] = __builtin_fmaf(hv, wq[4 * jj][0], oacc[jj & 3][0]);
        else
          oacc[jj & 3] = __builtin_elementwise_fma(hvv, wq[4 * jj], oacc[jj & 3]);
      }
      const outvec o = (oacc[0] + oacc[1]) + (oacc[2] + oacc[3]);
#pragma unroll
      for (int q = 0; q < kOut; ++q) {
        float v = o[q];
        v += __shfl_xor(v, 1, kWave);
        v += __shfl_xor(v, 2, kWave);
        if (q < n_out && q4 == 0 && s < rows) out[(r0 + s) * n_out + q] = v + b3r[q];
      }
    }
    RL8_TRACE(iteration, 8);
  }
}

// Backward of one tower over m rows, data-gradient half ("dgrad"):
//   dZ2 = (dOut x W3) * (h2 > 0)            VALU, thread = column, two rows per packed op
//   dH1 = dZ2 x W2                          MFMA (W2 packed transposed)
//   dZ1 = dH1 * (h1 > 0)                    accumulator epilogue
// and every small parameter gradient on the way:
//   dW3 = dOut^T h2, db3 = sum dOut, db2 = sum dZ2, dW1 = dZ1^T x, db1 = sum dZ1.
// dZ2 is also stored ([m][256]) for the one remaining large product,
// dW2 = dZ2^T h1 (rl8_mlp_wgrad_f32).  h1 arrives in accumulator layout by
// buffer loads issued ahead of the matrix loop (its sign is the ReLU mask).
// Small gradients leave as one row of per-workgroup partials,
//   [dW1 (256*d_in) | db1 (256) | db2 (256) | dW3 (n_out*256) | db3 (n_out)],
// summed on the host side in a fixed order (bitwise reproducible, no atomics).
template <int DIN, int NOUT>
__global__ __launch_bounds__(kBlock, (DIN == 0 || NOUT == 0) ? 1 : 2) void mlp_tower_backward_kernel(
    const float *__restrict__ x, const float *__restrict__ h1, const float *__restrict__ h2,
    const float *__restrict__ dout, int64_t m, int d_in_rt, const float4 *__restrict__ w2tp,
    const float *__restrict__ w3, int n_out_rt, float *__restrict__ dz2_out,
    float *__restrict__ partials, int partial_stride) {
  constexpr int kIn = DIN > 0 ? DIN : kMaxIn;
  constexpr int kOut = NOUT > 0 ? pad_out(NOUT) : kMaxOut;
  const int d_in = DIN > 0 ? DIN : d_in_rt;
  const int n_out = NOUT > 0 ? NOUT : n_out_rt;
  extern __shared__ float lds[];
  // Small arrays first (uniform addresses within the 16-bit LDS immediate); the
  // observation / dOut tiles are double-buffered (tile parity) so that the next
  // tile can be staged while slower waves still fold the current one.
  float *xs0 = lds;                                  // [2][64][kIn], zero-padded columns
  float *ds0 = xs0 + 2 * kTileRows * kMaxIn;         // [2][kOut][64] dOut tile (output-major), zero-padded
  float *zt = ds0 + 2 * kTileRows * kMaxOut;         // [64][257]: h2 -> dZ2
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hh = lane >> 5;

  float w3r[kOut];  // column `tid` of W3, zero-padded
#pragma unroll
  for (int q = 0; q < kOut; ++q) w3r[q] = q < n_out ? w3[q * kHidden + tid] : 0.0f;
  // Running sums owned by this thread; pairs = (even rows, odd rows) or
  // (column nt = 0, column nt = 1), folded once at the end.
  constexpr int kPairs = kOut <= 2 ? 4 : 2;  // row pairs per block of phase 1 (one accumulator set each)
  constexpr int kSets = 4;                   // accumulator sets of phase 3
  f32x2 dw3[kOut][kPairs], db2[kPairs];
#pragma unroll
  for (int u = 0; u < kPairs; ++u) {
    db2[u] = f32x2{0.0f, 0.0f};
#pragma unroll
    for (int q = 0; q < kOut; ++q) dw3[q][u] = f32x2{0.0f, 0.0f};
  }
  f32x2 dw1[kIn][kSets], db1[kSets];  // columns 64*wave + 32*nt + (lane&31), this half's rows
#pragma unroll
  for (int c = 0; c < kSets; ++c) {
    db1[c] = f32x2{0.0f, 0.0f};
#pragma unroll
    for (int i = 0; i < kIn; ++i) dw1[i][c] = f32x2{0.0f, 0.0f};
  }
  constexpr int kXPerThread = (kTileRows * kIn + kBlock - 1) / kBlock;
  constexpr int kDoutPerThread = (kTileRows * kOut + kBlock - 1) / kBlock;
  float db3 = 0.0f;  // output tid % kOut, this thread's share of the rows
  TileGemm gemm(buffer_rsrc(w2tp, kHidden * kHidden * 4), wave, lane);
  __builtin_amdgcn_s_setprio(kValuPhasePriority);
  if constexpr (DIN == 0) {
    for (int idx = tid; idx < 2 * kTileRows * kMaxIn; idx += kBlock) xs0[idx] = 0.0f;
    __syncthreads();
  }

  const int64_t tiles = (m + kTileRows - 1) / kTileRows;
  // Everything a tile reads from HBM is requested one phase or more ahead:
  //   h2 tile   -> LDS by direct-to-LDS loads (16 one-row loads per wave; rows past
  //               the end arrive as zeros), issued as soon as the previous tile's
  //               matrix loop has released the LDS tile;
  //   x, dOut   -> registers at the same point, to LDS at the top of the tile;
  //   h1        -> registers in accumulator layout (its sign is the ReLU mask):
  //               upper half ahead of phase 1, lower half behind the matrix loop
  //               (all 64 live across the loop would not fit two waves per SIMD);
  //   W2 frags  -> TileGemm::prefetch ahead of phase 1's stores.
  float xreg[kXPerThread], dreg[kDoutPerThread];
  auto request_tile = [&](int64_t tile) {
    const int64_t r0 = tile * kTileRows;
    const int rows = (int)((m - r0) < kTileRows ? (m - r0) : kTileRows);
    if constexpr (!(kDiagSkip & 64)) {
      const __amdgpu_buffer_rsrc_t h2rsrc = buffer_rsrc(h2 + r0 * kHidden, rows * kHidden * 4);
#pragma unroll
      for (int u = 0; u < kTileRows / 4; ++u) {
        const int row = wave + 4 * u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(h2rsrc, zt + row * kLdsStride, 16, lane * 16,
                                                 row * (kHidden * 4), 0, 0);
      }
    }
#pragma unroll
    for (int u = 0; u < kXPerThread; ++u) {
      const int idx = tid + u * kBlock;
      xreg[u] = (idx < kTileRows * d_in && idx < rows * d_in) ? x[r0 * d_in + idx] : 0.0f;
    }
#pragma unroll
    for (int u = 0; u < kDoutPerThread; ++u) {
      const int idx = tid + u * kBlock;
      const int s = idx / kOut, q = idx - s * kOut;
      dreg[u] = (idx < kTileRows * kOut && s < rows && q < n_out) ? dout[(r0 + s) * n_out + q] : 0.0f;
    }
  };
  if ((int64_t)blockIdx.x < tiles) request_tile(blockIdx.x);
  int parity = 0;
  [[maybe_unused]] int iteration = -1;
  for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x, parity ^= 1) {
    ++iteration;
    RL8_TRACE(iteration, 0);
    const int64_t r0 = tile * kTileRows;
    const int rows = (int)((m - r0) < kTileRows ? (m - r0) : kTileRows);
    const __amdgpu_buffer_rsrc_t h1rsrc = buffer_rsrc(h1 + r0 * kHidden, rows * kHidden * 4);
    const __amdgpu_buffer_rsrc_t dzrsrc = buffer_rsrc(dz2_out + r0 * kHidden, rows * kHidden * 4);
    float *xs = xs0 + parity * kTileRows * kMaxIn;
    float *ds = ds0 + parity * kTileRows * kMaxOut;
#pragma unroll
    for (int u = 0; u < kXPerThread; ++u) {
      const int idx = tid + u * kBlock;
      if (idx < kTileRows * d_in) {
        const int s = DIN > 0 ? idx / kIn : idx / d_in;
        xs[s * kIn + (idx - s * d_in)] = xreg[u];
      }
    }
#pragma unroll
    for (int u = 0; u < kDoutPerThread; ++u) {
      const int idx = tid + u * kBlock;
      if (idx < kTileRows * kOut) {
        const int s = idx / kOut, q = idx - s * kOut;
        ds[q * kTileRows + s] = dreg[u];  // rows (s, s+1) adjacent: one 8-byte read per packed operand
        db3 += dreg[u];
      }
    }
    wait_vmcnt0();    // this wave's h2 rows have landed
    RL8_TRACE(iteration, 1);
    __syncthreads();  // ... everyone's have; xs / ds are written
    RL8_TRACE(iteration, 2);
    float h1a[2][2][16];
    auto fetch_h1 = [&](int mt) {
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int sr = 32 * mt + (r & 3) + 8 * (r >> 2);
          h1a[mt][nt][r] = (kDiagSkip & 256) ? 1.0f : buffer_load_f32(
              h1rsrc, (4 * hh * kHidden + 64 * wave + 32 * nt + (lane & 31)) * 4, sr * (kHidden * 4));
        }
    };
    gemm.prefetch();
    fetch_h1(0);
    // Phase 1 (VALU, thread = column j, two rows per packed op): dZ2 and the head
    // gradients.  Four row pairs per block, stage by stage, with one accumulator
    // set per pair: consecutive instructions are independent, so the phase
    // issues as a burst (see the forward kernel's layer 1).  (Blocks, not a full
    // unroll: the scheduler would hoist every LDS read of the phase and spill.)
#pragma unroll 2
    for (int s0 = 0; s0 < kTileRows; s0 += 2 * kPairs) {
      f32x2 hv[kPairs], d[kOut][kPairs], g[kPairs];
#pragma unroll
      for (int u = 0; u < kPairs; ++u) {
        const int s = s0 + 2 * u;
        hv[u] = f32x2{zt[s * kLdsStride + tid], zt[(s + 1) * kLdsStride + tid]};
#pragma unroll
        for (int q = 0; q < kOut; ++q) d[q][u] = *reinterpret_cast<const f32x2 *>(ds + q * kTileRows + s);
      }
#pragma unroll
      for (int u = 0; u < kPairs; ++u) g[u] = d[0][u] * f32x2{w3r[0], w3r[0]};
#pragma unroll
      for (int q = 1; q < kOut; ++q)
#pragma unroll
        for (int u = 0; u < kPairs; ++u) g[u] = __builtin_elementwise_fma(d[q][u], f32x2{w3r[q], w3r[q]}, g[u]);
#pragma unroll
      for (int q = 0; q < kOut; ++q)
#pragma unroll
        for (int u = 0; u < kPairs; ++u) dw3[q][u] = __builtin_elementwise_fma(d[q][u], hv[u], dw3[q][u]);
      __builtin_amdgcn_sched_barrier(0);
      f32x2 dz[kPairs];
      {
        unsigned long long gate[kPairs][2];
#pragma unroll
        for (int u = 0; u < kPairs; ++u) {
          gate[u][0] = positive_mask(hv[u].x);
          gate[u][1] = positive_mask(hv[u].y);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < kPairs; ++u)
          dz[u] = f32x2{select_or_zero(gate[u][0], g[u].x), select_or_zero(gate[u][1], g[u].y)};
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < kPairs; ++u) db2[u] += dz[u];
#pragma unroll
      for (int u = 0; u < kPairs; ++u) {
        const int s = s0 + 2 * u;
        zt[s * kLdsStride + tid] = dz[u].x;
        zt[(s + 1) * kLdsStride + tid] = dz[u].y;
        if constexpr (!(kDiagSkip & 128)) {
          buffer_store_f32(dz[u].x, dzrsrc, tid * 4, s * (kHidden * 4));
          buffer_store_f32(dz[u].y, dzrsrc, tid * 4, (s + 1) * (kHidden * 4));
        }
      }
    }
    RL8_TRACE(iteration, 3);
    __syncthreads();
    RL8_TRACE(iteration, 4);
    // Phase 2 (MFMA): dH1 = dZ2 x W2.
    f32x16 acc[2][2];
    gemm.run(zt, acc);
    RL8_TRACE(iteration, 5);
    __syncthreads();  // every wave is done with the dZ2 tile: the next h2 tile may land
    RL8_TRACE(iteration, 6);
    if (tile + gridDim.x < tiles) request_tile(tile + gridDim.x);
    fetch_h1(1);
    RL8_TRACE(iteration, 7);
    // Phase 3: dZ1 = dH1 * (h1 > 0); fold into dW1 / db1, the two columns of a
    // lane as one packed op; four values per stage, one accumulator set each.
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r0 = 0; r0 < 16; r0 += kSets) {
        f32x2 dz[kSets];
        {
          unsigned long long gate[kSets][2];
#pragma unroll
          for (int u = 0; u < kSets; ++u) {
            gate[u][0] = positive_mask(h1a[mt][0][r0 + u]);
            gate[u][1] = positive_mask(h1a[mt][1][r0 + u]);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int u = 0; u < kSets; ++u)
            dz[u] = f32x2{select_or_zero(gate[u][0], acc[mt][0][r0 + u]),
                          select_or_zero(gate[u][1], acc[mt][1][r0 + u])};
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < kSets; ++u) db1[u] += dz[u];
#pragma unroll
        for (int c = 0; c < kIn; ++c)
#pragma unroll
          for (int u = 0; u < kSets; ++u) {
            const int r = r0 + u;
            const int s = 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * hh;
            const float xv = xs[s * kIn + c];
            dw1[c][u] = __builtin_elementwise_fma(dz[u], f32x2{xv, xv}, dw1[c][u]);
          }
      }
    RL8_TRACE(iteration, 8);
  }
  // Workgroup partial row.
  float *row = partials + (int64_t)blockIdx.x * partial_stride;
  const int off_db1 = kHidden * d_in, off_db2 = off_db1 + kHidden, off_dw3 = off_db2 + kHidden;
  const int off_db3 = off_dw3 + n_out * kHidden;
  // fold the accumulator sets (fixed order)
  f32x2 db1s = db1[0], db2s = db2[0];
#pragma unroll
  for (int c = 1; c < kSets; ++c) db1s += db1[c];
#pragma unroll
  for (int u = 1; u < kPairs; ++u) db2s += db2[u];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int i = 64 * wave + 32 * nt + (lane & 31);
    const float b = db1s[nt] + __shfl_xor(db1s[nt], 32, kWave);
    if (hh == 0) row[off_db1 + i] = b;
#pragma unroll
    for (int c = 0; c < kIn; ++c) {
      f32x2 ws = dw1[c][0];
#pragma unroll
      for (int u = 1; u < kSets; ++u) ws += dw1[c][u];
      const float w = ws[nt] + __shfl_xor(ws[nt], 32, kWave);
      if (hh == 0 && c < d_in) row[i * d_in + c] = w;
    }
  }
  row[off_db2 + tid] = db2s.x + db2s.y;
#pragma unroll
  for (int q = 0; q < kOut; ++q) {
    f32x2 ws = dw3[q][0];
#pragma unroll
    for (int u = 1; u < kPairs; ++u) ws += dw3[q][u];
    if (q < n_out) row[off_dw3 + q * kHidden + tid] = ws.x + ws.y;
  }
  // db3: thread t holds a share of output t % kOut; fold through LDS.
  __syncthreads();
  ds0[tid] = db3;
  __syncthreads();
  if (tid < n_out) {
    float sum = 0.0f;
    for (int t = tid; t < kBlock; t += kOut) sum += ds0[t];
    row[off_db3 + tid] = sum;
  }
}

inline size_t mlp_backward_lds_bytes() {
  return sizeof(float) * (kTileRows * kLdsStride + 2 * kTileRows * kMaxIn + 2 * kTileRows * kMaxOut);
}

inline size_t mlp_forward_lds_bytes() {
  return sizeof(float) * (kTileRows * kLdsStride + kTileRows * kMaxIn + kMaxOut * kHidden);
}

}  // namespace rl8

using namespace rl8;

#ifdef RL8_PHASE_TRACE
RL8_API int rl8_debug_phase_trace(unsigned long long *host_dst) {
  return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(g_phase_trace), sizeof(g_phase_trace));
}
#endif

RL8_API int rl8_mlp_pack_w2_f32(const float *w2, float *w2_packed, int transposed, void *stream) {
  if (!w2 || !w2_packed) return RL8_ENULL;
  if (!aligned16(w2) || !aligned16(w2_packed)) return RL8_EALIGN;
  const int grid = kHidden * kHidden / kBlock;
  if (transposed)
    mlp_pack_w2t_kernel<<<grid, kBlock, 0, (hipStream_t)stream>>>(w2, w2_packed);
  else
    mlp_pack_w2_kernel<<<grid, kBlock, 0, (hipStream_t)stream>>>(w2, w2_packed);
  return launch_status();
}

template <int DIN, int NOUT, bool SAVE>
static int launch_forward_save(int grid, hipStream_t s, const float *x, int64_t m, int d_in,
                               const float *w1, const float *b1, const float4 *w2p,
                               const float *b2, const float *w3, const float *b3, int n_out,
                               float *out, float *save_h1, float *save_h2) {
  static LdsOptIn lds_attr_set_0;
  if (const int e_lds_attr_set_0 = allow_dynamic_lds(lds_attr_set_0, reinterpret_cast<const void *>(&mlp_tower_forward_kernel<DIN, NOUT, SAVE>), 160 * 1024)) return e_lds_attr_set_0;
  mlp_tower_forward_kernel<DIN, NOUT, SAVE><<<grid, kBlock, mlp_forward_lds_bytes(), s>>>(
      x, m, d_in, w1, b1, w2p, b2, w3, b3, n_out, out, save_h1, save_h2);
  return launch_status();
}

template <int DIN, int NOUT>
static int launch_forward(int grid, hipStream_t s, const float *x, int64_t m, int d_in,
                          const float *w1, const float *b1, const float4 *w2p, const float *b2,
                          const float *w3, const float *b3, int n_out, float *out, float *h1,
                          float *h2) {
  return h1 ? launch_forward_save<DIN, NOUT, true>(grid, s, x, m, d_in, w1, b1, w2p, b2, w3, b3, n_out, out, h1, h2)
            : launch_forward_save<DIN, NOUT, false>(grid, s, x, m, d_in, w1, b1, w2p, b2, w3, b3, n_out, out, h1, h2);
}

template <int DIN>
static int dispatch_forward_nout(int n_out, int grid, hipStream_t s, const float *x, int64_t m,
                                 int d_in, const float *w1, const float *b1, const float4 *w2p,
                                 const float *b2, const float *w3, const float *b3, float *out,
                                 float *h1, float *h2) {
  switch (n_out) {
    case 1: return launch_forward<DIN, 1>(grid, s, x, m, d_in, w1, b1, w2p, b2, w3, b3, n_out, out, h1, h2);
    case 2: return launch_forward<DIN, 2>(grid, s, x, m, d_in, w1, b1, w2p, b2, w3, b3, n_out, out, h1, h2);
    case 3: return launch_forward<DIN, 3>(grid, s, x, m, d_in, w1, b1, w2p, b2, w3, b3, n_out, out, h1, h2);
    default: return launch_forward<DIN, 0>(grid, s, x, m, d_in, w1, b1, w2p, b2, w3, b3, n_out, out, h1, h2);
  }
}

RL8_API int rl8_mlp_tower_forward_f32(const float *x, int64_t m, int d_in, const float *w1,
                                      const float *b1, const float *w2_packed, const float *b2,
                                      const float *w3, const float *b3, int n_out, float *out,
                                      float *save_h1, float *save_h2, void *stream) {
  if (!x || !w1 || !b1 || !w2_packed || !b2 || !w3 || !b3 || !out) return RL8_ENULL;
  if ((save_h1 == nullptr) != (save_h2 == nullptr)) return RL8_ENULL;
  if (m <= 0 || d_in <= 0 || d_in > kMaxIn || n_out <= 0 || n_out > kMaxOut) return RL8_ESIZE;
  if (!aligned16(w2_packed)) return RL8_EALIGN;
  const int64_t tiles = (m + kTileRows - 1) / kTileRows;
  static const int cap = env_int("RL8_MLP_GRID_CAP");
  const int max_grid = cap > 0 ? cap : 2 * kCUs;  // two resident workgroups per CU
  const int grid = (int)(tiles < max_grid ? tiles : max_grid);
  const float4 *w2p = reinterpret_cast<const float4 *>(w2_packed);
  hipStream_t s = (hipStream_t)stream;
  switch (d_in) {  // the built-in environments' observation widths compiled in; anything else run-time
    case 1: return dispatch_forward_nout<1>(n_out, grid, s, x, m, d_in, w1, b1, w2p, b2, w3, b3, out, save_h1, save_h2);
    case 2: return dispatch_forward_nout<2>(n_out, grid, s, x, m, d_in, w1, b1, w2p, b2, w3, b3, out, save_h1, save_h2);
    case 3: return dispatch_forward_nout<3>(n_out, grid, s, x, m, d_in, w1, b1, w2p, b2, w3, b3, out, save_h1, save_h2);
    case 5: return dispatch_forward_nout<5>(n_out, grid, s, x, m, d_in, w1, b1, w2p, b2, w3, b3, out, save_h1, save_h2);
    default: return dispatch_forward_nout<0>(n_out, grid, s, x, m, d_in, w1, b1, w2p, b2, w3, b3, out, save_h1, save_h2);
  }
}

RL8_API int64_t rl8_mlp_backward_partial_floats(int d_in, int n_out) {
  return (int64_t)kHidden * d_in + 2 * kHidden + (int64_t)n_out * kHidden + n_out;
}

RL8_API int rl8_mlp_backward_max_rows(void) { return 2 * kCUs; }

// Weight gradient of the 256x256 layer: dW2[j][i] = sum_s dZ2[s][j] * h1[s][i],
// a [256 x M] x [M x 256] product with the reduction over SAMPLES.  Every
// workgroup owns the whole 256x256 output (8 waves x 8 accumulator tiles) and a
// slice of the rows.  32-row tiles of dZ2 and h1 go HBM -> LDS with the
// direct-to-LDS buffer loads (one 1-KiB row per instruction: no registers, no
// VALU, no ds_write), double-buffered so the next tile lands during the matrix
// loop of the current one; the loop itself is MFMA + LDS reads with immediate
// offsets only.  The partial sum is left in a workspace slab that a second
// kernel adds up in slab order (bitwise reproducible; no atomics).
constexpr int kWgradThreads = 512;  // 8 waves: wave = (pair of j-tiles, quad of i-tiles)
constexpr int kWgradRows = 32;      // rows per staged tile
// LDS row stride 320 floats = 5 x 64 dwords, so the row offset fits the
// immediates of ds_read2st64_b32 (units of 64 dwords) and the matrix loop needs
// no address arithmetic; rows with bit 2 set -- the rows the upper half-wave reads
// -- are shifted by 32 floats so the two half-waves fall on disjoint banks.
constexpr int kWgradStride = 320;
constexpr int kWgradTile = kWgradRows * kWgradStride;  // floats per array per buffer
constexpr int wgrad_row_offset(int row) { return row * kWgradStride + ((row & 4) ? 32 : 0); }

// Operands of one k-group (8 rows; this lane's four are 8g + 4*kh + e), read
// with ds_read2st64_b32: two rows of one column per instruction, row offsets as
// immediates.  Issued through inline asm because the compiler would rather pair
// the columns (ds_read2_b32) and then has to add a new base per row -- one VALU
// instruction between MFMAs per row, each costing an MFMA->VALU->MFMA switch.
// Being invisible to the compiler's counters, the reads are fenced by
// wgrad_wait<N>() (s_waitcnt lgkmcnt(N), with the operands as in/out so that
// no use can be scheduled above it).
typedef __attribute__((address_space(3))) float lds_float_t;
__device__ __forceinline__ unsigned lds_address(const float *p) {
  return (unsigned)(uintptr_t)(lds_float_t *)p;
}

struct WgradFrag {
  f32x2 a[2][2], b[4][2];  // [column tile][row pair (e, e+1)]
};

template <int O0, int O1>
__device__ __forceinline__ f32x2 lds_read2st64(unsigned addr) {
  f32x2 v;
  asm volatile("ds_read2st64_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(v) : "v"(addr), "n"(O0), "n"(O1));
  return v;
}

// addr[0..1]: dZ2 columns of the two j-tiles, addr[2..5]: h1 columns of the four i-tiles.
template <int G>
__device__ __forceinline__ void wgrad_load(WgradFrag &f, const unsigned (&addr)[6]) {
  constexpr int R = 8 * G * (kWgradStride / 64);  // row offset in units of 64 dwords
  constexpr int S = kWgradStride / 64;
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    f.a[p][0] = lds_read2st64<R, R + S>(addr[p]);
    f.a[p][1] = lds_read2st64<R + 2 * S, R + 3 * S>(addr[p]);
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    f.b[t][0] = lds_read2st64<R, R + S>(addr[2 + t]);
    f.b[t][1] = lds_read2st64<R + 2 * S, R + 3 * S>(addr[2 + t]);
  }
}

template <int N>
__device__ __forceinline__ void wgrad_wait(WgradFrag &f) {
  asm volatile("s_waitcnt lgkmcnt(%12)"
               : "+v"(f.a[0][0]), "+v"(f.a[0][1]), "+v"(f.a[1][0]), "+v"(f.a[1][1]), "+v"(f.b[0][0]),
                 "+v"(f.b[0][1]), "+v"(f.b[1][0]), "+v"(f.b[1][1]), "+v"(f.b[2][0]), "+v"(f.b[2][1]),
                 "+v"(f.b[3][0]), "+v"(f.b[3][1])
               : "n"(N));
}

__device__ __forceinline__ void wgrad_mma(const WgradFrag &f, f32x16 (&acc)[2][4]) {
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int ja = 0; ja < 2; ++ja)
#pragma unroll
      for (int t = 0; t < 4; ++t)
        acc[ja][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[ja][e >> 1][e & 1], f.b[t][e >> 1][e & 1],
                                                          acc[ja][t], 0, 0, 0);
}

__global__ __launch_bounds__(kWgradThreads, 1) void mlp_wgrad_kernel(
    const float *__restrict__ dz2, int64_t z_pitch, const float *__restrict__ h1, int64_t h_pitch,
    int64_t m, float *__restrict__ slabs) {
  extern __shared__ float lds[];  // [2 buffers][dZ2 tile | h1 tile][32 rows x 320]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int jl = lane & 31, kh = lane >> 5;
  const int wj = wave >> 1, wi = wave & 1;   // j-tiles {2wj, 2wj+1}, i-tiles {4wi .. 4wi+3}
  f32x16 acc[2][4];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

  const int64_t tiles = (m + kWgradRows - 1) / kWgradRows;
  // Wave w fetches rows w, w+8, w+16, w+24 of both arrays; rows past the end of
  // the data are out of range of the descriptor and arrive as zeros.
  auto fetch = [&](int64_t tile, int buffer) {
    const int64_t r0 = tile * kWgradRows;
    const int rows = (int)((m - r0) < kWgradRows ? (m - r0) : kWgradRows);
    // row pitches in floats (256 for dense operands; 1024 for one gate of an LSTM's [rows][4][256] gradients)
    const __amdgpu_buffer_rsrc_t zr = buffer_rsrc(dz2 + r0 * z_pitch, (uint32_t)(((rows - 1) * z_pitch + kHidden) * 4));
    const __amdgpu_buffer_rsrc_t hr = buffer_rsrc(h1 + r0 * h_pitch, (uint32_t)(((rows - 1) * h_pitch + kHidden) * 4));
    float *zb = lds + buffer * 2 * kWgradTile, *hb = zb + kWgradTile;
#pragma unroll
    for (int u = 0; u < kWgradRows / 8; ++u) {
      const int row = wave + 8 * u;  // bit 2 of the row = bit 2 of the wave
      const int off = row * kWgradStride + ((wave & 4) ? 32 : 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(zr, zb + off, 16, lane * 16, row * (int)z_pitch * 4, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(hr, hb + off, 16, lane * 16, row * (int)h_pitch * 4, 0, 0);
    }
  };
  // Per-lane LDS addresses of the six operand columns, for either buffer.
  auto columns = [&](int buffer, unsigned (&addr)[6]) {
    const float *za = lds + buffer * 2 * kWgradTile + wgrad_row_offset(4) * kh + 64 * wj + jl;
    addr[0] = lds_address(za);
    addr[1] = addr[0] + 32 * 4;
#pragma unroll
    for (int t = 0; t < 4; ++t) addr[2 + t] = addr[0] + (kWgradTile - 64 * wj + 128 * wi + 32 * t) * 4;
  };
  // Pipeline.  The one barrier per tile sits BEFORE the tile's last k-group, when
  // that group's operands are already in registers: behind it the buffer is
  // free (the tile after next starts streaming into it) and the next tile has
  // landed (its first operands are read while the last 32 MFMAs run).  A
  // barrier at the tile boundary instead leaves all eight waves without
  // operands at the same moment -- ~1000 idle cycles of the matrix pipes per
  // tile, measured.
  const int64_t t0 = blockIdx.x, stride = gridDim.x;
  fetch(t0, 0);
  if (t0 + stride < tiles) fetch(t0 + stride, 1);
  if (t0 + stride < tiles)
    __builtin_amdgcn_s_waitcnt(0x0f70 | 8);  // vmcnt(8): the first tile's 8 loads are done
  else
    wait_vmcnt0();
  __syncthreads();
  unsigned addr0[6], addr1[6];
  WgradFrag fa, fb;
  columns(0, addr0);
  columns(1, addr1);
  wgrad_load<0>(fa, addr0);
  auto one_tile = [&](int64_t tile, int buffer, const unsigned (&addr)[6], const unsigned (&next_addr)[6]) {
    wgrad_load<1>(fb, addr);
    wgrad_wait<12>(fa);
    __builtin_amdgcn_sched_barrier(0);
    wgrad_mma(fa, acc);
    __builtin_amdgcn_sched_barrier(0);
    wgrad_load<2>(fa, addr);
    wgrad_wait<12>(fb);
    __builtin_amdgcn_sched_barrier(0);
    wgrad_mma(fb, acc);
    __builtin_amdgcn_sched_barrier(0);
    wgrad_load<3>(fb, addr);
    wgrad_wait<12>(fa);
    __builtin_amdgcn_sched_barrier(0);
    wgrad_mma(fa, acc);
    __builtin_amdgcn_sched_barrier(0);
    wgrad_wait<0>(fb);  // every read of this buffer is complete
    wait_vmcnt0();      // this wave's rows of the next tile have landed
    __syncthreads();
    if (tile + 2 * stride < tiles) fetch(tile + 2 * stride, buffer);
    if (tile + stride < tiles) wgrad_load<0>(fa, next_addr);
    __builtin_amdgcn_sched_barrier(0);
    wgrad_mma(fb, acc);
    __builtin_amdgcn_sched_barrier(0);
  };
  for (int64_t tile = t0; tile < tiles; tile += 2 * stride) {
    one_tile(tile, 0, addr0, addr1);
    if (tile + stride < tiles) one_tile(tile + stride, 1, addr1, addr0);
  }
  // No hand-issued LDS read is in flight here (a tile prefetches fragments only when
  // a next tile follows), but that is a fact about correlated branches; this wait
  // (once per kernel) makes it a fact about the instruction stream, which is what
  // tools/check_inflight_regs.py verifies before the epilogue reuses the registers.
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);  // (the epilogue's address arithmetic stays behind the wait)
  // Partial slab of this workgroup: slab[j][i].
  float *slab = slabs + (int64_t)blockIdx.x * kHidden * kHidden;
#pragma unroll
  for (int ja = 0; ja < 2; ++ja)
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int j = 64 * wj + 32 * ja + (r & 3) + 8 * (r >> 2) + 4 * kh;
        const int i = 128 * wi + 32 * t + jl;
        slab[j * kHidden + i] = acc[ja][t][r];
      }
}

// out[idx] (+)= sum over slabs, in slab order.
__global__ __launch_bounds__(kBlock) void mlp_wgrad_reduce_kernel(const float *__restrict__ slabs,
                                                                 int rows, float *__restrict__ out,
                                                                 int accumulate) {
  const int idx = blockIdx.x * kBlock + threadIdx.x;  // one float4 each
  if (idx >= kHidden * kHidden / 4) return;
  const float4 *p = reinterpret_cast<const float4 *>(slabs) + idx;
  float4 sum = accumulate ? reinterpret_cast<float4 *>(out)[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
  for (int r = 0; r < rows; ++r) {
    const float4 v = p[(int64_t)r * (kHidden * kHidden / 4)];
    sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
  }
  reinterpret_cast<float4 *>(out)[idx] = sum;
}

template <int DIN, int NOUT>
static int launch_backward(int grid, hipStream_t s, const float *x, const float *h1,
                           const float *h2, const float *dout, int64_t m, int d_in,
                           const float4 *w2tp, const float *w3, int n_out, float *dz2_out,
                           float *partials, int stride) {
  static LdsOptIn lds_attr_set_0;
  if (const int e_lds_attr_set_0 = allow_dynamic_lds(lds_attr_set_0, reinterpret_cast<const void *>(&mlp_tower_backward_kernel<DIN, NOUT>), 160 * 1024)) return e_lds_attr_set_0;
  mlp_tower_backward_kernel<DIN, NOUT><<<grid, kBlock, mlp_backward_lds_bytes(), s>>>(
      x, h1, h2, dout, m, d_in, w2tp, w3, n_out, dz2_out, partials, stride);
  return launch_status();
}

template <int DIN>
static int dispatch_backward_nout(int n_out, int grid, hipStream_t s, const float *x,
                                  const float *h1, const float *h2, const float *dout,
                                  int64_t m, int d_in, const float4 *w2tp, const float *w3,
                                  float *dz2_out, float *partials, int stride) {
  switch (n_out) {
    case 1: return launch_backward<DIN, 1>(grid, s, x, h1, h2, dout, m, d_in, w2tp, w3, n_out, dz2_out, partials, stride);
    case 2: return launch_backward<DIN, 2>(grid, s, x, h1, h2, dout, m, d_in, w2tp, w3, n_out, dz2_out, partials, stride);
    case 3: return launch_backward<DIN, 3>(grid, s, x, h1, h2, dout, m, d_in, w2tp, w3, n_out, dz2_out, partials, stride);
    default: return launch_backward<DIN, 0>(grid, s, x, h1, h2, dout, m, d_in, w2tp, w3, n_out, dz2_out, partials, stride);
  }
}

RL8_API int rl8_mlp_tower_backward_f32(const float *x, const float *h1, const float *h2,
                                       const float *dout, int64_t m, int d_in,
                                       const float *w2t_packed, const float *w3, int n_out,
                                       float *dz2_out, float *partials, int *partial_rows_out,
                                       void *stream) {
  if (!x || !h1 || !h2 || !dout || !w2t_packed || !w3 || !dz2_out || !partials ||
      !partial_rows_out)
    return RL8_ENULL;
  if (!aligned16(h2) || !aligned16(h1) || !aligned16(dz2_out)) return RL8_EALIGN;
  if (m <= 0 || d_in <= 0 || d_in > kMaxIn || n_out <= 0 || n_out > kMaxOut) return RL8_ESIZE;
  if (!aligned16(w2t_packed)) return RL8_EALIGN;
  const int64_t tiles = (m + kTileRows - 1) / kTileRows;
  const int grid = (int)(tiles < 2 * kCUs ? tiles : 2 * kCUs);
  *partial_rows_out = grid;
  const float4 *w2tp = reinterpret_cast<const float4 *>(w2t_packed);
  const int stride = (int)rl8_mlp_backward_partial_floats(d_in, n_out);
  hipStream_t s = (hipStream_t)stream;
  switch (d_in) {
    case 1: return dispatch_backward_nout<1>(n_out, grid, s, x, h1, h2, dout, m, d_in, w2tp, w3, dz2_out, partials, stride);
    case 2: return dispatch_backward_nout<2>(n_out, grid, s, x, h1, h2, dout, m, d_in, w2tp, w3, dz2_out, partials, stride);
    case 3: return dispatch_backward_nout<3>(n_out, grid, s, x, h1, h2, dout, m, d_in, w2tp, w3, dz2_out, partials, stride);
    case 5: return dispatch_backward_nout<5>(n_out, grid, s, x, h1, h2, dout, m, d_in, w2tp, w3, dz2_out, partials, stride);
    default: return dispatch_backward_nout<0>(n_out, grid, s, x, h1, h2, dout, m, d_in, w2tp, w3, dz2_out, partials, stride);
  }
}

RL8_API int64_t rl8_mlp_wgrad_workspace_bytes(void) {
  // one 256 x 256 slab per CU, and 256 bytes behind them (rl8_mlp_wgrad_gate_bits_f32's operand bounds)
  return (int64_t)kCUs * kHidden * kHidden * (int64_t)sizeof(float) + 256;
}

RL8_API int rl8_mlp_wgrad_strided_f32(const float *dz2, int64_t dz2_pitch, const float *h1,
                                      int64_t h1_pitch, int64_t m, float *workspace, float *dw2_out,
                                      int accumulate, void *stream) {
  if (!dz2 || !h1 || !workspace || !dw2_out) return RL8_ENULL;
  if (m <= 0 || dz2_pitch < kHidden || h1_pitch < kHidden || dz2_pitch > 65536 || h1_pitch > 65536 ||
      dz2_pitch % 4 || h1_pitch % 4)
    return RL8_ESIZE;
  if (!aligned16(dz2) || !aligned16(h1) || !aligned16(workspace) || !aligned16(dw2_out))
    return RL8_EALIGN;
  static LdsOptIn lds_attr_set_0;
  if (const int e_lds_attr_set_0 = allow_dynamic_lds(lds_attr_set_0, reinterpret_cast<const void *>(&mlp_wgrad_kernel), 160 * 1024)) return e_lds_attr_set_0;
  const int64_t tiles = (m + kWgradRows - 1) / kWgradRows;
  const int grid = (int)(tiles < kCUs ? tiles : kCUs);
  hipStream_t s = (hipStream_t)stream;
  const size_t lds_bytes = sizeof(float) * 4 * kWgradTile;
  mlp_wgrad_kernel<<<grid, kWgradThreads, lds_bytes, s>>>(dz2, dz2_pitch, h1, h1_pitch, m, workspace);
  int st = launch_status();
  if (st != RL8_OK) return st;
  mlp_wgrad_reduce_kernel<<<kHidden * kHidden / 4 / kBlock, kBlock, 0, s>>>(workspace, grid,
                                                                          dw2_out, accumulate);
  return launch_status();
}

RL8_API int rl8_mlp_wgrad_f32(const float *dz2, const float *h1, int64_t m, float *workspace,
                              float *dw2_out, int accumulate, void *stream) {
  return rl8_mlp_wgrad_strided_f32(dz2, kHidden, h1, kHidden, m, workspace, dw2_out, accumulate, stream);
}
