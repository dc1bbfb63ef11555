// a-9 (recurrent PPO): the default recurrent models' one-layer,
// 256-wide LSTM (src/rl8/models/_recurrent.py:201-321 -> torch.nn.LSTM, batch_first)
// as fused fp32 kernels on the matrix cores, on the machinery of the towers
// (mfma_tile.hip.h): per timestep the recurrent product h_{t-1} x W_hh^T is four
// 256x256 tile products (one per gate, each against its own fragment-packed
// weight block), the input projection (d_in is tiny) and every gate non-linearity
// and cell update happen on the accumulators, and the time loop runs inside the
// kernel with h_t handed to the next step through LDS.  PyTorch's own path is a
// GEMM per operand and timestep (64 TFLOP/s here) plus a pointwise kernel per
// timestep that round-trips the 4 KiB of gates per row through HBM.
//
// Layout: a workgroup owns 32 sequences (one 32-row M-tile) at a time; wave w owns
// hidden units [64w, 64w + 64) of all four gates, so a lane holds i, f, g, o, c of
// the same (row, unit) in the same accumulator slot: row (r&3) + 8(r>>2) + 4(l>>5),
// unit 64w + 32nt + (l&31).  Gate order in the weights is torch's: i, f, g, o.
// Gates are evaluated i, g, f, o so that at most one gate's activations, the
// product i*g and the cell state are live next to the accumulators.
#include "mfma_tile.hip.h"

namespace rl8 {

constexpr int kLstmRows = 32;  // sequences per tile
constexpr int kLstmGroups = kGroups + 1;              // K = 256 hidden + 8 (inputs, bias column, zero padding)
constexpr int kLstmStride = kHidden + 9;              // LDS row pitch of the [h | x | 1 | 0..] tile, odd
constexpr int kLstmMaxIn = 7;                         // inputs + the bias column fit the extra k-group
constexpr int kLstmPackFloats = 4 * 8 * kLstmGroups * 4 * kWave;  // per direction
constexpr int kGateOrder[4] = {0, 2, 1, 3};

// Gate non-linearities on the hardware exp2 / rcp (1 ulp each; the reference's
// tolerance is 1e-5): sigmoid(x) = 1 / (1 + e^-x), tanh(x) = 1 - 2 / (e^2x + 1).
__device__ __forceinline__ float sigmoid_f(float x) {
  return __builtin_amdgcn_rcpf(1.0f + __expf(-x));
}
__device__ __forceinline__ float tanh_f(float x) {
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(__expf(2.0f * x) + 1.0f);
}

// Forward weights in fragment order, the input projection and the biases folded
// into the recurrent product as eight extra k: with the operand row
// [h_{t-1} (256) | x_t (d) | 1 | 0 ...] and
//   Wcat[j][k] = w_hh[j][k] (k < 256), w_ih[j][k - 256] (k < 256 + d),
//                b_ih[j] + b_hh[j] (k = 256 + d), 0 beyond,
// a gate's pre-activation is one tile product -- no VALU work and no registers
// for inputs or biases.  packed[(((q*8 + n)*33 + g)*4 + e)*64 + l] =
//   Wcat[256q + 32n + (l&31)][8g + 4(l>>5) + e]   (cf. rl8_mlp_pack_w2_f32).
__global__ __launch_bounds__(kBlock) void lstm_pack_kernel(const float *__restrict__ w_ih,
                                                           const float *__restrict__ w_hh,
                                                           const float *__restrict__ b_ih,
                                                           const float *__restrict__ b_hh, int d_in,
                                                           float *__restrict__ packed) {
  const int idx = blockIdx.x * kBlock + threadIdx.x;
  if (idx >= kLstmPackFloats) return;
  const int l = idx & 63, e = (idx >> 6) & 3, gn = idx >> 8;
  const int g = gn % kLstmGroups, qn = gn / kLstmGroups;  // qn = q*8 + n
  const int j = 32 * qn + (l & 31), k = 8 * g + 4 * (l >> 5) + e;
  float v = 0.0f;
  if (k < kHidden)
    v = w_hh[j * kHidden + k];
  else if (k < kHidden + d_in)
    v = w_ih[j * d_in + (k - kHidden)];
  else if (k == kHidden + d_in)
    v = b_ih[j] + b_hh[j];
  packed[idx] = v;
}

// Forward over b sequences of l steps.  SAVE: also store the post-activation gates
// ([b][l][4][256], order i, f, g, o) and the cell states ([b][l][256]) for the
// backward kernel.  LDS: one [32][265] tile: h_{t-1} (then h_t), x_t, the ones
// column = 34 KB.
template <bool SAVE>
__global__ __launch_bounds__(kBlock, 2) void lstm_forward_kernel(
    const float *__restrict__ x, int64_t b, int l, int d_in, const float *__restrict__ h0,
    const float *__restrict__ c0, const float *__restrict__ w_packed, float *__restrict__ hs,
    float *__restrict__ hn, float *__restrict__ cn, float *__restrict__ save_gates,
    float *__restrict__ save_c) {
  extern __shared__ float lds[];
  float *ht = lds;  // [32][265]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hh = lane >> 5, l31 = lane & 31;

  __amdgpu_buffer_rsrc_t wrsrc[4];
#pragma unroll
  for (int q = 0; q < 4; ++q)
    wrsrc[q] = buffer_rsrc(w_packed + q * (kLstmPackFloats / 4), kLstmPackFloats);  // bytes of one gate
  TileGemmT<1, kLstmGroups, kLstmStride> gemm(wrsrc[0], wave, lane);
  __builtin_amdgcn_s_setprio(kValuPhasePriority);
  // Accumulator slot (nt, r) of this lane is row (r&3) + 8(r>>2) + 4*hh, unit
  // 64*wave + 32*nt + l31.  Byte offsets: the lane part (4*hh rows + unit) goes in
  // the VGPR offset, the (r) row part in the scalar offset; the row pitch depends
  // on the array.
  const int unit4 = (64 * wave + l31) * 4;
  const int v_state = 4 * hh * kHidden * 4 + unit4;          // [rows][256]: c0, hn, cn
  const int v_seq = 4 * hh * l * kHidden * 4 + unit4;        // [rows][l][256]: hs, cs
  const int v_gates = 4 * hh * l * 4 * kHidden * 4 + unit4;  // [rows][l][4][256]
  // columns 256.. of the tile: zeros, then the ones column; x_t is written per step
  for (int i = tid; i < kLstmRows * 9; i += kBlock) {
    const int row = i / 9, c = i - row * 9;
    ht[row * kLstmStride + kHidden + c] = c == d_in ? 1.0f : 0.0f;
  }
  // this thread's element of the [32][d] observation tile (threads beyond 32*d idle)
  const int x_row = tid / d_in, x_col = tid - x_row * d_in;
  const bool x_owner = tid < kLstmRows * d_in;

  const int64_t tiles = (b + kLstmRows - 1) / kLstmRows;
  for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const int64_t b0 = tile * kLstmRows;
    const int rows = (int)((b - b0) < kLstmRows ? (b - b0) : kLstmRows);
    const __amdgpu_buffer_rsrc_t h0rsrc = buffer_rsrc(h0 + b0 * kHidden, rows * kHidden * 4);
    const __amdgpu_buffer_rsrc_t c0rsrc = buffer_rsrc(c0 + b0 * kHidden, rows * kHidden * 4);
    const float *xt = x + (b0 + x_row) * l * d_in + x_col;  // + t*d_in
    const bool x_valid = x_owner && x_row < rows;
    __syncthreads();  // the previous tile's last readers of ht are done
    // h0 tile -> LDS (one 1-KiB row per direct-to-LDS load; rows past the end = 0)
#pragma unroll
    for (int u = 0; u < kLstmRows / 4; ++u) {
      const int row = wave + 4 * u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(h0rsrc, ht + row * kLstmStride, 16, lane * 16, row * (kHidden * 4), 0, 0);
    }
    if (x_owner) ht[x_row * kLstmStride + kHidden + x_col] = x_valid ? xt[0] : 0.0f;
    float c[2][16], h[2][16];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        c[nt][r] = buffer_load_f32(c0rsrc, v_state + nt * 128, ((r & 3) + 8 * (r >> 2)) * (kHidden * 4));
    gemm.bp = wrsrc[kGateOrder[0]];
    gemm.prefetch();
    wait_vmcnt0();
    __syncthreads();

    for (int t = 0; t < l; ++t) {
      const float x_next = (x_valid && t + 1 < l) ? xt[(t + 1) * d_in] : 0.0f;  // lands during the products
      const int64_t row_step0 = b0 * l + t;  // tile row 0 at this step (row pitch l)
      const __amdgpu_buffer_rsrc_t hsrsrc =
          buffer_rsrc(hs + row_step0 * kHidden, ((rows - 1) * l + 1) * kHidden * 4);
      const __amdgpu_buffer_rsrc_t csrsrc =
          buffer_rsrc(SAVE ? save_c + row_step0 * kHidden : nullptr, ((rows - 1) * l + 1) * kHidden * 4);
      const __amdgpu_buffer_rsrc_t gsrsrc = buffer_rsrc(
          SAVE ? save_gates + row_step0 * 4 * kHidden : nullptr, ((rows - 1) * l + 1) * 4 * kHidden * 4);
      float p[2][16];
#pragma unroll
      for (int step = 0; step < 4; ++step) {
        const int q = kGateOrder[step];
        f32x16 acc[1][2];
        gemm.run(ht, acc);
        if (step < 3) {
          gemm.bp = wrsrc[kGateOrder[step + 1]];
          gemm.prefetch();
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float a = q == 2 ? tanh_f(acc[0][nt][r]) : sigmoid_f(acc[0][nt][r]);
            if constexpr (SAVE)
              buffer_store_f32(a, gsrsrc, v_gates + nt * 128 + q * (kHidden * 4),
                               ((r & 3) + 8 * (r >> 2)) * l * (4 * kHidden * 4));
            if (q == 0) {
              p[nt][r] = a;
            } else if (q == 2) {
              p[nt][r] *= a;
            } else if (q == 1) {
              c[nt][r] = __builtin_fmaf(a, c[nt][r], p[nt][r]);
            } else {
              h[nt][r] = a * tanh_f(c[nt][r]);
            }
          }
      }
      __syncthreads();  // every wave has read the tile for all four gates
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int sr = (r & 3) + 8 * (r >> 2);
          ht[(sr + 4 * hh) * kLstmStride + 64 * wave + 32 * nt + l31] = h[nt][r];
          buffer_store_f32(h[nt][r], hsrsrc, v_seq + nt * 128, sr * l * (kHidden * 4));
          if constexpr (SAVE) buffer_store_f32(c[nt][r], csrsrc, v_seq + nt * 128, sr * l * (kHidden * 4));
        }
      if (x_owner) ht[x_row * kLstmStride + kHidden + x_col] = x_next;
      if (t + 1 < l) {
        gemm.bp = wrsrc[kGateOrder[0]];
        gemm.prefetch();
      }
      __syncthreads();
    }
    // final states
    const __amdgpu_buffer_rsrc_t hnrsrc = buffer_rsrc(hn + b0 * kHidden, rows * kHidden * 4);
    const __amdgpu_buffer_rsrc_t cnrsrc = buffer_rsrc(cn + b0 * kHidden, rows * kHidden * 4);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int sr = (r & 3) + 8 * (r >> 2);
        buffer_store_f32(h[nt][r], hnrsrc, v_state + nt * 128, sr * (kHidden * 4));
        buffer_store_f32(c[nt][r], cnrsrc, v_state + nt * 128, sr * (kHidden * 4));
      }
  }
}

// Backward through time over b sequences of l steps ("dgrad" half):
// per step, from dL/dh_t (the heads' gradient plus what flows back from t+1) and
// the carried dL/dc_t,
//   do = dh * tanh(c_t) * o(1-o)            dc  = dh * o * (1 - tanh^2 c_t) + dc_carry
//   di = dc * g * i(1-i)     dg = dc * i * (1 - g^2)     df = dc * c_{t-1} * f(1-f)
//   dc_carry = dc * f        dh_carry = sum_q dgate_q x W_hh[q]      (four MFMA tile products)
// The pre-activation gate gradients dG ([b][l][4][256]) are stored: the weight
// gradients are products of dG with [h_{t-1} | x_t | 1] over all rows
// (rl8_mlp_wgrad_strided_f32 per gate for W_hh, lstm_input_grad_kernel for W_ih and
// the biases).  Gradients with respect to x, h0 and c0 are not produced: in PPO the
// observations and the initial states come out of the rollout buffer.
// LDS: two [32][257] tiles (gate q+1's gradients are written while slower waves
// still multiply gate q's): 66 KB.
struct LstmBackwardStep {
  __amdgpu_buffer_rsrc_t dg;  // this step's dG rows
  int v_gates, l;
};

template <int SLOT, bool FIRST>
__device__ __forceinline__ void lstm_emit_gate(int q, const float (&dgq)[2][16], const LstmBackwardStep &st,
                                               TileGemmT<1> &gemm, const __amdgpu_buffer_rsrc_t (&wrsrc)[4],
                                               float *lds, int wave, int hh, int l31, f32x16 (&acc)[1][2]) {
  float *tile = lds + SLOT * kLstmRows * kLdsStride;
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int sr = (r & 3) + 8 * (r >> 2);
      buffer_store_f32(dgq[nt][r], st.dg, st.v_gates + nt * 128 + q * (kHidden * 4), sr * st.l * (4 * kHidden * 4));
      tile[(sr + 4 * hh) * kLdsStride + 64 * wave + 32 * nt + l31] = dgq[nt][r];
    }
  gemm.bp = wrsrc[q];
  gemm.prefetch();
  __syncthreads();  // the tile is complete (its previous readers passed the last barrier before writing here)
  gemm.run<!FIRST>(tile, acc);
}

__global__ __launch_bounds__(kBlock, 2) void lstm_backward_kernel(
    int64_t b, int l, const float *__restrict__ c0, const float *__restrict__ gates,
    const float *__restrict__ cs, const float *__restrict__ dhs,
    const float *__restrict__ whht_packed, float *__restrict__ dgates) {
  extern __shared__ float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hh = lane >> 5, l31 = lane & 31;

  __amdgpu_buffer_rsrc_t wrsrc[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) wrsrc[q] = buffer_rsrc(whht_packed + q * kHidden * kHidden, kHidden * kHidden * 4);
  TileGemmT<1> gemm(wrsrc[0], wave, lane);
  __builtin_amdgcn_s_setprio(kValuPhasePriority);
  const int unit4 = (64 * wave + l31) * 4;
  const int v_state = 4 * hh * kHidden * 4 + unit4;
  const int v_seq = 4 * hh * l * kHidden * 4 + unit4;
  const int v_gates = 4 * hh * l * 4 * kHidden * 4 + unit4;

  const int64_t tiles = (b + kLstmRows - 1) / kLstmRows;
  for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const int64_t b0 = tile * kLstmRows;
    const int rows = (int)((b - b0) < kLstmRows ? (b - b0) : kLstmRows);
    const __amdgpu_buffer_rsrc_t c0rsrc = buffer_rsrc(c0 + b0 * kHidden, rows * kHidden * 4);
    // Software pipeline over the steps: every group of loads is issued ahead of a
    // tile product and consumed behind it -- {i, g} before gate o's product, {f,
    // c_{t-1}} before gate g's, the next step's {o, dh} before gate f's; c_{t-1}
    // becomes the next step's c_t without a reload.
    float dh_carry[2][16], dc_carry[2][16], o[2][16], dh_out[2][16], c_t[2][16];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) dh_carry[nt][r] = dc_carry[nt][r] = 0.0f;
    auto seq_rsrc = [&](const float *base, int t, int per_row) {
      return buffer_rsrc(base + (b0 * l + t) * (int64_t)per_row, (uint32_t)(((rows - 1) * l + 1) * per_row * 4));
    };
    auto load_gate = [&](const __amdgpu_buffer_rsrc_t &gs, int q, float (&dst)[2][16]) {
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          dst[nt][r] = buffer_load_f32(gs, v_gates + nt * 128 + q * (kHidden * 4),
                                       ((r & 3) + 8 * (r >> 2)) * l * (4 * kHidden * 4));
    };
    auto load_seq = [&](const __amdgpu_buffer_rsrc_t &rs, float (&dst)[2][16]) {
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          dst[nt][r] = buffer_load_f32(rs, v_seq + nt * 128, ((r & 3) + 8 * (r >> 2)) * l * (kHidden * 4));
    };
    {
      const int t = l - 1;
      load_gate(seq_rsrc(gates, t, 4 * kHidden), 3, o);
      load_seq(seq_rsrc(dhs, t, kHidden), dh_out);
      load_seq(seq_rsrc(cs, t, kHidden), c_t);
    }

    for (int t = l - 1; t >= 0; --t) {
      const __amdgpu_buffer_rsrc_t gsrsrc = seq_rsrc(gates, t, 4 * kHidden);
      const LstmBackwardStep st = {seq_rsrc(dgates, t, 4 * kHidden), v_gates, l};
      f32x16 acc[1][2];  // dL/dh_{t-1} through the recurrent weights
      float dc[2][16], gi[2][16], gg[2][16], dgq[2][16];
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float dh = dh_out[nt][r] + dh_carry[nt][r];
          const float tc = tanh_f(c_t[nt][r]);
          dgq[nt][r] = dh * tc * (o[nt][r] * (1.0f - o[nt][r]));
          dc[nt][r] = __builtin_fmaf(dh * o[nt][r], 1.0f - tc * tc, dc_carry[nt][r]);
        }
      load_gate(gsrsrc, 0, gi);
      load_gate(gsrsrc, 2, gg);
      lstm_emit_gate<0, true>(3, dgq, st, gemm, wrsrc, lds, wave, hh, l31, acc);
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) dgq[nt][r] = dc[nt][r] * gg[nt][r] * (gi[nt][r] * (1.0f - gi[nt][r]));
      lstm_emit_gate<1, false>(0, dgq, st, gemm, wrsrc, lds, wave, hh, l31, acc);
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) dgq[nt][r] = dc[nt][r] * gi[nt][r] * (1.0f - gg[nt][r] * gg[nt][r]);
      // f and c_{t-1} (the previous step's saved cell state, or c0) into the registers i, g just left
      float (&f)[2][16] = gi;
      float (&cp)[2][16] = gg;
      load_gate(gsrsrc, 1, f);
      if (t > 0) {
        load_seq(seq_rsrc(cs, t - 1, kHidden), cp);
      } else {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            cp[nt][r] = buffer_load_f32(c0rsrc, v_state + nt * 128, ((r & 3) + 8 * (r >> 2)) * (kHidden * 4));
      }
      lstm_emit_gate<0, false>(2, dgq, st, gemm, wrsrc, lds, wave, hh, l31, acc);
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          dgq[nt][r] = dc[nt][r] * cp[nt][r] * (f[nt][r] * (1.0f - f[nt][r]));
          dc_carry[nt][r] = dc[nt][r] * f[nt][r];
          c_t[nt][r] = cp[nt][r];  // the next (earlier) step's c_t
        }
      if (t > 0) {
        load_gate(seq_rsrc(gates, t - 1, 4 * kHidden), 3, o);
        load_seq(seq_rsrc(dhs, t - 1, kHidden), dh_out);
      }
      lstm_emit_gate<1, false>(1, dgq, st, gemm, wrsrc, lds, wave, hh, l31, acc);
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) dh_carry[nt][r] = acc[0][nt][r];
    }
  }
}

// Gradients of the input weights and the biases: column sums over all M = B*L
// rows of dG[row][j] * x[row][c] and of dG[row][j] (j < 1024).  Memory-bound (one
// more read of dG); thread = column, a workgroup walks a slice of the rows, and
// leaves one partial row [dW_ih (1024*d) | db (1024)] summed by the host side in
// workgroup order.
template <int DIN>
__global__ __launch_bounds__(kBlock) void lstm_input_grad_kernel(const float *__restrict__ x,
                                                                 const float *__restrict__ dgates,
                                                                 int64_t m, float *__restrict__ partials) {
  const int j = blockIdx.y * kBlock + threadIdx.x;  // column of dG (gridDim.y = 4)
  float db = 0.0f, dw[DIN];
#pragma unroll
  for (int c = 0; c < DIN; ++c) dw[c] = 0.0f;
  const int64_t rows_per_block = (m + gridDim.x - 1) / gridDim.x;
  const int64_t r_begin = blockIdx.x * rows_per_block;
  const int64_t r_end = r_begin + rows_per_block < m ? r_begin + rows_per_block : m;
#pragma unroll 4
  for (int64_t r = r_begin; r < r_end; ++r) {
    const float g = dgates[r * (4 * kHidden) + j];
    db += g;
#pragma unroll
    for (int c = 0; c < DIN; ++c) dw[c] = __builtin_fmaf(g, x[r * DIN + c], dw[c]);
  }
  float *row = partials + (int64_t)blockIdx.x * (4 * kHidden * (DIN + 1));
  row[4 * kHidden * DIN + j] = db;
#pragma unroll
  for (int c = 0; c < DIN; ++c) row[j * DIN + c] = dw[c];
}

inline size_t lstm_lds_bytes() { return sizeof(float) * kLstmRows * kLstmStride; }

}  // namespace rl8

using namespace rl8;

RL8_API int rl8_lstm_supports(int d_in) { return d_in >= 1 && d_in <= kLstmMaxIn; }

RL8_API int64_t rl8_lstm_pack_floats(void) { return kLstmPackFloats; }

RL8_API int rl8_lstm_pack_f32(const float *w_ih, const float *w_hh, const float *b_ih,
                              const float *b_hh, int d_in, float *packed, void *stream) {
  if (!w_ih || !w_hh || !b_ih || !b_hh || !packed) return RL8_ENULL;
  if (!rl8_lstm_supports(d_in)) return RL8_ESIZE;
  if (!aligned16(packed)) return RL8_EALIGN;
  lstm_pack_kernel<<<(kLstmPackFloats + kBlock - 1) / kBlock, kBlock, 0, (hipStream_t)stream>>>(
      w_ih, w_hh, b_ih, b_hh, d_in, packed);
  return launch_status();
}

RL8_API int rl8_lstm_forward_f32(const float *x, int64_t b, int l, int d_in, const float *h0,
                                 const float *c0, const float *w_packed, float *hs, float *hn,
                                 float *cn, float *save_gates, float *save_c, void *stream) {
  if (!x || !h0 || !c0 || !w_packed || !hs || !hn || !cn) return RL8_ENULL;
  if ((save_gates == nullptr) != (save_c == nullptr)) return RL8_ENULL;
  if (b <= 0 || l <= 0) return RL8_ESIZE;
  if (!rl8_lstm_supports(d_in)) return RL8_ESIZE;
  // buffer descriptors address one tile's rows with 32-bit offsets
  if ((int64_t)kLstmRows * l * 4 * kHidden * 4 >= (int64_t)1 << 31) return RL8_ESIZE;
  if (!aligned16(w_packed) || !aligned16(h0)) return RL8_EALIGN;
  const int64_t tiles = (b + kLstmRows - 1) / kLstmRows;
  const int grid = (int)(tiles < 2 * kCUs ? tiles : 2 * kCUs);
  hipStream_t s = (hipStream_t)stream;
  if (save_gates)
    lstm_forward_kernel<true><<<grid, kBlock, lstm_lds_bytes(), s>>>(x, b, l, d_in, h0, c0, w_packed, hs, hn, cn,
                                                                   save_gates, save_c);
  else
    lstm_forward_kernel<false><<<grid, kBlock, lstm_lds_bytes(), s>>>(x, b, l, d_in, h0, c0, w_packed, hs, hn, cn,
                                                                    save_gates, save_c);
  return launch_status();
}

RL8_API int64_t rl8_lstm_backward_partial_floats(int d_in) { return (int64_t)4 * kHidden * (d_in + 1); }

RL8_API int rl8_lstm_backward_max_rows(void) { return 4 * kCUs; }

template <int DIN>
static int launch_lstm_input_grad(int rows, hipStream_t s, const float *x, const float *dgates,
                                  int64_t m, float *partials) {
  lstm_input_grad_kernel<DIN><<<dim3(rows, 4), kBlock, 0, s>>>(x, dgates, m, partials);
  return launch_status();
}

RL8_API int rl8_lstm_backward_f32(const float *x, int64_t b, int l, int d_in, const float *c0,
                                  const float *gates, const float *cs, const float *dhs,
                                  const float *whht_packed, float *dgates, float *partials,
                                  int *partial_rows_out, void *stream) {
  // partials == NULL: the data-gradient half only -- the caller forms dW_ih / db elsewhere
  // (rl8_mlp_wgrad_split_strided_f32 leaves them as column sums of the dG it reads anyway)
  if (!x || !c0 || !gates || !cs || !dhs || !whht_packed || !dgates || (partials && !partial_rows_out))
    return RL8_ENULL;
  if (b <= 0 || l <= 0) return RL8_ESIZE;
  if (!rl8_lstm_supports(d_in)) return RL8_ESIZE;
  if ((int64_t)kLstmRows * l * 4 * kHidden * 4 >= (int64_t)1 << 31) return RL8_ESIZE;
  if (!aligned16(whht_packed)) return RL8_EALIGN;
  const int64_t tiles = (b + kLstmRows - 1) / kLstmRows;
  const int grid = (int)(tiles < 2 * kCUs ? tiles : 2 * kCUs);
  hipStream_t s = (hipStream_t)stream;
  lstm_backward_kernel<<<grid, kBlock, 2 * sizeof(float) * kLstmRows * kLdsStride, s>>>(
      b, l, c0, gates, cs, dhs, whht_packed, dgates);
  int st = launch_status();
  if (st != RL8_OK || !partials) return st;
  const int64_t m = b * l;
  const int rows = (int)(m < 4 * kCUs ? m : 4 * kCUs);
  *partial_rows_out = rows;
  switch (d_in) {
    case 1: return launch_lstm_input_grad<1>(rows, s, x, dgates, m, partials);
    case 2: return launch_lstm_input_grad<2>(rows, s, x, dgates, m, partials);
    case 3: return launch_lstm_input_grad<3>(rows, s, x, dgates, m, partials);
    case 4: return launch_lstm_input_grad<4>(rows, s, x, dgates, m, partials);
    case 5: return launch_lstm_input_grad<5>(rows, s, x, dgates, m, partials);
    case 6: return launch_lstm_input_grad<6>(rows, s, x, dgates, m, partials);
    default: return launch_lstm_input_grad<7>(rows, s, x, dgates, m, partials);
  }
}

// ---------------------------------------------------------------------------
// The recurrent models' output heads: a few Linear(256, n) layers on the LSTM's
// outputs (src/rl8/models/_recurrent.py:230-236, 287-292), evaluated together:
// out [M][n] = h [M][256] x W^T + b with n <= 8.  A library GEMM with N = 1..3 runs
// at a small fraction of HBM speed here (3.5 ms per head and direction at M = 2^21);
// these are single passes over h.  Memory-bound: 1 KiB read per row forward,
// 1 KiB read + 1 KiB written backward.
// ---------------------------------------------------------------------------
namespace rl8 {

constexpr int kHeadsMaxOut = 8;

// Forward: 4 lanes per row (a quarter of the 256 inputs each, 16-byte loads), the
// weights broadcast from LDS, two shuffles.  The NOUT outputs may belong to two layers
// (the first n_a to {w, bias, out}, the rest to {w_b, bias_b, out_b}; n_a = NOUT: one
// layer): the rollout's logits and value heads are one pass over h_t, one launch.
struct HeadsForwardArgs {
  const float *w, *bias;
  float *out;
  int n_a;
  const float *w_b, *bias_b;
  float *out_b;
};

template <int NOUT>
__global__ __launch_bounds__(kBlock) void linear_heads_forward_kernel(const float4 *__restrict__ h, int64_t m,
                                                                      HeadsForwardArgs a) {
  __shared__ float4 ws[NOUT * kHidden / 4];
  const int n_a = a.n_a, n_b = NOUT - n_a;
  for (int i = threadIdx.x; i < NOUT * kHidden / 4; i += kBlock)
    ws[i] = i < n_a * (kHidden / 4) ? reinterpret_cast<const float4 *>(a.w)[i]
                                    : reinterpret_cast<const float4 *>(a.w_b)[i - n_a * (kHidden / 4)];
  __syncthreads();
  const int q4 = threadIdx.x & 3;
  const int64_t stride = (int64_t)gridDim.x * (kBlock / 4);
  for (int64_t row = (int64_t)blockIdx.x * (kBlock / 4) + (threadIdx.x >> 2); row < m; row += stride) {
    float o[NOUT];
#pragma unroll
    for (int q = 0; q < NOUT; ++q) o[q] = 0.0f;
    const float4 *hr = h + row * (kHidden / 4) + q4 * 16;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float4 v = hr[i];
#pragma unroll
      for (int q = 0; q < NOUT; ++q) {
        const float4 wv = ws[q * (kHidden / 4) + q4 * 16 + i];
        o[q] = __builtin_fmaf(v.x, wv.x, o[q]);
        o[q] = __builtin_fmaf(v.y, wv.y, o[q]);
        o[q] = __builtin_fmaf(v.z, wv.z, o[q]);
        o[q] = __builtin_fmaf(v.w, wv.w, o[q]);
      }
    }
#pragma unroll
    for (int q = 0; q < NOUT; ++q) {
      float v = o[q];
      v += __shfl_xor(v, 1, kWave);
      v += __shfl_xor(v, 2, kWave);
      if (q4 == 0) {
        if (q < n_a) a.out[row * n_a + q] = v + a.bias[q];
        else a.out_b[row * n_b + (q - n_a)] = v + a.bias_b[q - n_a];
      }
    }
  }
}

// Backward: thread = input unit j; a workgroup walks a contiguous slice of the rows:
// dh[s][j] = sum_q dout[s][q] w[q][j]; dW[q][j] += dout[s][q] h[s][j]; db[q] += dout[s][q].
// One partial row per workgroup: [dW (n*256) | db (n)], summed by the host side in
// workgroup order.
template <int NOUT>
__global__ __launch_bounds__(kBlock) void linear_heads_backward_kernel(const float *__restrict__ h,
                                                                       const float *__restrict__ dout,
                                                                       int64_t m,
                                                                       const float *__restrict__ w,
                                                                       float *__restrict__ dh,
                                                                       float *__restrict__ partials) {
  const int j = threadIdx.x;
  float wr[NOUT], dw[NOUT], db[NOUT];
#pragma unroll
  for (int q = 0; q < NOUT; ++q) {
    wr[q] = w[q * kHidden + j];
    dw[q] = db[q] = 0.0f;
  }
  const int64_t per_block = (m + gridDim.x - 1) / gridDim.x;
  const int64_t r0 = blockIdx.x * per_block;
  const int64_t r1 = r0 + per_block < m ? r0 + per_block : m;
#pragma unroll 4
  for (int64_t s = r0; s < r1; ++s) {
    const float hv = h[s * kHidden + j];
    float g = 0.0f;
#pragma unroll
    for (int q = 0; q < NOUT; ++q) {
      const float d = dout[s * NOUT + q];
      g = __builtin_fmaf(d, wr[q], g);
      dw[q] = __builtin_fmaf(d, hv, dw[q]);
      db[q] += d;
    }
    if (dh) dh[s * kHidden + j] = g;  // (null: parameter gradients only -- the caller forms dh where it is used)
  }
  float *row = partials + (int64_t)blockIdx.x * (NOUT * kHidden + NOUT);
#pragma unroll
  for (int q = 0; q < NOUT; ++q) {
    row[q * kHidden + j] = dw[q];
    if (j == 0) row[NOUT * kHidden + q] = db[q];
  }
}

}  // namespace rl8

RL8_API int rl8_linear_heads_max_rows(void) { return kMaxGrid; }

static int heads_forward(const float *h, int64_t m, const rl8::HeadsForwardArgs &a, int n_out, void *stream) {
  const int grid = grid_for(m, kBlock / 4);
  hipStream_t s = (hipStream_t)stream;
  const float4 *h4 = reinterpret_cast<const float4 *>(h);
  switch (n_out) {
    case 1: linear_heads_forward_kernel<1><<<grid, kBlock, 0, s>>>(h4, m, a); break;
    case 2: linear_heads_forward_kernel<2><<<grid, kBlock, 0, s>>>(h4, m, a); break;
    case 3: linear_heads_forward_kernel<3><<<grid, kBlock, 0, s>>>(h4, m, a); break;
    case 4: linear_heads_forward_kernel<4><<<grid, kBlock, 0, s>>>(h4, m, a); break;
    case 5: linear_heads_forward_kernel<5><<<grid, kBlock, 0, s>>>(h4, m, a); break;
    case 6: linear_heads_forward_kernel<6><<<grid, kBlock, 0, s>>>(h4, m, a); break;
    case 7: linear_heads_forward_kernel<7><<<grid, kBlock, 0, s>>>(h4, m, a); break;
    default: linear_heads_forward_kernel<8><<<grid, kBlock, 0, s>>>(h4, m, a); break;
  }
  return launch_status();
}

RL8_API int rl8_linear_heads_forward_f32(const float *h, int64_t m, const float *w, const float *b,
                                         int n_out, float *out, void *stream) {
  if (!h || !w || !b || !out) return RL8_ENULL;
  if (m <= 0 || n_out <= 0 || n_out > kHeadsMaxOut) return RL8_ESIZE;
  if (!aligned16(h) || !aligned16(w)) return RL8_EALIGN;
  return heads_forward(h, m, HeadsForwardArgs{w, b, out, n_out, nullptr, nullptr, nullptr}, n_out, stream);
}

// Two layers on the same rows in one pass: out_a [M][n_a] = h x w_a^T + b_a, out_b [M][n_b] = h x w_b^T + b_b.
RL8_API int rl8_linear_heads_forward_pair_f32(const float *h, int64_t m, const float *w_a, const float *b_a, int n_a,
                                              float *out_a, const float *w_b, const float *b_b, int n_b, float *out_b,
                                              void *stream) {
  if (!h || !w_a || !b_a || !out_a || !w_b || !b_b || !out_b) return RL8_ENULL;
  if (m <= 0 || n_a <= 0 || n_b <= 0 || n_a + n_b > kHeadsMaxOut) return RL8_ESIZE;
  if (!aligned16(h) || !aligned16(w_a) || !aligned16(w_b)) return RL8_EALIGN;
  return heads_forward(h, m, HeadsForwardArgs{w_a, b_a, out_a, n_a, w_b, b_b, out_b}, n_a + n_b, stream);
}

RL8_API int rl8_linear_heads_backward_f32(const float *h, const float *dout, int64_t m,
                                          const float *w, int n_out, float *dh_out, float *partials,
                                          int *partial_rows_out, void *stream) {
  if (!h || !dout || !w || !partials || !partial_rows_out) return RL8_ENULL;  // (dh_out may be null)
  if (m <= 0 || n_out <= 0 || n_out > kHeadsMaxOut) return RL8_ESIZE;
  const int grid = (int)(m < kMaxGrid ? m : kMaxGrid);
  *partial_rows_out = grid;
  hipStream_t s = (hipStream_t)stream;
  switch (n_out) {
    case 1: linear_heads_backward_kernel<1><<<grid, kBlock, 0, s>>>(h, dout, m, w, dh_out, partials); break;
    case 2: linear_heads_backward_kernel<2><<<grid, kBlock, 0, s>>>(h, dout, m, w, dh_out, partials); break;
    case 3: linear_heads_backward_kernel<3><<<grid, kBlock, 0, s>>>(h, dout, m, w, dh_out, partials); break;
    case 4: linear_heads_backward_kernel<4><<<grid, kBlock, 0, s>>>(h, dout, m, w, dh_out, partials); break;
    case 5: linear_heads_backward_kernel<5><<<grid, kBlock, 0, s>>>(h, dout, m, w, dh_out, partials); break;
    case 6: linear_heads_backward_kernel<6><<<grid, kBlock, 0, s>>>(h, dout, m, w, dh_out, partials); break;
    case 7: linear_heads_backward_kernel<7><<<grid, kBlock, 0, s>>>(h, dout, m, w, dh_out, partials); break;
    default: linear_heads_backward_kernel<8><<<grid, kBlock, 0, s>>>(h, dout, m, w, dh_out, partials); break;
  }
  return launch_status();
}
