// How fast HBM delivers rows read in pieces (lstm_rows_kernels.hip reads 64 bytes of a row per unit chunk and comes back for
// the next 64 a chunk-time later: 2.6 TB/s).  2^21 rows of 4 KiB (the saved gates of the recurrent bench); one launch
// reads bytes [p P, (p + 1) P) of every row, the wave's lanes laid out as in that kernel: lane n (and n + 32) = row n of
// 32, two lanes together 32 bytes per instruction, P / 32 instructions per row piece, sixteen 16-byte loads in flight per
// lane.  Also the same bytes written.
//   hipcc --offload-arch=gfx950 -O3 -o piece_size_probe piece_size_probe.hip && ./piece_size_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

constexpr int64_t kRows = 1 << 21;
constexpr int kRowBytes = 4096;

template <int P, bool WRITE>
__global__ __launch_bounds__(256) void pieces(unsigned char *buf, int piece, float *out) {
  constexpr int kInstr = P / 32;                         // per row piece
  constexpr int kGroups = kInstr >= 16 ? 1 : 16 / kInstr;  // 32-row groups per iteration: sixteen loads in flight
  const int lane = threadIdx.x & 63, n = lane & 31, hh = lane >> 5;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), waves = (int64_t)gridDim.x * 4;
  float acc = 0.f;
  for (int64_t g0 = wave * kGroups; g0 < kRows / 32; g0 += waves * kGroups) {
    float4 v[kGroups][kInstr];
#pragma unroll
    for (int g = 0; g < kGroups; ++g)
#pragma unroll
      for (int i = 0; i < kInstr; ++i) {
        float4 *p = reinterpret_cast<float4 *>(buf + ((g0 + g) * 32 + n) * kRowBytes + (int64_t)piece * P + i * 32 + hh * 16);
        if (WRITE) *p = make_float4(1.f, 2.f, 3.f, (float)i);
        else v[g][i] = *p;
      }
    if (!WRITE) {
#pragma unroll
      for (int g = 0; g < kGroups; ++g)
#pragma unroll
        for (int i = 0; i < kInstr; ++i) acc += v[g][i].x + v[g][i].w;
    }
  }
  if (acc == 123.456f) out[0] = acc;
}

template <int P, bool WRITE>
static void run(unsigned char *buf, float *out, int grid) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  const int pieces_per_row = kRowBytes / P;
  pieces<P, WRITE><<<grid, 256>>>(buf, 0, out);
  hipEventRecord(a);
  int launches = 0;
  for (int p = 0; p < pieces_per_row && launches < 16; ++p, ++launches) pieces<P, WRITE><<<grid, 256>>>(buf, p, out);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0.f;
  hipEventElapsedTime(&ms, a, b);
  const double bytes = (double)kRows * P * launches;
  printf("%5d-byte pieces %s grid %5d: %7.1f us per launch  %6.2f TB/s\n", P, WRITE ? "written" : "read   ", grid,
         ms * 1000.0 / launches, bytes / (ms * 1e-3) / 1e12);
}

int main() {
  unsigned char *buf;
  float *out;
  hipMalloc(&buf, (size_t)kRows * kRowBytes);
  hipMalloc(&out, 64);
  hipMemset(buf, 1, (size_t)kRows * kRowBytes);
  for (int grid : {256, 2048}) {  // one wave per SIMD (as the kernel), eight
    run<64, false>(buf, out, grid);
    run<128, false>(buf, out, grid);
    run<256, false>(buf, out, grid);
    run<512, false>(buf, out, grid);
    run<1024, false>(buf, out, grid);
    run<64, true>(buf, out, grid);
    run<128, true>(buf, out, grid);
    run<256, true>(buf, out, grid);
    run<1024, true>(buf, out, grid);
  }
  return 0;
}
