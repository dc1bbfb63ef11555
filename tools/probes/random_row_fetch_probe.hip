// Calibration probe for rocprofv3's FETCH_SIZE on random row reads (MI355X_MICROARCH.md, HBM: "other access widths are
// uncalibrated: calibrate on a known byte count in your own access pattern before trusting an absolute").  Three kernels
// over a 1 GiB buffer (far beyond the 256 MiB Infinity Cache), each touching every byte it reads exactly once:
//   stream_read      2^26 x 16 B coalesced                      (the pattern the guide's x2 correction was derived on)
//   random_rows<32>  2^22 random 32-byte rows, one per lane      (rl8_gather_packed's access: 134 MB useful, 268 MB of 64-B sectors)
//   random_rows<64>  2^22 random 64-byte rows, one per lane
//   cd /tmp && rocprofv3 --pmc FETCH_SIZE --kernel-trace -d out -o p --output-format csv -- ./random_row_fetch_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

__global__ void stream_read(const float4 *src, int64_t n, float *out) {
  float acc = 0.0f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float4 v = src[i];
    acc += v.x + v.y + v.z + v.w;
  }
  if (acc == 123.456f) out[0] = acc;
}

template <int BYTES>
__global__ void random_rows(const unsigned char *src, int64_t rows, uint32_t mult, float *out) {
  float acc = 0.0f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < rows; i += (int64_t)gridDim.x * blockDim.x) {
    // a permutation of [0, 2^k): odd multiplier mod 2^k -- every row exactly once, neighbours far apart
    const uint64_t total = (uint64_t)(1u << 30) / BYTES;
    const uint64_t r = ((uint64_t)i * mult) & (total - 1);
    const float4 *row = reinterpret_cast<const float4 *>(src + r * BYTES);
#pragma unroll
    for (int v = 0; v < BYTES / 16; ++v) {
      const float4 t = row[v];
      acc += t.x + t.y + t.z + t.w;
    }
  }
  if (acc == 123.456f) out[0] = acc;
}

int main() {
  unsigned char *buf;
  float *out;
  hipMalloc(&buf, (size_t)1 << 30);
  hipMalloc(&out, 64);
  hipMemset(buf, 1, (size_t)1 << 30);
  const int64_t rows = 1 << 22;
  for (int rep = 0; rep < 3; ++rep) {
    stream_read<<<2048, 256>>>(reinterpret_cast<const float4 *>(buf), (int64_t)1 << 26, out);
    random_rows<32><<<2048, 256>>>(buf, rows, 2654435761u | 1u, out);
    random_rows<64><<<2048, 256>>>(buf, rows, 2654435761u | 1u, out);
  }
  hipDeviceSynchronize();
  printf("stream_read: %lld bytes; random_rows<32>: %lld useful bytes (%lld in 64-B sectors); random_rows<64>: %lld bytes\n",
         (long long)1 << 30, (long long)rows * 32, (long long)rows * 64, (long long)rows * 64);
  return 0;
}
