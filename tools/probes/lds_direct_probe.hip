// Hardware probe (kernel-tuning aid): semantics of the direct-to-LDS buffer loads
// on gfx950 -- where lane l's bytes land for the dword and dwordx4 forms, and
// whether the LDS base (M0) needs 16-byte alignment for dwordx4.
//   hipcc --offload-arch=gfx950 -O3 -o lds_direct_probe lds_direct_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int STRIDE, int BYTES>
__global__ __launch_bounds__(256) void probe(const float *src, float *dst) {
  extern __shared__ float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < 64 * STRIDE; i += 256) lds[i] = -1.0f;
  __syncthreads();
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, 64 * 256 * 4, 0x00020000);
  if constexpr (BYTES == 16) {
    for (int row = wave; row < 64; row += 4)  // one full row (1 KiB) per instruction
      __builtin_amdgcn_raw_ptr_buffer_load_lds(r, lds + row * STRIDE, 16, lane * 16, row * 1024, 0, 0);
  } else {
    for (int row = wave; row < 64; row += 4)
      for (int c = 0; c < 4; ++c)  // 64 floats per instruction
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, lds + row * STRIDE + 64 * c, 4, lane * 4, row * 1024 + c * 256, 0, 0);
  }
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  for (int i = tid; i < 64 * STRIDE; i += 256) dst[i] = lds[i];
}

template <int STRIDE, int BYTES>
void run() {
  std::vector<float> h(64 * 256), out(64 * STRIDE);
  for (int i = 0; i < 64 * 256; ++i) h[i] = (float)i;
  float *src, *dst;
  hipMalloc(&src, h.size() * 4);
  hipMalloc(&dst, out.size() * 4);
  hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipFuncSetAttribute(reinterpret_cast<const void *>(&probe<STRIDE, BYTES>),
                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  probe<STRIDE, BYTES><<<1, 256, 64 * STRIDE * 4>>>(src, dst);
  hipError_t e = hipDeviceSynchronize();
  hipMemcpy(out.data(), dst, out.size() * 4, hipMemcpyDeviceToHost);
  int bad = 0, firstbad = -1;
  for (int s = 0; s < 64; ++s)
    for (int c = 0; c < 256; ++c)
      if (out[s * STRIDE + c] != h[s * 256 + c]) {
        if (firstbad < 0) firstbad = s * 256 + c;
        ++bad;
      }
  printf("stride %3d bytes %2d: %s, mismatches %d (first at %d: got %g)  [row0: %g %g %g %g %g | row1: %g %g]\n", STRIDE,
         BYTES, hipGetErrorString(e), bad, firstbad, firstbad >= 0 ? out[(firstbad / 256) * STRIDE + firstbad % 256] : 0.f,
         out[0], out[1], out[2], out[3], out[4], out[STRIDE], out[STRIDE + 1]);
  hipFree(src);
  hipFree(dst);
}

int main() {
  run<257, 4>();
  run<260, 4>();
  run<257, 16>();
  run<258, 16>();
  run<260, 16>();
  run<288, 16>();
  return 0;
}
