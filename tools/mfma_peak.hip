// Micro-benchmark: sustained v_mfma_f32_32x32x2_f32 rate with operands in registers (no memory), for
// 1/2/4 waves per SIMD; prints TFLOP/s and the effective clock (s_memtime / s_memrealtime).
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void k(float* out, int iters, unsigned long long* clk) {
  f32x16 a0 = {}, a1 = {}, a2 = {}, a3 = {};
  float x = threadIdx.x * 1e-3f + 1.0f, y = 0.5f + blockIdx.x * 1e-4f;
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, a3, 0, 0, 0);
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0;
  for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}
int main() {
  float* out; unsigned long long* clk;
  hipMalloc(&out, 256 * 4096 * 4); hipMalloc(&clk, 16);
  for (int bpc = 1; bpc <= 4; bpc *= 2) {
    const int blocks = 256 * bpc, iters = 20000 / bpc;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, 100, clk);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, iters, clk);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
      const double flop = (double)blocks * 4 * iters * 16 * 4096.0;
      printf("waves/SIMD=%d  %.2f ms  %.1f TFLOP/s  clock=%.3f GHz (cycles/mfma/SIMD=%.1f)\n", bpc, ms, flop / ms / 1e9,
             (double)h[0] / h[1] * 0.1, (double)h[0] / (iters * 16.0 * bpc));
    }
  }
  return 0;
}
