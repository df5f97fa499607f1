// Micro-benchmark: do VALU instructions of one wave run under the MFMAs of ANOTHER wave on the same SIMD?
// Workgroup = 512 threads = 2 waves per SIMD; waves 0-3 run a v_mfma_f32_32x32x2_f32 loop, waves 4-7 a v_fma_f32 loop.
// Prints the time of MFMA alone, VALU alone, and both together (sum => serialised, max => co-executed).
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_valu_overlap.hip -o /tmp/mvo && /tmp/mvo
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int CHAINS>
__global__ __launch_bounds__(512) void k(float* out, int mfma_iters, int valu_iters) {
  const int wave = threadIdx.x >> 6;
  float x = threadIdx.x * 1e-3f + 1.0f, y = 0.5f + blockIdx.x * 1e-4f;
  float s = 0;
  if (wave < 4) {
    f32x16 a[4] = {};
    for (int i = 0; i < mfma_iters; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int c = 0; c < 4; ++c) a[c % CHAINS] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a[c % CHAINS], 0, 0, 0);
    }
    for (int c = 0; c < 4; ++c) for (int r = 0; r < 16; ++r) s += a[c][r];
  } else {
    float v[8];
    for (int q = 0; q < 8; ++q) v[q] = x + q;
    for (int i = 0; i < valu_iters; ++i) {
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = __builtin_fmaf(v[q], y, x);
    }
    for (int q = 0; q < 8; ++q) s += v[q];
  }
  out[blockIdx.x * 512 + threadIdx.x] = s;
}
template <int CHAINS>
float run(float* out, int mi, int vi) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<CHAINS>, dim3(256), dim3(512), 0, 0, out, mi, vi);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<CHAINS>, dim3(256), dim3(512), 0, 0, out, mi, vi);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms;
}
int main() {
  float* out; hipMalloc(&out, 256 * 512 * 4);
  const int mi = 4000, vi = 4000;   // 64000 MFMAs x 64 cycles = 4.1 M cycles; 256000 FMAs x 4 cycles = 1.0 M cycles
  printf("independent chains (4 accumulators):  mfma %.3f ms  valu %.3f ms  both %.3f ms\n", run<4>(out, mi, 0), run<4>(out, 0, 4 * vi), run<4>(out, mi, 4 * vi));
  printf("one dependent chain (1 accumulator):  mfma %.3f ms  valu %.3f ms  both %.3f ms\n", run<1>(out, mi, 0), run<1>(out, 0, 4 * vi), run<1>(out, mi, 4 * vi));
  return 0;
}
