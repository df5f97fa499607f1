// How fast does the bf16 MFMA actually issue?  N independent v_mfma_f32_32x32x16_bf16 per wave, no memory traffic:
// wall time (HIP events), shader clocks (s_memtime) and real time (s_memrealtime, 100 MHz) per wave.
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_rate_probe.hip -o /tmp/mr && /tmp/mr [waves_per_simd] [workgroups]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC, int KIND>
__global__ __launch_bounds__(256) void probe(float* out, long long* clk, int iters) {
  bf16x8 a, b;
  for (int k = 0; k < 8; ++k) { a[k] = (__bf16)(float)(threadIdx.x & 3); b[k] = (__bf16)(float)(threadIdx.x & 1); }
  f32x16 acc[NACC];
  f32x4 acc4[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 4; ++r) acc4[i][r] = 0.f;
  const long long c0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      if (KIND == 0) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
      else acc4[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc4[i], 0, 0, 0);
    }
  }
  const long long c1 = clock64(), w1 = wall_clock64();
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc4[i][0];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}

template <int NACC, int KIND>
void run(const char* name, int wgs, int iters) {
  float* out; long long* clk;
  hipMalloc(&out, (size_t)wgs * 256 * 4); hipMalloc(&clk, (size_t)wgs * 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9f;
  for (int r = 0; r < 5; ++r) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<NACC, KIND>), dim3(wgs), dim3(256), 0, 0, out, clk, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
  const double n = (double)iters * NACC;
  const double flop = n * (KIND == 0 ? 32.0 * 32 * 16 * 2 : 16.0 * 16 * 32 * 2) * 4 * wgs;
  printf("%-28s wgs %5d: %8.1f us  %7.0f TFLOP/s | per MFMA: %.2f shader clocks, %.2f ns real -> %.2f GHz, event-time per MFMA %.2f ns\n",
         name, wgs, best * 1e3, flop / best / 1e9, h[0] / n, h[1] * 10.0 / n, (h[0] / n) / (h[1] * 10.0 / n), best * 1e6 / n);
  hipFree(out); hipFree(clk);
}

int main(int argc, char** argv) {
  const int iters = 20000;
  for (int wgs : {64, 256, 512, 1024}) {
    run<4, 0>("32x32x16 bf16, 4 acc", wgs, iters);
    run<1, 0>("32x32x16 bf16, 1 acc (chain)", wgs, iters);
    run<8, 1>("16x16x32 bf16, 8 acc", wgs, iters);
  }
  return 0;
}
