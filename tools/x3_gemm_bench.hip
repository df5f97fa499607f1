// Stand-alone check + timing of the bf16x3-split GEMM core (csrc/mfma_x3.h) against fp64 on the host.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/x3_gemm_bench.hip -o tools/scratch/x3_gemm_bench
//   x3_gemm_bench [batch M N K [iters]]
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
#include <random>

#include "../strotss-tensorflow_amd/csrc/mfma_x3.h"

#ifndef SYMM_SKIP
#define SYMM_SKIP 0
#endif
struct EpiStore : X3NoPrefetch<EpiStore> {
  using X3NoPrefetch<EpiStore>::apply;
  static constexpr bool SYMM = SYMM_SKIP != 0;      // SYMM_SKIP: tiles below the diagonal return at once (timing experiment)
  static constexpr int symm = 1;
  __device__ __forceinline__ float value(int, int, float v) const { return v; }
  __device__ __forceinline__ void mirror(int r, int c, float v) const { C[(size_t)r * ldc + c] = v; }
  float* C; int ldc; int M, N; long long strideC;
  __device__ __forceinline__ void set_batch(int z) { C += (long long)z * strideC; }
  __device__ __forceinline__ float apply(int r, int c, float v) const {
    if (r < M && c < N) C[(size_t)r * ldc + c] = v;
    return 0.f;
  }
  __device__ __forceinline__ void finish(float*, float) const {}
};

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

int main(int argc, char** argv) {
  int batch = argc > 1 ? atoi(argv[1]) : 36, M = argc > 2 ? atoi(argv[2]) : 1024, N = argc > 3 ? atoi(argv[3]) : 512,
      K = argc > 4 ? atoi(argv[4]) : 512, iters = argc > 5 ? atoi(argv[5]) : 20;
  if (K % 32) { printf("K %% 32\n"); return 1; }
  std::mt19937 rng(1);
  std::normal_distribution<float> nd(0.f, 1.f);
  std::vector<float> hA((size_t)batch * M * K), hB((size_t)batch * N * K);
  for (auto& v : hA) { v = nd(rng); if (v < 0) v = 0; }     // ReLU-like activations
  for (auto& v : hB) v = 0.05f * nd(rng);
  float *dA, *dB, *dC; __bf16 *pA, *pB;
  CK(hipMalloc(&dA, hA.size() * 4)); CK(hipMalloc(&dB, hB.size() * 4)); CK(hipMalloc(&dC, (size_t)batch * M * N * 4));
  CK(hipMalloc(&pA, hA.size() * 6)); CK(hipMalloc(&pB, hB.size() * 6));
  CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemset(dC, 0xff, (size_t)batch * M * N * 4));
  hipStream_t st; CK(hipStreamCreate(&st));
  auto split = [&](const float* x, int rows, __bf16* out) {
    hipLaunchKernelGGL(x3_split_rows_kernel, dim3(1024, batch), dim3(256), 0, st, x, rows, K, K, (long long)rows * K, out,
                       (long long)3 * rows * K);
  };
  split(dA, M, pA); split(dB, N, pB);
#ifndef TILE
#define TILE 128
#endif
  using Cfg = X3Cfg<TILE>;
  struct NoMirror {
    template <class P> __device__ void operator()(const EpiStore&, P&, float*, int, int, int, int, f32x16 (&)[Cfg::T][Cfg::T], const PipeAccMap<Cfg>&) const {}
  };
  EpiStore e{{}, dC, N, M, N, (long long)M * N};
  const unsigned gt = (M + TILE - 1) / TILE;
  dim3 grid(SYMM_SKIP ? gt * (gt + 1) / 2 : (unsigned)(gt * ((N + TILE - 1) / TILE) * batch));
  auto run = [&]() {
    hipLaunchKernelGGL((gemm_x3_kernel<Cfg, EpiStore, NoMirror>), grid, dim3(256), 0, st, pA, M, (long long)3 * M * K, pB, N,
                       (long long)3 * N * K, K, e, NoMirror{});
  };
  run(); CK(hipStreamSynchronize(st)); CK(hipGetLastError());
  std::vector<float> hC((size_t)batch * M * N);
  CK(hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost));
  double worst = 0, worst_f32 = 0;
  std::uniform_int_distribution<size_t> pick(0, hC.size() - 1);
  for (int s = 0; s < 4000; ++s) {
    size_t o = s < 8 ? (s & 1 ? hC.size() - 1 - s : s) : pick(rng);
    int z = (int)(o / ((size_t)M * N)); size_t rc = o % ((size_t)M * N); int r = (int)(rc / N), c = (int)(rc % N);
    double ref = 0, mag = 0; float f = 0.f;
    for (int k = 0; k < K; ++k) {
      double a = hA[((size_t)z * M + r) * K + k], b = hB[((size_t)z * N + c) * K + k];
      ref += a * b; mag += fabs(a * b); f = fmaf((float)a, (float)b, f);
    }
    worst = fmax(worst, fabs(hC[o] - ref) / (mag + 1e-30));
    worst_f32 = fmax(worst_f32, fabs((double)f - ref) / (mag + 1e-30));
  }
  printf("max |err| / sum|a b|: x3 %.3e   (sequential f32 fma chain: %.3e)\n", worst, worst_f32);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) run();
  CK(hipEventRecord(e0, st));
  for (int i = 0; i < iters; ++i) run();
  CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= iters;
  const double fl = 2.0 * batch * M * N * K;
  printf("batch %d M %d N %d K %d: %.1f us  %.1f TFLOP/s f32-equivalent (%.1f bf16 TFLOP/s executed)\n", batch, M, N, K,
         ms * 1e3, fl / ms / 1e9, 6 * fl / ms / 1e9);
  CK(hipEventRecord(e0, st));
  for (int i = 0; i < iters; ++i) split(dA, M, pA);
  CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
  CK(hipEventElapsedTime(&ms, e0, e1)); ms /= iters;
  printf("split of A (%zu MB f32): %.1f us\n", hA.size() * 4 >> 20, ms * 1e3);
  return worst < 5e-7 ? 0 : 2;
}
