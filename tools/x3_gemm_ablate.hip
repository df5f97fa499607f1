// Where does the bf16x3 GEMM core's time go?  Stand-alone timing of gemm_x3_kernel (csrc/mfma_x3.h) on the shapes of the
// 512-channel / 256-channel Winograd layers, built with ablation macros (X3_ABL_NO_DMA, X3_ABL_NO_FRAG, X3_ABL_NO_MFMA,
// X3_ABL_NO_BARRIER).  Results are garbage under ablation; only the time is read.   tools/x3_gemm_ablate.sh
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstring>
#include "mfma_x3.h"

struct EpiStore : X3NoPrefetch<EpiStore> {
  static constexpr bool SYMM = false;
  float* C; int M, N, ldc; long long strideC; float* c;
  using X3NoPrefetch<EpiStore>::apply;
  __device__ __forceinline__ void set_batch(unsigned bz) { c = C + bz * strideC; }
  __device__ __forceinline__ float apply(int r, int col, float v) const {
#ifdef X3_ABL_NO_STORE
    if (v == 12345.678f)
#endif
    if (r < M && col < N) c[(size_t)r * ldc + col] = v;
    return 0.f;
  }
  __device__ __forceinline__ void finish(float*, float) const {}
};
struct NoMirror {
  template <class Epi, class Pre, class Acc, class Map>
  __device__ __forceinline__ void operator()(const Epi&, Pre&, float*, int, int, int, int, Acc&, const Map&) const {}
};
#ifndef TILE
#define TILE 128
#endif
int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 1024, N = argc > 2 ? atoi(argv[2]) : 512, K = argc > 3 ? atoi(argv[3]) : 512;
  const int batch = argc > 4 ? atoi(argv[4]) : 36, reps = 20;
#ifdef USE_K16
  using Cfg = X3CfgK16<3>;
#else
  using Cfg = X3Cfg<TILE, 3>;
#endif
  __bf16 *A, *B; float* C;
  const size_t ea = x3_panel_elems(M, K), eb = x3_panel_elems(N, K);
  hipMalloc(&A, ea * batch * 2); hipMalloc(&B, eb * batch * 2); hipMalloc(&C, (size_t)M * N * batch * 4); hipMemset(C, 0xff, (size_t)M * N * batch * 4);
  {  // small integers: exact in the h plane, the products check the indexing (C = K * a * b)
    std::vector<unsigned short> ha(ea * batch, 0), hb(eb * batch, 0);
    auto fill = [&](std::vector<unsigned short>& v, int rows, size_t per) {
      for (int z = 0; z < batch; ++z)
        for (int kb = 0; kb < K / 32; ++kb)
          for (int r = 0; r < rows; ++r)
            for (int k = 0; k < 32; ++k) {
              const float val = (float)((r + 3 * (kb * 32 + k) + z) % 5 - 2);
              unsigned u; memcpy(&u, &val, 4);
              v[z * per + ((size_t)(kb * 3 + 0) * rows + r) * 32 + k] = (unsigned short)(u >> 16);
            }
    };
    fill(ha, M, ea); fill(hb, N, eb);
    hipMemcpy(A, ha.data(), ea * batch * 2, hipMemcpyHostToDevice); hipMemcpy(B, hb.data(), eb * batch * 2, hipMemcpyHostToDevice);
  }
  EpiStore epi{{}, C, M, N, N, (long long)M * N, nullptr};
  unsigned grid = ((M + TILE - 1) / TILE) * ((N + TILE - 1) / TILE) * batch;
#ifdef PERSIST
  grid = (grid + 7) / 8 * 8;
  if (grid > PERSIST) grid = PERSIST;
#endif
  hipFuncSetAttribute((const void*)gemm_x3_kernel<Cfg, EpiStore, NoMirror>, hipFuncAttributeMaxDynamicSharedMemorySize, 0);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9f;
  for (int r = 0; r < reps; ++r) {
    hipEventRecord(e0);
#ifdef PERSIST
    hipLaunchKernelGGL((gemm_x3_persistent_kernel<Cfg, EpiStore>), dim3(grid), dim3(Cfg::NT), 0, 0, A, M, (long long)ea, B, N,
                       (long long)eb, K, batch, epi);
#else
    hipLaunchKernelGGL((gemm_x3_kernel<Cfg, EpiStore, NoMirror>), dim3(grid), dim3(Cfg::NT), 0, 0, A, M, (long long)ea, B, N,
                       (long long)eb, K, epi, NoMirror{}, 0, 0);
#endif
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (r > 2 && ms < best) best = ms;
  }
  if (hipGetLastError() != hipSuccess) { printf("launch failed\n"); return 1; }
  {  // spot check (meaningful without ablation macros only)
    std::vector<float> hc((size_t)M * N);
    int bad = 0;
    for (int z : {0, batch / 2, batch - 1}) {
    hipMemcpy(hc.data(), C + (size_t)z * M * N, hc.size() * 4, hipMemcpyDeviceToHost);
    for (int r = 0; r < M; r += 37)
      for (int c = 0; c < N; c += 29) {
        double ref = 0;
        for (int k = 0; k < K; ++k) ref += (double)((r + 3 * k + z) % 5 - 2) * ((c + 3 * k + z) % 5 - 2);
        if (hc[(size_t)r * N + c] != (float)ref) ++bad;
      }
    }
    printf("%s ", bad ? "MISMATCH" : "ok");
  }
  const double flop = 2.0 * M * N * K * batch * 6;
  printf("M %d N %d K %d batch %d tile %d: %.1f us  %.0f TFLOP/s bf16 (%.2f of 2500)  grid %u\n", M, N, K, batch, TILE,
         best * 1e3, flop / best / 1e9, flop / best / 1e9 / 2500, grid);
  return 0;
}
