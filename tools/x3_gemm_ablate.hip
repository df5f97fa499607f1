// Where does the bf16x3 GEMM core's time go?  Stand-alone timing of gemm_x3_kernel (csrc/mfma_x3.h) on the shapes of the
// 512-channel / 256-channel Winograd layers, built with ablation macros (X3_ABL_NO_DMA, X3_ABL_NO_FRAG, X3_ABL_NO_MFMA,
// X3_ABL_NO_BARRIER).  Results are garbage under ablation; only the time is read.   tools/x3_gemm_ablate.sh
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstring>
#include "mfma_x3.h"

// (experiment, kept out of the product header) PERSISTENT form of the batched product (K16 ring, two workgroups per CU, plain store epilogues): the launch holds
// 2 x 256 workgroups (or fewer) and each walks a strided list of tiles.  A workgroup of the one-tile kernel pays about
// 5 us of launch, first-tile latency and store time around 11 us of MFMAs (36 x (1024 x 512 x 512), tools/x3_gemm_ablate
// .hip); here the next tile's first loads are in flight while the current tile is stored, and nothing is relaunched.
// XCD x (workgroups x, x + 8, ...) owns the contiguous tile range [x, x + 1) * ceil(tiles / 8): its 64 workgroups walk
// 64 consecutive tiles at a time (N-tile fastest, then M-tile, then batch), so an XCD's L2 holds the panels it streams.
template <class Cfg, class Epi>
__global__ __launch_bounds__(Cfg::NT) void gemm_x3_persistent_kernel(const __bf16* __restrict__ A, int M, long long strideA,
                                                                     const __bf16* __restrict__ B, int N, long long strideB,
                                                                     int K, int batch, Epi epi) {
  static_assert(Cfg::K16 && !Epi::SYMM, "persistent form: K16 ring, store epilogues");
  __shared__ __attribute__((aligned(1024))) unsigned char lds[Cfg::LDS_BYTES];
  const unsigned gx = (N + Cfg::BN - 1) / Cfg::BN, gy = (M + Cfg::BM - 1) / Cfg::BM;
  const unsigned total = gx * gy * (unsigned)batch;
  const unsigned x = blockIdx.x & 7, slot = blockIdx.x >> 3, per_x = gridDim.x >> 3;     // gridDim.x % 8 == 0
  const unsigned chunk = (total + 7) / 8;
  const unsigned hi = min(total, (x + 1) * chunk);
  unsigned tile = x * chunk + slot;
  if (tile >= hi) return;
  auto locate = [&](unsigned tl, unsigned& bz, int& m0, int& n0) {
    bz = tl / (gx * gy);
    const unsigned rem = tl - bz * (gx * gy);
    m0 = (int)(rem / gx) * Cfg::BM; n0 = (int)(rem % gx) * Cfg::BN;
  };
  unsigned bz; int m0, n0;
  locate(tile, bz, m0, n0);
  PipeAccMap<Cfg> map;
  {
    X3OperandK16 oa(A + (long long)bz * strideA, M, m0), ob(B + (long long)bz * strideB, N, n0);
    x3_k16_prologue<Cfg>(lds, oa, ob);
    for (;;) {
      f32x16 acc[2][2];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
      x3_mainloop_k16<Cfg>(lds, K >> 4, oa, ob, acc);
      const unsigned cbz = bz; const int cm0 = m0, cn0 = n0;
      tile += per_x;
      const bool more = tile < hi;
      // every wave's fragment reads are complete (the last step used registers only): the ring is free
      __builtin_amdgcn_s_barrier();
      if (more) {
        locate(tile, bz, m0, n0);
        oa = X3OperandK16(A + (long long)bz * strideA, M, m0);
        ob = X3OperandK16(B + (long long)bz * strideB, N, n0);
        x3_k16_prologue<Cfg>(lds, oa, ob);
      }
      Epi e = epi;                          // (set_batch may offset the epilogue's pointers in place)
      e.set_batch(cbz);
      typename Epi::template Pre<Cfg::T> pre;
#pragma unroll
      for (int im = 0; im < 2; ++im)
#pragma unroll
        for (int in = 0; in < 2; ++in)
#pragma unroll
          for (int reg = 0; reg < 16; ++reg)
            e.apply(pre, im, in, reg, cm0 + map.row(im, reg), cn0 + map.colof(in), acc[im][in][reg]);
      if (!more) break;
    }
  }
}


struct EpiStore : X3NoPrefetch<EpiStore> {
  static constexpr bool SYMM = false;
  float* C; int M, N, ldc; long long strideC; float* c;
  long long* trace;          // TRACE: per workgroup {start, end (100 MHz real time), hw id, xcc id}
  using X3NoPrefetch<EpiStore>::apply;
  __device__ __forceinline__ void set_batch(unsigned bz) {
    c = C + bz * strideC;
#ifdef TRACE
    if (threadIdx.x == 0) { trace[blockIdx.x * 4] = wall_clock64(); trace[262144 + blockIdx.x] = clock64(); }
#endif
  }
  __device__ __forceinline__ float apply(int r, int col, float v) const {
#ifdef X3_ABL_NO_STORE
    if (v == 12345.678f)
#endif
    if (r < M && col < N) c[(size_t)r * ldc + col] = v;
    return 0.f;
  }
  __device__ __forceinline__ void finish(float*, float) const {
#ifdef TRACE
    if (threadIdx.x == 0) {
      trace[blockIdx.x * 4 + 1] = wall_clock64();
      trace[262144 + blockIdx.x] = clock64() - trace[262144 + blockIdx.x];
      trace[blockIdx.x * 4 + 2] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
      trace[blockIdx.x * 4 + 3] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    }
#endif
  }
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
  long long* trace; hipMalloc(&trace, (size_t)(262144 + 65536) * 8); hipMemset(trace, 0, (size_t)(262144 + 65536) * 8);
  EpiStore epi{{}, C, M, N, N, (long long)M * N, nullptr, trace};
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
#ifdef TRACE
  {
    std::vector<long long> h((size_t)grid * 4);
    hipMemcpy(h.data(), trace, h.size() * 8, hipMemcpyDeviceToHost);
    long long t0 = h[0], t1 = 0;
    for (unsigned i = 0; i < grid; ++i) { if (h[i * 4] < t0) t0 = h[i * 4]; if (h[i * 4 + 1] > t1) t1 = h[i * 4 + 1]; }
    double sum = 0; long long mn = 1 << 30, mx = 0;
    for (unsigned i = 0; i < grid; ++i) { long long d = h[i * 4 + 1] - h[i * 4]; sum += d; if (d < mn) mn = d; if (d > mx) mx = d; }
    std::vector<long long> hc2(grid);
    hipMemcpy(hc2.data(), trace + 262144, grid * 8, hipMemcpyDeviceToHost);
    double cs = 0; for (unsigned i = 0; i < grid; ++i) cs += hc2[i];
    printf("shader clocks per workgroup %.0f -> %.2f GHz; ", cs / grid, cs / sum * 100.0 / 1e3);
    printf("TRACE span %.1f us; workgroup duration mean %.2f min %.2f max %.2f us\n", (t1 - t0) / 100.0, sum / grid / 100.0, mn / 100.0, mx / 100.0);
    // per (xcc, se, cu): number of workgroups, busy span
    FILE* f = fopen(getenv("TRACE_OUT") ? getenv("TRACE_OUT") : "/tmp/x3trace.csv", "w");
    fprintf(f, "wg,start_us,end_us,xcc,se,cu,simd_wave\n");
    for (unsigned i = 0; i < grid; ++i) {
      const unsigned hw = (unsigned)h[i * 4 + 2], xcc = (unsigned)h[i * 4 + 3] & 0xf;
      fprintf(f, "%u,%.2f,%.2f,%u,%u,%u,%u\n", i, (h[i * 4] - t0) / 100.0, (h[i * 4 + 1] - t0) / 100.0, xcc, (hw >> 13) & 7, (hw >> 8) & 15, hw & 0xff);
    }
    fclose(f);
  }
#endif
  const double flop = 2.0 * M * N * K * batch * 6;
  printf("M %d N %d K %d batch %d tile %d: %.1f us  %.0f TFLOP/s bf16 (%.2f of 2500)  grid %u\n", M, N, K, batch, TILE,
         best * 1e3, flop / best / 1e9, flop / best / 1e9 / 2500, grid);
  return 0;
}
