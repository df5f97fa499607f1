// Could the bf16x3 GEMM stream its operands global -> VGPR (no LDS) with a deeper pipeline than the 96 KiB LDS ring allows?
// One workgroup = 4 waves on a 128 x 128 tile of the batched product; every wave loads the MFMA fragments of its 64 x 64
// quarter itself: per K-step of 16, 12 dwordx4 loads per lane (A: 2 row tiles x 3 planes, B likewise) = 12 KiB per wave, of
// which the A half is also loaded by the wave beside it and the B half by the wave below (the L1 has to merge them, or the L2
// sees twice the LDS ring's traffic).  S stages of 48 registers stay in flight.  No MFMA here: the question is the delivery
// rate in UNIQUE bytes (24 KiB per workgroup and step), to compare with the 49 GB/s per CU of the LDS-DMA ring
// (tools/x3_gemm_ablate.hip -DX3_ABL_NO_MFMA).   hipcc -O3 --offload-arch=gfx950 tools/reg_stream_probe.hip -o /tmp/rs
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int S>
__global__ __launch_bounds__(256) void stream(const char* __restrict__ A, int M, const char* __restrict__ B, int N, int K,
                                              int batch, unsigned* __restrict__ sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, hh = lane >> 5;
  const unsigned gx = N / 128, gy = M / 128;
  const unsigned tile = blockIdx.x;
  const unsigned bz = tile / (gx * gy), rem = tile % (gx * gy);
  const int m0 = (rem / gx) * 128, n0 = (rem % gx) * 128;
  const size_t pa = (size_t)M * 64, pb = (size_t)N * 64;              // bytes per plane and K-block of 32
  const char* a = A + (size_t)bz * 3 * M * K * 2 + (size_t)(m0 + wm * 64 + l31) * 64 + hh * 16;
  const char* b = B + (size_t)bz * 3 * N * K * 2 + (size_t)(n0 + wn * 64 + l31) * 64 + hh * 16;
  const int steps = K / 16;
  u32x4 r[S][12];
  unsigned acc = 0;
  auto issue = [&](int s, int slot) {
    const size_t ko = (size_t)(s >> 1) * 3, half = (s & 1) * 32;
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        r[slot][p * 2 + i] = *reinterpret_cast<const u32x4*>(a + (ko + p) * pa + half + i * 2048);
        r[slot][6 + p * 2 + i] = *reinterpret_cast<const u32x4*>(b + (ko + p) * pb + half + i * 2048);
      }
  };
#pragma unroll
  for (int s = 0; s < S - 1; ++s) issue(s, s);
  for (int s0 = 0; s0 < steps; s0 += S) {
#pragma unroll
    for (int j = 0; j < S; ++j) {
      const int s = s0 + j;
      if (s + S - 1 < steps) issue(s + S - 1, (j + S - 1) % S);
#pragma unroll
      for (int q = 0; q < 12; ++q) acc ^= r[j][q][0] ^ r[j][q][3];      // consume (forces the wait for this stage only)
    }
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

template <int S>
void run(int M, int N, int K, int batch) {
  char *A, *B; unsigned* sink;
  const size_t ea = (size_t)3 * M * K * 2 * batch, eb = (size_t)3 * N * K * 2 * batch;
  hipMalloc(&A, ea); hipMalloc(&B, eb); hipMalloc(&sink, 64);
  hipMemset(A, 1, ea); hipMemset(B, 2, eb);
  const unsigned grid = (M / 128) * (N / 128) * batch;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9f;
  for (int r = 0; r < 12; ++r) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(stream<S>, dim3(grid), dim3(256), 0, 0, A, M, B, N, K, batch, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (r > 2 && ms < best) best = ms;
  }
  const double unique = (double)grid * (K / 16) * 24576.0;
  printf("stages %d: M %d N %d K %d batch %d: %.1f us, unique operand bytes %.0f MB -> %.1f TB/s = %.1f GB/s per CU (requested: 2x)\n", S, M,
         N, K, batch, best * 1e3, unique / 1e6, unique / best / 1e9, unique / best / 1e9 / 256 * 1e3);
  hipFree(A); hipFree(B); hipFree(sink);
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 1024, N = argc > 2 ? atoi(argv[2]) : 512, K = argc > 3 ? atoi(argv[3]) : 512;
  const int batch = argc > 4 ? atoi(argv[4]) : 36;
  run<2>(M, N, K, batch); run<4>(M, N, K, batch); run<8>(M, N, K, batch);
  return 0;
}
