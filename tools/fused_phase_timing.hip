// Phase timestamps (s_memtime) of workgroup 0, waves 0 (transforming) and 4, of the fused Winograd kernel.
//   (-DFUSED_NO_TRANSFORM / -DFUSED_NO_MFMA / -DFUSED_NO_RAW: timing ablations, without -DFUSED_PROF: plain timing)
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DFUSED_PROF -I strotss-tensorflow_amd/csrc tools/fused_phase_timing.hip -o /tmp/fpt && /tmp/fpt [hw] [cin] [cout]
#include "../strotss-tensorflow_amd/csrc/winograd_fused.hip"
#include <cstdio>
#include <vector>
int main(int argc, char** argv) {
  const int hw = argc > 1 ? atoi(argv[1]) : 1024, cin = argc > 2 ? atoi(argv[2]) : 64, cout = argc > 3 ? atoi(argv[3]) : 64;
  float *in, *U, *out, *bias;
  const size_t nin = (size_t)hw * hw * cin, nout = (size_t)hw * hw * cout, nu = (size_t)36 * cin * cout;
  hipMalloc(&in, nin * 4); hipMalloc(&out, nout * 4); hipMalloc(&U, nu * 4); hipMalloc(&bias, cout * 4);
  hipMemset(in, 0, nin * 4); hipMemset(U, 0, nu * 4); hipMemset(bias, 0, cout * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) {
    hipEventRecord(e0, 0);
    int rc = st_winograd43_fused(in, hw, hw, cin, U, bias, cout, nullptr, 1, out, nullptr, nullptr, nullptr, nullptr, 0);
    hipEventRecord(e1, 0); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("rc=%d  %.1f us\n", rc, ms * 1e3);
  }
#ifdef FUSED_PROF
  static long long h[2][64][8];
  hipMemcpyFromSymbol(h, HIP_SYMBOL(fused_prof), sizeof(h));
  for (int w = 0; w < 2; ++w) {
    printf("wave %d: phase: issue  transform  mfma_issue  store_raw  barrier | total   (cycles of s_memtime)\n", w * 4);
    for (int p = 0; p < 40; ++p) {
      long long* q = h[w][p];
      if (q[6]) { printf("  %2d epilogue: gap %lld  write0 %lld  comp0 %lld  write1 %lld  comp1 %lld  stores %lld | %lld\n", p, q[0] - h[w][p - 1][5], q[1] - q[0], q[2] - q[1], q[3] - q[2], q[4] - q[3], q[5] - q[4], q[5] - h[w][p - 1][5]); continue; }
      if (!q[0]) break;
      printf("  %2d  %6lld %6lld %6lld %6lld %6lld | %6lld\n", p, q[1] - q[0], q[2] - q[1], q[3] - q[2], q[4] - q[3], q[5] - q[4], q[5] - q[0]);
    }
  }
#endif
  return 0;
}
