// Times st_winograd43_fused on the eight layer shapes the fused kernel serves at the 1024-px scale (random data), for the
// kernel selected by STROTSS_WINO_FUSED_ROLES (0: round-1..3 kernel, 1: SIMD-specialised roles).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-slp-vectorize -I strotss-tensorflow_amd/csrc tools/fused_layer_times.hip -o /tmp/flt
#include "../strotss-tensorflow_amd/csrc/winograd_fused.hip"
#include <cstdio>
#include <vector>
int main(int argc, char** argv) {
  struct L { const char* name; int hw, cin, cout; bool dgrad; };
  const L layers[] = {{"block1_conv2 fwd", 1024, 64, 64, false},  {"block1_conv2 dgrad", 1024, 64, 64, true},
                      {"block2_conv1 fwd", 512, 64, 128, false},  {"block2_conv1 dgrad", 512, 128, 64, true},
                      {"block2_conv2 fwd", 512, 128, 128, false}, {"block2_conv2 dgrad", 512, 128, 128, true},
                      {"block3_conv1 dgrad", 256, 256, 128, true}, {"block4_conv1 dgrad", 128, 512, 256, true}};
  const int reps = argc > 1 ? atoi(argv[1]) : 20;
  const size_t nmax = (size_t)1024 * 1024 * 64;
  std::vector<float> h(nmax);
  unsigned s = 12345u;
  for (size_t i = 0; i < nmax; ++i) { s = s * 1664525u + 1013904223u; h[i] = ((s >> 8) & 0xFFFF) / 65536.f - ((i & 1) ? 0.5f : 0.25f); }
  float *in, *U, *out, *bias; unsigned *bits_in, *bits_out;
  hipMalloc(&in, nmax * 4); hipMalloc(&out, nmax * 4); hipMalloc(&U, (size_t)36 * 512 * 256 * 4); hipMalloc(&bias, 1024);
  hipMalloc(&bits_in, nmax / 4); hipMalloc(&bits_out, nmax / 4);
  hipMemcpy(in, h.data(), nmax * 4, hipMemcpyHostToDevice);
  hipMemcpy(U, h.data(), (size_t)36 * 512 * 256 * 4, hipMemcpyHostToDevice);
  hipMemcpy(bits_in, h.data(), nmax / 4, hipMemcpyHostToDevice);
  hipMemset(bias, 0, 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  double total = 0;
  for (const L& l : layers) {
    for (int i = 0; i < 3; ++i)
      st_winograd43_fused(in, l.hw, l.hw, l.cin, U, l.dgrad ? nullptr : bias, l.cout, nullptr, l.dgrad ? 0 : 1, out, nullptr, nullptr,
                          l.dgrad ? bits_in : nullptr, l.dgrad ? nullptr : bits_out, 0);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    int rc = 0;
    for (int i = 0; i < reps; ++i)
      rc |= st_winograd43_fused(in, l.hw, l.hw, l.cin, U, l.dgrad ? nullptr : bias, l.cout, nullptr, l.dgrad ? 0 : 1, out, nullptr, nullptr,
                                l.dgrad ? bits_in : nullptr, l.dgrad ? nullptr : bits_out, 0);
    hipEventRecord(e1, 0); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / reps, tflop = 2.0 * 2.25 * l.cin * l.cout * (double)l.hw * l.hw / 1e12;
    total += us;
    printf("%-20s %4d px %3d -> %3d  rc=%d  %7.1f us  %6.1f TFLOP/s (Winograd-domain)\n", l.name, l.hw, l.cin, l.cout, rc, us, tflop / (us * 1e-6));
  }
  printf("sum of the eight launches: %.1f us\n", total);
  return 0;
}
