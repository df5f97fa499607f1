// Timing ablations of the fused bf16x3 Winograd kernel (csrc/winograd_fused_x3.hip) on one layer shape, random data.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=on [-DX3_NO_TRANSFORM|-DX3_NO_SPLIT|-DX3_NO_VSTORE|-DX3_NO_MFMA|-DX3_NO_U|-DX3_NO_RAW ...]
//         -I strotss-tensorflow_amd/csrc tools/fused_x3_ablate.hip -o /tmp/fx3 && /tmp/fx3 [hw] [cin] [cout]
// (results are wrong with any ablation defined; the plain build prints the kernel's time)
#include "winograd_fused_x3.hip"
#include <cstdio>
#include <vector>
int main(int argc, char** argv) {
  const int hw = argc > 1 ? atoi(argv[1]) : 1024, cin = argc > 2 ? atoi(argv[2]) : 64, cout = argc > 3 ? atoi(argv[3]) : 64;
  float *in, *U, *out, *bias; void* Ux;
  const size_t nin = (size_t)hw * hw * cin, nout = (size_t)hw * hw * cout, nu = (size_t)36 * cin * cout;
  (void)hipMalloc(&in, nin * 4); (void)hipMalloc(&out, nout * 4); (void)hipMalloc(&U, nu * 4); (void)hipMalloc(&bias, cout * 4);
  (void)hipMalloc(&Ux, nu * 6);
  std::vector<float> h(nin > nu ? nin : nu);
  unsigned s = 12345u;
  for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 65536.0f - 0.3f; }
  (void)hipMemcpy(in, h.data(), nin * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(U, h.data(), nu * 4, hipMemcpyHostToDevice);
  (void)hipMemset(bias, 0, cout * 4);
  st_winograd43_pack_x3(U, cout, cin, Ux, 0);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float best = 1e9f;
  for (int i = 0; i < 6; ++i) {
    (void)hipEventRecord(e0, 0);
    int rc = st_winograd43_fused_x3(in, hw, hw, cin, Ux, bias, cout, nullptr, 1, out, nullptr, nullptr, 0);
    (void)hipEventRecord(e1, 0); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    if (rc) { printf("rc=%d\n", rc); return 1; }
    if (i > 0 && ms < best) best = ms;
  }
  printf("%.1f us\n", best * 1e3);
  return 0;
}
