// Do float atomics lose updates when a second process shares the GPU?  (tests/test_hip_regions.py runs two ranks on one card)
//   hipcc -O3 --offload-arch=gfx950 tools/atomic_probe.hip -o /tmp/ap && (/tmp/ap & /tmp/ap; wait)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
__global__ void hammer(float* f, unsigned* u, int slots, int reps, const float* gate) {
  const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
  for (int r = 0; r < reps; ++r) {
    unsigned s = (t * 2654435761u + r * 40503u) % slots;
#ifdef CONTIG
    // the tap adjoint's pattern: a wave adds to 64 CONSECUTIVE floats (two whole cache lines), many workgroups to the same lines
    s = (((blockIdx.x * 2654435761u + r * 40503u) >> 8) % (slots / 256)) * 256 + threadIdx.x;
#endif
    if (gate && !(gate[(t + r) & 4095] > 0.f)) continue;          // data-dependent lane mask, as in the tap adjoint
    atomicAdd(&f[s], 1.0f);
    atomicAdd(&u[s], 1u);
  }
}
int main() {
  const int slots = 4096, reps = 64, blocks = 2048, threads = 256;
  float* f; unsigned* u; float* gate = nullptr;
  hipMalloc(&f, slots * 4); hipMalloc(&u, slots * 4);
  unsigned long long want_gated = 0;
  if (getenv("GATE")) {
    std::vector<float> hg(4096);
    for (int i = 0; i < 4096; ++i) hg[i] = ((i * 2654435761u) >> 7) % 3 ? 1.f : -1.f;
    hipMalloc(&gate, 4096 * 4); hipMemcpy(gate, hg.data(), 4096 * 4, hipMemcpyHostToDevice);
    for (unsigned long long t = 0; t < (unsigned long long)blocks * threads; ++t)
      for (int r = 0; r < reps; ++r) want_gated += hg[(t + r) & 4095] > 0.f;
  }
  int bad_f = 0, bad_u = 0;
  std::vector<float> hf(slots); std::vector<unsigned> hu(slots);
  for (int it = 0; it < 200; ++it) {
    hipMemset(f, 0, slots * 4); hipMemset(u, 0, slots * 4);
    hipLaunchKernelGGL(hammer, dim3(blocks), dim3(threads), 0, 0, f, u, slots, reps, gate);
    hipDeviceSynchronize();
    hipMemcpy(hf.data(), f, slots * 4, hipMemcpyDeviceToHost); hipMemcpy(hu.data(), u, slots * 4, hipMemcpyDeviceToHost);
    double sf = 0; unsigned long long su = 0; bool mism = false;
    for (int i = 0; i < slots; ++i) { sf += hf[i]; su += hu[i]; if (hf[i] != (float)hu[i]) mism = true; }
    const unsigned long long want = gate ? want_gated : (unsigned long long)blocks * threads * reps;
    if (su != want) ++bad_u;
    if (sf != (double)want || mism) ++bad_f;
  }
  printf("200 launches: float totals wrong in %d, integer totals wrong in %d\n", bad_f, bad_u);
  return 0;
}
