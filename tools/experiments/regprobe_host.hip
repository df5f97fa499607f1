// Host side of the register-integrity probe (tools/experiments/gen_regprobe.py): launches regprobe.co repeatedly and
// reports every register whose dumped value differs from the pattern it was given.   usage: regprobe_host regprobe.co ITER SPIN
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <map>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main(int argc, char** argv) {
  const int iters = atoi(argv[2]), spin = atoi(argv[3]);
  const int NS = 96, NV = 64, WGS = 256, WAVES = WGS * 4;
  hipModule_t mod; hipFunction_t fn;
  CK(hipModuleLoad(&mod, argv[1])); CK(hipModuleGetFunction(&fn, mod, "regprobe"));
  const size_t per_wave = (size_t)(NS + NV) * 64, total = per_wave * WAVES;
  unsigned* out; float* dummy;
  CK(hipMalloc(&out, total * 4)); CK(hipMalloc(&dummy, 1 << 16)); CK(hipMemset(dummy, 0, 1 << 16));
  std::vector<unsigned> h(total);
  struct { void* out; int spin; int pad; void* dummy; } args = {out, spin, 0, dummy};
  size_t asz = sizeof(args);
  void* cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &asz, HIP_LAUNCH_PARAM_END};
  std::map<int, long> bad_s, bad_v; long bad_launches = 0, shown = 0;
  for (int it = 0; it < iters; ++it) {
    CK(hipMemset(out, 0xEE, total * 4));
    CK(hipModuleLaunchKernel(fn, WGS, 1, 1, 256, 1, 1, 0, 0, nullptr, cfg));
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h.data(), out, total * 4, hipMemcpyDeviceToHost));
    bool any = false;
    for (int w = 0; w < WAVES; ++w) {
      const unsigned* d = h.data() + (size_t)w * per_wave;
      for (int n = 2; n < NV; ++n) for (int l = 0; l < 64; ++l) {
        const unsigned v = d[n * 64 + l], want = 0x7B000000u | n;
        if (v != want) { bad_v[n]++; any = true; if (shown++ < 40) printf("iter %d wave %d v%d lane %d = %08x (want %08x)\n", it, w, n, l, v, want); }
      }
      for (int n = 8; n < NS; ++n) {
        const unsigned v = d[(NV + n) * 64], want = 0x5A000000u | n;
        bool lanes_differ = false;
        for (int l = 1; l < 64; ++l) lanes_differ |= d[(NV + n) * 64 + l] != v;
        if (v != want || lanes_differ) { bad_s[n]++; any = true; if (shown++ < 40) printf("iter %d wave %d s%d = %08x (want %08x)%s\n", it, w, n, v, want, lanes_differ ? " lanes differ" : ""); }
      }
      if (d[(NV + 2) * 64] != (unsigned)(w / 4)) { bad_s[2]++; any = true; if (shown++ < 40) printf("iter %d wave %d s2 (workgroup id) = %08x (want %08x)\n", it, w, d[(NV + 2) * 64], w / 4); }
    }
    bad_launches += any;
  }
  printf("regprobe: %ld of %d launches with a changed register\n", bad_launches, iters);
  for (auto& kv : bad_s) printf("  s%d: %ld\n", kv.first, kv.second);
  for (auto& kv : bad_v) printf("  v%d: %ld\n", kv.first, kv.second);
  return 0;
}
