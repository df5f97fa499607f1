// Which SIMD does wave w of a 768- / 1024-thread workgroup run on?  (HW_REG_HW_ID: bits 5:4 = SIMD id on gfx9.)
//   hipcc -O3 --offload-arch=gfx950 tools/experiments/simd_map_probe.hip -o /tmp/simd_map_probe && /tmp/simd_map_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned* out, int nw) {
  __shared__ float pad[36000];                     // 144 KB: one workgroup per CU, as the fused Winograd kernel
  pad[threadIdx.x] = 0.f;
  unsigned id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * nw + (threadIdx.x >> 6)] = id;
}
int main() {
  unsigned* d; hipMalloc(&d, 4096 * 16 * 4);
  for (int nw : {12, 16}) {
    const int blocks = 512;
    hipMemset(d, 0, 4096 * 16 * 4);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(nw * 64), 0, 0, d, nw);
    hipDeviceSynchronize();
    static unsigned h[4096 * 16];
    hipMemcpy(h, d, blocks * nw * 4, hipMemcpyDeviceToHost);
    int pat_ok = 0, three = 0;
    for (int b = 0; b < blocks; ++b) {
      int simd[16], cnt[4] = {0, 0, 0, 0};
      for (int w = 0; w < nw; ++w) { simd[w] = (h[b * nw + w] >> 4) & 3; cnt[simd[w]]++; }
      bool same = true;
      for (int w = 4; w < nw; ++w) same = same && simd[w] == simd[w - 4];
      pat_ok += same;
      three += (cnt[0] == nw / 4 && cnt[1] == nw / 4 && cnt[2] == nw / 4 && cnt[3] == nw / 4);
      if (b < 6) { printf("nw=%d block %d simd:", nw, b); for (int w = 0; w < nw; ++w) printf(" %d", simd[w]); printf("  (cu %u se %u)\n", (h[b * nw] >> 8) & 15, (h[b * nw] >> 13) & 7); }
    }
    printf("nw=%d: waves w and w+4 on the same SIMD in %d of %d workgroups; %d waves on every SIMD in %d\n", nw, pat_ok, blocks, nw / 4, three);
  }
  return 0;
}
