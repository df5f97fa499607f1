// Probe: does v_mfma_f32_32x32x16_bf16 honour BLGP (B-operand lane-group broadcast) on gfx950?  Expected: blgp0 808, blgp1 16
// (lanes 0-31 of B broadcast to both K halves), blgp2 1600 (lanes 32-63 broadcast).  hipcc --offload-arch=gfx950 -O2 -w
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int BLGP>
__global__ void k(const bf16x8* a, const bf16x8* b, f32x16* c) {
  f32x16 acc = {0};
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[threadIdx.x], b[threadIdx.x], acc, 0, 0, BLGP);
  c[threadIdx.x] = acc;
}
int main() {
  bf16x8 *a, *b; f32x16* c;
  hipMalloc(&a, 64 * 16); hipMalloc(&b, 64 * 16); hipMalloc(&c, 64 * 64 * 3);
  __bf16 ha[64][8], hb[64][8];
  // A[row][k]: lane l holds row l%32, k = 8*(l/32)+i.  A = 1 everywhere.  B[k][col]: lane l col l%32, k = 8*(l/32)+i.
  // B lower half (k<8) = 1, upper half (k>=8) = 100 -> normal: sum = 8*1 + 8*100 = 808; blgp=1 (lower broadcast): 16; blgp=2: 1600
  for (int l = 0; l < 64; ++l) for (int i = 0; i < 8; ++i) { ha[l][i] = (__bf16)1.0f; hb[l][i] = (__bf16)(l < 32 ? 1.0f : 100.0f); }
  hipMemcpy(a, ha, sizeof ha, hipMemcpyHostToDevice); hipMemcpy(b, hb, sizeof hb, hipMemcpyHostToDevice);
  float out[64][16];
  k<0><<<1, 64>>>(a, b, c); hipMemcpy(out, c, sizeof out, hipMemcpyDeviceToHost); printf("blgp0 %g\n", out[0][0]);
  k<1><<<1, 64>>>(a, b, c); hipMemcpy(out, c, sizeof out, hipMemcpyDeviceToHost); printf("blgp1 %g\n", out[0][0]);
  k<2><<<1, 64>>>(a, b, c); hipMemcpy(out, c, sizeof out, hipMemcpyDeviceToHost); printf("blgp2 %g\n", out[0][0]);
  return 0;
}
