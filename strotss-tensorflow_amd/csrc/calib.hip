// Box calibration for bench.py: three tiny kernels whose rates say what THIS device sustains, so that bench lines taken on
// different boxes of the pool can be compared (boxes differ by up to 5 % in steps/s on the same binary; a register-only MFMA
// loop on random-ish operands differs by up to 12 % between MI355X devices -- MI355X_MICROARCH.md, "DVFS give-back" (5)).
// Nothing here is on the product path (the metric's step is run_strotss.py:131-148 of the reference).
#include "internal.h"

namespace {

typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

// one wave per SIMD, four independent accumulator chains, operands in registers: the dense f32 MFMA rate and the clock held
__global__ __launch_bounds__(256) void calib_mfma_f32_kernel(float* __restrict__ sink, int iters, unsigned long long* __restrict__ clk) {
  f32x16_t a0 = {}, a1 = {}, a2 = {}, a3 = {};
  const float x = threadIdx.x * 1e-3f + 1.0f, y = 0.5f + blockIdx.x * 1e-4f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, a3, 0, 0, 0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0;
  for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
  sink[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

__global__ __launch_bounds__(256) void calib_mfma_bf16_kernel(float* __restrict__ sink, int iters, unsigned long long* __restrict__ clk) {
  f32x16_t a0 = {}, a1 = {}, a2 = {}, a3 = {};
  bf16x8_t x, y;
  for (int k = 0; k < 8; ++k) {            // values with full bf16 mantissas, different per lane (power depends on the data)
    x[k] = (__bf16)(1.0f + 0.0078125f * (float)((threadIdx.x * 7 + k * 13) & 127));
    y[k] = (__bf16)(0.5f + 0.00390625f * (float)((threadIdx.x * 11 + k * 5 + blockIdx.x) & 127));
  }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y, x, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, x, a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y, y, a3, 0, 0, 0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0;
  for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
  sink[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

// The loops above run on constant operands in registers: every box holds ~2.39 GHz in them and they do not tell boxes apart
// (two boxes 5.6 % apart in steps/s calibrated 0.8 % apart).  What differs between devices is the clock they hold under a
// POWER-hungry load (MI355X_MICROARCH.md, "DVFS give-back" (5): an LDS-read + MFMA loop on random operands ran 12 % apart
// across devices).  This one is that loop: pseudo-random bf16 data in 64 KB of LDS, every MFMA's operands re-read by
// ds_read_b128 from a rotating offset, four independent accumulators, two waves per SIMD.
__global__ __launch_bounds__(512) void calib_mfma_lds_kernel(float* __restrict__ sink, int iters, unsigned long long* __restrict__ clk) {
  __shared__ __attribute__((aligned(16))) unsigned int lds[16384];          // 64 KB
  unsigned st = 0x9E3779B9u * (blockIdx.x * 512u + threadIdx.x + 1u);
  for (int i = threadIdx.x; i < 16384; i += 512) {
    st = st * 1664525u + 1013904223u;
    // two bf16 per word with exponents near 1.0 and random mantissas / signs (finite, no denormals)
    const unsigned m = st >> 9;
    lds[i] = (0x3F80u | (m & 0x7Fu) | ((m >> 7) & 0x8000u)) | ((0x3F80u | ((m >> 8) & 0x7Fu) | ((m >> 15) & 0x8000u)) << 16);
  }
  __syncthreads();
  f32x16_t a0 = {}, a1 = {}, a2 = {}, a3 = {};
  const unsigned lane_off = (threadIdx.x & 511u) * 16u;                      // bytes; 512 lanes x 16 B = 8 KB per sweep
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  unsigned rot = 0;
  for (int i = 0; i < iters; ++i) {
    const unsigned char* base = reinterpret_cast<const unsigned char*>(lds);
    const bf16x8_t x0 = *reinterpret_cast<const bf16x8_t*>(base + ((rot + lane_off) & 0xFFF0u));
    const bf16x8_t y0 = *reinterpret_cast<const bf16x8_t*>(base + ((rot + lane_off + 8192u) & 0xFFF0u));
    const bf16x8_t x1 = *reinterpret_cast<const bf16x8_t*>(base + ((rot + lane_off + 16384u) & 0xFFF0u));
    const bf16x8_t y1 = *reinterpret_cast<const bf16x8_t*>(base + ((rot + lane_off + 24576u) & 0xFFF0u));
    a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x0, y0, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y0, x1, a1, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x1, y1, a2, 0, 0, 0);
    a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y1, x0, a3, 0, 0, 0);
    rot += 4112u;                                                            // a different 16-byte slot pattern every trip
    if ((i & 63) == 63) {                                                    // keep the sums finite without leaving the pipe idle
#pragma unroll
      for (int r = 0; r < 16; ++r) { a0[r] *= 0.001f; a1[r] *= 0.001f; a2[r] *= 0.001f; a3[r] *= 0.001f; }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0;
  for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
  sink[blockIdx.x * 512 + threadIdx.x] = s;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

// Dependent-load latency: one lane per workgroup chases `steps` links of a pre-built random cycle (next[i] = index of the
// following element, 64-byte stride so that every hop is a new line).  Footprints of 2 MB / 64 MB / 1 GB put the chain in the L2,
// the Infinity Cache and HBM.  clk[b] = s_memtime ticks of workgroup b's chase.
__global__ __launch_bounds__(64) void calib_chase_kernel(const unsigned* __restrict__ next, unsigned start_stride, int steps,
                                                         unsigned* __restrict__ sink, unsigned long long* __restrict__ clk) {
  if (threadIdx.x != 0) return;
  unsigned i = blockIdx.x * start_stride;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int s = 0; s < steps; ++s) i = __builtin_nontemporal_load(next + (size_t)i * 16);
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  sink[blockIdx.x] = i;
  clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0;
}

// 16 bytes per lane, grid-stride: a plain streaming copy
__global__ __launch_bounds__(256) void calib_copy_kernel(const f32x4* __restrict__ src, f32x4* __restrict__ dst, size_t n16) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}

}  // namespace

extern "C" {

int strotss_calib_mfma(int bf16, int blocks, int iters, float* sink, unsigned long long* clocks, void* stream) {
  ST_CHECK_ARG(sink && clocks && blocks > 0 && blocks <= 4096 && iters > 0, STROTSS_EINVAL);
  if (bf16 == 2) {                                   // the loaded loop: 512-thread workgroups, 4 MFMAs (32768 FLOP each) per trip and wave
    hipLaunchKernelGGL(calib_mfma_lds_kernel, dim3((unsigned)blocks), dim3(512), 0, (hipStream_t)stream, sink, iters, clocks);
    ST_LAUNCH_RET();
  }
  if (bf16) hipLaunchKernelGGL(calib_mfma_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, sink, iters, clocks);
  else hipLaunchKernelGGL(calib_mfma_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, sink, iters, clocks);
  ST_LAUNCH_RET();
}

int strotss_calib_chase(const unsigned* next, unsigned start_stride, int blocks, int steps, unsigned* sink,
                        unsigned long long* clocks, void* stream) {
  ST_CHECK_ARG(next && sink && clocks && blocks > 0 && blocks <= 4096 && steps > 0, STROTSS_EINVAL);
  hipLaunchKernelGGL(calib_chase_kernel, dim3((unsigned)blocks), dim3(64), 0, (hipStream_t)stream, next, start_stride, steps, sink,
                     clocks);
  ST_LAUNCH_RET();
}

int strotss_calib_copy(const void* src, void* dst, size_t bytes, void* stream) {
  ST_CHECK_ARG(src && dst && bytes >= 16 && bytes % 16 == 0, STROTSS_EINVAL);
  hipLaunchKernelGGL(calib_copy_kernel, dim3(256 * 16), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const f32x4*>(src),
                     reinterpret_cast<f32x4*>(dst), bytes / 16);
  ST_LAUNCH_RET();
}

}  // extern "C"
