// fp32 MFMA tile engine for gfx950 (v_mfma_f32_32x32x2_f32; exact f32, 64 FLOP/clk/SIMD).
//
// A workgroup is 256 threads = 4 waves arranged 2(M) x 2(N).  A block tile is BM x BN with a
// K-step of 32; each wave owns (BM/2) x (BN/2) as TM x TN MFMA tiles of 32x32, accumulators
// in registers.  Operand tiles go global -> registers -> LDS (one LDS buffer, the next
// tile's global loads are issued before the current tile's MFMAs so their latency hides under
// the matrix work; 2-3 workgroups per CU cover the barriers).
//
// Both operands are indexed (row, k):  C[m][n] = sum_k A(m,k) * B(n,k).
// Two LDS images per operand, chosen by how the operand lies in global memory:
//   KC  (k contiguous in global, e.g. NHWC activations, feature rows):
//        LDS [row][36]  (32 k + 4 pad) -> one ds_read_b128 per lane gives the 4 k-values
//        k = 8t+4h+j (h = lane>>5, j = 0..3) for 4 consecutive MFMAs; 36-float rows make the
//        16-lane b128 groups hit 16 distinct 4-bank slots (conflict-free).
//   RC  (row contiguous in global, i.e. stored [k][row]):
//        LDS [k][ROWS+4] -> 4 ds_read_b32 (lanes = consecutive rows: conflict-free).
// Both images use the SAME k order inside a K-step, so any A/B combination is valid.
#pragma once
#include "common.h"

#define KC_LD 36

template <int ROWS>
struct OperandLds {
  static constexpr int kc_floats = ROWS * KC_LD;
  static constexpr int rc_ld = ROWS + 4;
  static constexpr int rc_floats = 32 * rc_ld;
  static constexpr int floats = kc_floats > rc_floats ? kc_floats : rc_floats;
  static constexpr int nvec = ROWS / 32;  // float4 registers per thread per K-step
};

// ---- register -> LDS -------------------------------------------------------------------
// KC: thread t holds, for i < ROWS/32, the float4 at (row = (t>>3) + 32 i, k = 4 (t&7)).
template <int ROWS>
__device__ __forceinline__ void lds_store_kc(float* lds, const f32x4* regs) {
  const int t = threadIdx.x;
  const int c4 = t & 7, r0 = t >> 3;
#pragma unroll
  for (int i = 0; i < ROWS / 32; ++i)
    *reinterpret_cast<f32x4*>(&lds[(r0 + 32 * i) * KC_LD + c4 * 4]) = regs[i];
}
// RC: thread t holds, for i < ROWS/32, the float4 at (k = t/(ROWS/4) + (1024/ROWS) i, row = 4 (t % (ROWS/4))).
template <int ROWS>
__device__ __forceinline__ void lds_store_rc(float* lds, const f32x4* regs) {
  constexpr int F4 = ROWS / 4;
  constexpr int KSTEP = 256 / F4;
  const int t = threadIdx.x;
  const int c4 = t % F4, k0 = t / F4;
#pragma unroll
  for (int i = 0; i < ROWS / 32; ++i)
    *reinterpret_cast<f32x4*>(&lds[(k0 + KSTEP * i) * (ROWS + 4) + c4 * 4]) = regs[i];
}

// ---- global -> register (plain row-major matrices) -------------------------------------
// KC operand: element (row, k) at p[row*ld + k]; rows >= nrows read as zero.
template <int ROWS>
__device__ __forceinline__ void gload_kc(f32x4* regs, const float* __restrict__ p, int ld, int row0,
                                         int nrows, int k0) {
  const int t = threadIdx.x;
  const int c4 = t & 7, r0 = t >> 3;
#pragma unroll
  for (int i = 0; i < ROWS / 32; ++i) {
    const int r = row0 + r0 + 32 * i;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (r < nrows) v = *reinterpret_cast<const f32x4*>(&p[(size_t)r * ld + k0 + c4 * 4]);
    regs[i] = v;
  }
}
// RC operand: element (row, k) at p[k*ld + row]; nrows % 4 == 0; rows >= nrows read as zero.
template <int ROWS>
__device__ __forceinline__ void gload_rc(f32x4* regs, const float* __restrict__ p, int ld, int row0,
                                         int nrows, int k0) {
  constexpr int F4 = ROWS / 4;
  constexpr int KSTEP = 256 / F4;
  const int t = threadIdx.x;
  const int c4 = t % F4, kk = t / F4;
#pragma unroll
  for (int i = 0; i < ROWS / 32; ++i) {
    const int r = row0 + c4 * 4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (r < nrows) v = *reinterpret_cast<const f32x4*>(&p[(size_t)(k0 + kk + KSTEP * i) * ld + r]);
    regs[i] = v;
  }
}

// ---- LDS -> MFMA fragments -------------------------------------------------------------
template <bool KC, int ROWS>
__device__ __forceinline__ f32x4 lds_frag(const float* lds, int row, int t, int h) {
  if constexpr (KC) {
    return *reinterpret_cast<const f32x4*>(&lds[row * KC_LD + 8 * t + 4 * h]);
  } else {
    const float* q = &lds[(8 * t + 4 * h) * (ROWS + 4) + row];
    f32x4 v;
    v[0] = q[0];
    v[1] = q[(ROWS + 4)];
    v[2] = q[2 * (ROWS + 4)];
    v[3] = q[3 * (ROWS + 4)];
    return v;
  }
}

// One K-step (32) of MFMAs for this wave out of the LDS images.
template <int BM, int BN, bool AKC, bool BKC>
__device__ __forceinline__ void mma_kstep(const float* ldsA, const float* ldsB,
                                          f32x16 (&acc)[BM / 64][BN / 64]) {
  constexpr int TM = BM / 64, TN = BN / 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, h = lane >> 5;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    f32x4 a[TM], b[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) a[i] = lds_frag<AKC, BM>(ldsA, wm * (BM / 2) + i * 32 + l31, t, h);
#pragma unroll
    for (int i = 0; i < TN; ++i) b[i] = lds_frag<BKC, BN>(ldsB, wn * (BN / 2) + i * 32 + l31, t, h);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int im = 0; im < TM; ++im)
#pragma unroll
        for (int in = 0; in < TN; ++in)
          acc[im][in] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[im][j], b[in][j], acc[im][in], 0, 0, 0);
  }
}

// Accumulator element -> (row, col) inside the block tile.
// C/D layout of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
template <int BM, int BN>
struct AccMap {
  int row_base, col;  // for tile (0,0), reg 0
  __device__ __forceinline__ AccMap() {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    row_base = (wave >> 1) * (BM / 2) + 4 * (lane >> 5);
    col = (wave & 1) * (BN / 2) + (lane & 31);
  }
  __device__ __forceinline__ int row(int im, int reg) const {
    return row_base + im * 32 + (reg & 3) + 8 * (reg >> 2);
  }
  __device__ __forceinline__ int colof(int in) const { return col + in * 32; }
};

template <int BM, int BN>
__device__ __forceinline__ void acc_zero(f32x16 (&acc)[BM / 64][BN / 64]) {
#pragma unroll
  for (int i = 0; i < BM / 64; ++i)
#pragma unroll
    for (int j = 0; j < BN / 64; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
}
