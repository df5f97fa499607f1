// Streaming form of the bf16x3 GEMM core (mfma_x3.h) for long lists of equal products -- the 36 transform-domain GEMMs of
// a Winograd layer (nn/model.py:44-48 of the reference: the VGG trunk) -- as ONE persistent workgroup per CU.
//
// What the per-tile form (gemm_x3_kernel<X3CfgK16>) loses: a workgroup lives for one 128 x 128 x K product, 16-32 K-steps;
// its first tiles' latency, its 64 KiB of result stores and its launch are exposed every time, a second co-resident
// workgroup hides part of that at the price of a 3-stage ring (two tiles in flight) and of two waves per SIMD sharing
// the matrix pipe -- measured 42-47 % of the bf16 MFMA's x3 bound.  Here a workgroup walks its list of tiles with ONE
// continuous pipeline:
//   * the LDS ring (NS stages of one K-step of 16: six plane images of [128 rows][32 B]) never drains: the DMA of the
//     NEXT tile's first K-steps is issued during the current tile's last ones (NS - 1 K-steps in flight);
//   * a tile's 64 accumulator registers are copied aside when it ends and leave during the NEXT tile's K-steps, 64 / KS
//     stores per step between the MFMAs; the first tile of a workgroup stores zeros to its own (later overwritten) rows
//     and the prologue does the same, so that EVERY step issues exactly 6 DMA pieces + S stores -- which is what makes the
//     counted `s_waitcnt vmcnt((6 + S)(NS - 2))` exact (loads, LDS-DMA and stores share one in-order counter, 6 bits);
//   * one raw s_barrier per K-step, the next step's fragments fetched behind it under the step's last 8 MFMAs, as in
//     mfma_x3.h (same LDS images, same swizzle, same product order -> bitwise the same results as gemm_x3_kernel).
// Every store is unconditional: rows and columns must be multiples of 128 IN MEMORY (the caller pads C; A / B rows past the
// operand's end are clamped and feed the padding).
#pragma once
#include "mfma_x3.h"

template <int KS_, int NS_ = 6>
struct X3StreamCfg {
  static constexpr int KS = KS_;                       // K-steps of 16 per product: K = 16 KS
  static constexpr int NS = NS_, D = NS_ - 1;          // ring stages, K-steps in flight
  static constexpr int PL = 128 * 32, STAGE = 6 * PL;  // one plane image, one stage (24 KiB)
  static constexpr int LDS_BYTES = NS_ * STAGE;
  static constexpr int S = 64 / KS_;                   // result stores of the previous tile per K-step
  static constexpr int NWAIT = (6 + S) * (D - 1);
  static_assert(64 % KS_ == 0 && KS_ % 2 == 0 && KS_ > D && S <= 10, "K-steps per product");
  static_assert(NWAIT <= 63, "vmcnt has 6 bits");
};

template <int N> __device__ __forceinline__ void x3s_wait_vm() {
  static_assert(N == 32 || N == 40 || N == 56 || N == 24 || N == 30 || N == 42, "add the count here");
  if constexpr (N == 32) X3_WAIT_VM(32); else if constexpr (N == 40) X3_WAIT_VM(40); else if constexpr (N == 56) X3_WAIT_VM(56);
  else if constexpr (N == 24) X3_WAIT_VM(24); else if constexpr (N == 30) X3_WAIT_VM(30); else X3_WAIT_VM(42);
}

// C[z] (Mpad x N, row stride ldc) = A[z] B[z]^T;  A: x3 panels of M rows, B: of N rows (N % 128 == 0), K = 16 KS.
// Tile ids: (z * MT + mt) * NT + nt;  workgroup w takes ids k * gridDim + sw(w), sw = XCD-aware permutation.
template <class Cfg>
__global__ __launch_bounds__(256) void gemm_x3_stream_kernel(const __bf16* __restrict__ A, int M, long long strideA,
                                                             const __bf16* __restrict__ B, int N, long long strideB,
                                                             float* __restrict__ C, int ldc, long long strideC, int MT, int NT,
                                                             int ntiles) {
  __shared__ __attribute__((aligned(1024))) unsigned char lds[Cfg::LDS_BYTES];
  constexpr int KS = Cfg::KS, NS = Cfg::NS, D = Cfg::D, PL = Cfg::PL, STAGE = Cfg::STAGE, S = Cfg::S;
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, hh = lane >> 5;
  const int G = (int)gridDim.x;
  const int first = (G & 7) == 0 ? (int)xcd_swizzle(blockIdx.x, G) : (int)blockIdx.x;
  if (first >= ntiles) return;
  const int nk = (ntiles - first + G - 1) / G;                    // tiles of this workgroup: first + k G
  const int per_z = MT * NT;
  // fragment read offsets within a stage (mfma_x3.h, K16 images)
  const int sw = (hh ^ ((l31 >> 3) & 1)) << 4;
  const int a_rd = (wm * 64 + l31) * 32 + sw;
  const int b_rd = 3 * PL + (wn * 64 + l31) * 32 + sw;
  // result element (im, in, reg) of this lane: row wm 64 + im 32 + 8 (reg >> 2) + 4 hh + (reg & 3), column wn 64 + in 32 + l31
  const int row_base = wm * 64 + 4 * hh, col_base = wn * 64 + l31;
  auto corner = [&](int id, int& z, int& m0, int& n0) {
    z = id / per_z;
    const int rem = id - z * per_z;
    m0 = (rem / NT) * 128; n0 = (rem % NT) * 128;
  };
  constexpr int PA[6] = {2, 0, 1, 1, 0, 0};
  constexpr int PB[6] = {0, 2, 1, 0, 1, 0};

  int z, m0, n0;
  corner(first, z, m0, n0);
  X3OperandK16 oa(A + (long long)z * strideA, M, m0), ob(B + (long long)z * strideB, N, n0);
  int half = 0;                                                    // which 32-byte half of the K-block the NEXT DMA reads
  bool frozen = false;                                             // past the last tile: the look-ahead re-reads its last step
  float* c_prev = C + (long long)z * strideC + (size_t)(m0 + row_base) * ldc + n0 + col_base;
  float* c_cur = c_prev;
  f32x16 acc[2][2], prev[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) { acc[i][j][e] = 0.f; prev[i][j][e] = 0.f; }
  auto dma_piece = [&](int j, unsigned char* stage) {             // j 0..2: A planes, 3..5: B planes; this wave's 32 rows
#ifdef X3_ABL_NO_DMA
    return;
#endif
    unsigned char* dst = stage + j * PL + wave * 1024;
    if (j < 3) oa.dma(j, dst); else ob.dma(j - 3, dst);
  };
  auto dma_advance = [&]() {
    if (!frozen) { oa.advance(half); ob.advance(half); half ^= 1; }
  };
  auto store_prev = [&](int slot) {                                // one of the 64 result registers of the previous tile
    const int im = slot >> 5, in = (slot >> 4) & 1, reg = slot & 15;
    float* p = c_prev + (size_t)(im * 32 + (reg & 3) + 8 * (reg >> 2)) * ldc + in * 32;
#ifndef X3_ABL_NO_STORE
    *p = prev[im][in][reg];
#endif
  };
  bf16x8 fa[2][3][2], fb[2][3][2];
  auto read_frags = [&](const unsigned char* stage, int slot) {
#pragma unroll
    for (int p = 0; p < 3; ++p) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        fa[slot][p][i] = *reinterpret_cast<const bf16x8*>(stage + a_rd + p * PL + i * 1024);
        fb[slot][p][i] = *reinterpret_cast<const bf16x8*>(stage + b_rd + p * PL + i * 1024);
      }
    }
  };

  // prologue: the first D K-steps of the first tile (KS > D), each followed by S (zero) stores like every later step
#pragma unroll
  for (int d = 0; d < D; ++d) {
#pragma unroll
    for (int j = 0; j < 6; ++j) dma_piece(j, lds + d * STAGE);
    dma_advance();
#pragma unroll
    for (int i = 0; i < S; ++i) store_prev(i);
  }
  x3s_wait_vm<Cfg::NWAIT>();
  __builtin_amdgcn_s_barrier();
  read_frags(lds, 0);
  int cur = 0;                                                     // stage of the current K-step
  int pf_steps_left = KS - D;                                      // K-steps of the tile the DMA is on that are not issued yet
  int pf_k = 0;                                                    // index (in this workgroup's list) of that tile

  for (int k = 0; k < nk; ++k) {
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int SLOT = s & 1;
      unsigned char* s_new = lds + (cur == 0 ? NS - 1 : cur - 1) * STAGE;        // free since the previous step's barrier
      const int nxt = cur == NS - 1 ? 0 : cur + 1;
      const unsigned char* s_nxt = lds + nxt * STAGE;
      // the DMA walks one K-step per step; when a tile's steps are all issued it moves to the workgroup's next tile
      if (pf_steps_left == 0) {
        if (pf_k + 1 < nk) {
          ++pf_k;
          int z2, m2, n2;
          corner(first + pf_k * G, z2, m2, n2);
          oa = X3OperandK16(A + (long long)z2 * strideA, M, m2);
          ob = X3OperandK16(B + (long long)z2 * strideB, N, n2);
          half = 0;
          pf_steps_left = KS;
        } else {
          frozen = true;                                           // re-read the last step: in bounds, never consumed
          pf_steps_left = 1 << 30;
          if (half == 0) { /* the pointer already moved past the last step: step back onto it */
            oa.cur -= 3u * oa.plane - 32u; ob.cur -= 3u * ob.plane - 32u;
          } else { oa.cur -= 32u; ob.cur -= 32u; }
        }
      }
#pragma unroll
      for (int q = 0; q < 6; ++q) {
#pragma unroll
        for (int im = 0; im < 2; ++im) {
#pragma unroll
          for (int in = 0; in < 2; ++in) {
#ifndef X3_ABL_NO_MFMA
            acc[im][in] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[SLOT][PA[q]][im], fb[SLOT][PB[q]][in], acc[im][in], 0, 0, 0);
#else
            acc[im][in][0] += (float)fa[SLOT][PA[q]][im][0] + (float)fb[SLOT][PB[q]][in][0];
#endif
            const int done = (q * 2 + im) * 2 + in + 1;
            if (done <= 6) {
              __builtin_amdgcn_sched_barrier(0);
              dma_piece(done - 1, s_new);
              __builtin_amdgcn_sched_barrier(0);
            } else if (done <= 6 + S) {
              __builtin_amdgcn_sched_barrier(0);
              store_prev(s * S + done - 7);
              __builtin_amdgcn_sched_barrier(0);
            }
          }
        }
        if (q == 3) {
          __builtin_amdgcn_sched_barrier(0);
          x3s_wait_vm<Cfg::NWAIT>();                               // the next K-step has landed (stricter by S old stores)
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
          asm volatile("" ::: "memory");
          read_frags(s_nxt, SLOT ^ 1);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      dma_advance();
      --pf_steps_left;
      cur = nxt;
    }
    // the tile is finished: its results leave during the next tile's steps (or in the burst below)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) { prev[i][j][e] = acc[i][j][e]; acc[i][j][e] = 0.f; }
    c_prev = c_cur;
    if (k + 1 < nk) {
      corner(first + (k + 1) * G, z, m0, n0);
      c_cur = C + (long long)z * strideC + (size_t)(m0 + row_base) * ldc + n0 + col_base;
    }
  }
#pragma unroll
  for (int slot = 0; slot < 64; ++slot) store_prev(slot);
}
