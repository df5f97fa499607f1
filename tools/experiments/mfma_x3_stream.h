// Streaming form of the bf16x3 GEMM core (mfma_x3.h) for long lists of equal products -- the 36 transform-domain GEMMs of
// a Winograd layer (nn/model.py:44-48 of the reference: the VGG trunk) -- as ONE persistent workgroup per CU.
//
// What the per-tile form (gemm_x3_kernel<X3CfgK16>) loses: a workgroup lives for one 128 x 128 x K product, 16-32 K-steps;
// its first tiles' latency, its 64 KiB of result stores and its launch are exposed every time, a second co-resident
// workgroup hides part of that at the price of a 3-stage ring (two tiles in flight) and of two waves per SIMD sharing
// the matrix pipe -- measured 42-47 % of the bf16 MFMA's x3 bound.  Here a workgroup of FIVE waves walks its list of
// tiles with one continuous pipeline:
//   * waves 0-3 compute (2 x 2, a 64 x 64 sub-tile each): the LDS ring (NS stages of one K-step of 16: six plane images
//     of [128 rows][32 B]) never drains -- the DMA of the NEXT tile's first K-steps is issued during the current tile's
//     last ones (NS - 1 K-steps in flight); one raw s_barrier per K-step, the next step's fragments fetched behind it
//     under the step's last 8 MFMAs; same LDS images, swizzle and product order as mfma_x3.h (bitwise the same results);
//   * their ONLY vector-memory operations are the LDS-DMA pieces, so `s_waitcnt vmcnt(6 (NS - 2))` is exact.  Round 3's
//     first form stored the previous tile's results from the computing waves, between the MFMAs: loads, LDS-DMA and
//     stores share ONE in-order counter, every wait for a DMA tile also waited for the stores in front of it, and the
//     block3 GEMMs took 138 us against 82 us with the stores compiled out (per-tile form: 125 us);
//   * a finished tile goes accumulators -> LDS (64 KiB tile image, row-major) and wave 4, the STORE wave, moves it to
//     memory during the next tile's K-steps, a few 16-byte rows per step (512 contiguous bytes per row and half wave); it
//     takes part in every barrier and owns its own vmcnt.
// Rows and columns must be multiples of 128 IN MEMORY (the caller pads C; A / B rows past the operand's end are clamped
// and feed the padding).
#pragma once
#include "mfma_x3.h"

template <int KS_, int NS_ = 4>
struct X3StreamCfg {
  static constexpr int KS = KS_;                       // K-steps of 16 per product: K = 16 KS
  static constexpr int NS = NS_, D = NS_ - 1;          // ring stages, K-steps in flight
  static constexpr int PL = 128 * 32, STAGE = 6 * PL;  // one plane image, one stage (24 KiB)
  static constexpr int RING = NS_ * STAGE;
  static constexpr int TILE = 128 * 128 * 4;           // the finished tile's image for the store wave (64 KiB)
  static constexpr int LDS_BYTES = RING + TILE;
  static constexpr int NT = 320;                       // four computing waves + the store wave
  static constexpr int CH = (64 + KS_ - 2) / (KS_ - 1);  // two-row chunks the store wave moves per K-step (all but the last)
  static constexpr int NWAIT = 6 * (D - 1);
  static_assert(KS_ % 2 == 0 && KS_ > D && KS_ >= 8, "K-steps per product");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS");
};

template <int N> __device__ __forceinline__ void x3s_wait_vm() {
  static_assert(N == 6 || N == 12 || N == 18, "add the count here");
  if constexpr (N == 6) X3_WAIT_VM(6); else if constexpr (N == 12) X3_WAIT_VM(12); else X3_WAIT_VM(18);
}

// C[z] (Mpad x N, row stride ldc) = A[z] B[z]^T;  A: x3 panels of M rows, B: of N rows (N % 128 == 0), K = 16 KS.
// Tile ids: (z * MT + mt) * NT + nt;  workgroup w takes ids k * gridDim + sw(w), sw = XCD-aware permutation.
template <class Cfg>
__global__ __launch_bounds__(Cfg::NT) void gemm_x3_stream_kernel(const __bf16* __restrict__ A, int M, long long strideA,
                                                                 const __bf16* __restrict__ B, int N, long long strideB,
                                                                 float* __restrict__ C, int ldc, long long strideC, int MT,
                                                                 int NT, int ntiles) {
  __shared__ __attribute__((aligned(1024))) unsigned char lds[Cfg::LDS_BYTES];
  constexpr int KS = Cfg::KS, NS = Cfg::NS, D = Cfg::D, PL = Cfg::PL, STAGE = Cfg::STAGE, CH = Cfg::CH;
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int G = (int)gridDim.x;
  const int first = (G & 7) == 0 ? (int)xcd_swizzle(blockIdx.x, G) : (int)blockIdx.x;
  if (first >= ntiles) return;
  const int nk = (ntiles - first + G - 1) / G;                    // tiles of this workgroup: first + k G
  const int per_z = MT * NT;
  unsigned char* tilebuf = lds + Cfg::RING;
  auto corner = [&](int id, int& z, int& m0, int& n0) {
    z = id / per_z;
    const int rem = id - z * per_z;
    m0 = (rem / NT) * 128; n0 = (rem % NT) * 128;
  };

  if (wave == 4) {
    // ------------------------------------------------------------------ the store wave
    // two-row chunk c of the tile image: lane l moves 16 bytes of row 2 c + (l >> 5): 512 contiguous bytes per row
    const int lrow = lane >> 5, lcol = (lane & 31) * 4;
    auto move = [&](int c0, int c1, float* corner_ptr) {
      for (int base = c0; base < c1; base += 8) {
        f32x4 v[8];
        const int n = min(8, c1 - base);
#pragma unroll
        for (int i = 0; i < 8; ++i)
          if (i < n) v[i] = *reinterpret_cast<const f32x4*>(tilebuf + (size_t)(2 * (base + i) + lrow) * 512 + lcol * 4);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < 8; ++i)
          if (i < n) *reinterpret_cast<f32x4*>(corner_ptr + (size_t)(2 * (base + i) + lrow) * ldc + lcol) = v[i];
      }
    };
    __builtin_amdgcn_s_barrier();                                  // the prologue's barrier
    float* c_prev = nullptr;
    for (int k = 0; k < nk; ++k) {
      int z, m0, n0;
      corner(first + k * G, z, m0, n0);
      float* c_cur = C + (long long)z * strideC + (size_t)m0 * ldc + n0;
      for (int s = 0; s < KS; ++s) {
        __builtin_amdgcn_s_barrier();                              // the K-step's barrier
        if (k > 0 && s < KS - 1) {                                 // tile k - 1 lies in the image since this tile's first barrier
          const int c0 = s * CH, c1 = min(64, c0 + CH);
          if (c0 < 64) move(c0, c1, c_prev);
        }
      }
      c_prev = c_cur;
    }
    __builtin_amdgcn_s_barrier();                                  // the last tile has been written to the image
    move(0, 64, c_prev);
    return;
  }

  // -------------------------------------------------------------------- the four computing waves
  const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, hh = lane >> 5;
  // fragment read offsets within a stage (mfma_x3.h, K16 images)
  const int sw = (hh ^ ((l31 >> 3) & 1)) << 4;
  const int a_rd = (wm * 64 + l31) * 32 + sw;
  const int b_rd = 3 * PL + (wn * 64 + l31) * 32 + sw;
  // result element (im, in, reg) of this lane: row wm 64 + im 32 + 8 (reg >> 2) + 4 hh + (reg & 3), column wn 64 + in 32 + l31
  unsigned char* my_tile = tilebuf + (size_t)(wm * 64 + 4 * hh) * 512 + (wn * 64 + l31) * 4;
  constexpr int PA[6] = {2, 0, 1, 1, 0, 0};
  constexpr int PB[6] = {0, 2, 1, 0, 1, 0};

  int z, m0, n0;
  corner(first, z, m0, n0);
  X3OperandK16 oa(A + (long long)z * strideA, M, m0), ob(B + (long long)z * strideB, N, n0);
  int half = 0;                                                    // which 32-byte half of the K-block the NEXT DMA reads
  bool frozen = false;                                             // past the last tile: the look-ahead re-reads its last step
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
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

  // prologue: the first D K-steps of the first tile (KS > D)
#pragma unroll
  for (int d = 0; d < D; ++d) {
#pragma unroll
    for (int j = 0; j < 6; ++j) dma_piece(j, lds + d * STAGE);
    dma_advance();
  }
  x3s_wait_vm<Cfg::NWAIT>();
  __builtin_amdgcn_s_barrier();
  read_frags(lds, 0);
  int cur = 0;                                                     // stage of the current K-step
  int pf_steps_left = KS - D;                                      // K-steps of the tile the DMA is on that are not issued yet
  int pf_k = 0;                                                    // index (in this workgroup's list) of that tile

  auto kstep = [&](auto slot_tag) {
    constexpr int SLOT = decltype(slot_tag)::value;
    unsigned char* s_new = lds + (cur == 0 ? NS - 1 : cur - 1) * STAGE;          // free since the previous step's barrier
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
        frozen = true;                                             // re-read the last step: in bounds, never consumed
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
          }
        }
      }
      if (q == 3) {
        __builtin_amdgcn_sched_barrier(0);
        x3s_wait_vm<Cfg::NWAIT>();                                 // the next K-step has landed, D - 1 stay in flight
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
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  for (int k = 0; k < nk; ++k) {
    for (int s = 0; s < KS; s += 2) { kstep(I0{}); kstep(I1{}); }
    // the tile is finished: accumulators -> the tile image (the store wave has read the previous one: its last rows went
    // in this tile's step KS - 2, before the barrier of step KS - 1); visible to it after the next barrier
#pragma unroll
    for (int im = 0; im < 2; ++im)
#pragma unroll
      for (int in = 0; in < 2; ++in)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
#ifndef X3_ABL_NO_STORE
          *reinterpret_cast<float*>(my_tile + (size_t)(im * 32 + (reg & 3) + 8 * (reg >> 2)) * 512 + in * 128) = acc[im][in][reg];
#endif
          acc[im][in][reg] = 0.f;
        }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                                    // hands the last tile to the store wave
}
