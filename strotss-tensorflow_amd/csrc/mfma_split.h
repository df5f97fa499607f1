// f32 GEMM cores on the bf16 MFMA by exact 3-way operand splitting ("bf16x6").
//
// gfx950 runs v_mfma_f32_32x32x2_f32 at 1/16 of the bf16 MFMA rate.  Every f32 value splits EXACTLY
// into three bf16 values, x = h + m + l (h = bf16(x), m = bf16(x-h), l = bf16(x-h-m): 3 x 8 mantissa
// bits + signs cover the 24-bit f32 mantissa), so an f32 product is
//     x*y = hh + (hm + mh) + (hl + mm + lh) + [ml + lm + ll],
// where each partial product is EXACT in f32 (8 x 8 bits) and the bracket is below 2^-24 relative.
// The six leading partial products on v_mfma_f32_32x32x16_bf16 with f32 accumulation give
// f32-class accuracy (measured vs an fp64 reference: not worse than the native f32 MFMA path, see
// tests/test_hip_ops.py::test_split_gemm_accuracy) at 6/16 of the f32-MFMA cost.  NPROD = 9 adds the
// bracket.  This is NOT a reduced-precision mode: inputs, outputs and accumulators are f32.
//
// Structure = mfma_pipe.h (double-buffered LDS, one early barrier per K-step, register
// double-buffered fragments, staging pieces between MFMA groups), with
//   * LDS images per operand: 3 bf16 planes of [row][32 k] with 80-byte rows (conflict-free b128);
//   * f32-sourced operands (activations, features) are split at the LDS store (VALU in the MFMA
//     shadow); frozen weights may be pre-split in global memory (three bf16 planes, BPRE = true).
#pragma once
#include "mfma_tile.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define SP_ROW_BYTES 80

template <int BM_, int BN_, bool APRE_, bool BPRE_, int NPROD_, int WM_ = 2, int WN_ = 2>
struct SplitCfg {
  static constexpr int BM = BM_, BN = BN_, WM = WM_, WN = WN_, NT = 64 * WM_ * WN_, NPROD = NPROD_;
  static constexpr bool APRE = APRE_, BPRE = BPRE_;
  static constexpr int TM = BM_ / WM_ / 32, TN = BN_ / WN_ / 32;
  static constexpr int RPP_F32 = NT / 8, RPP_PRE = NT / 4;   // rows staged per pass: f32 8 lanes/row, planes 4 lanes/row
  static constexpr int NA = APRE_ ? BM_ / RPP_PRE : BM_ / RPP_F32;
  static constexpr int NB = BPRE_ ? BN_ / RPP_PRE : BN_ / RPP_F32;
  static constexpr int A_PL = BM_ * SP_ROW_BYTES, B_PL = BN_ * SP_ROW_BYTES;   // bytes per plane
  static constexpr int BUF = 3 * (A_PL + B_PL);
  static constexpr int LDS_BYTES = 2 * BUF;
  static_assert(NA + NB <= 8, "staging pieces must fit before the barrier slot");
};

struct Split3 { bf16x4 h, m, l; };
__device__ __forceinline__ Split3 split3(const f32x4 x) {
  Split3 s;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const __bf16 h = (__bf16)x[k];
    const float r1 = x[k] - (float)h;
    const __bf16 m = (__bf16)r1;
    const float r2 = r1 - (float)m;
    s.h[k] = h; s.m[k] = m; s.l[k] = (__bf16)r2;
  }
  return s;
}

// Loader protocols:
//   f32 operand (A always, B when !BPRE): issue(i) / f32x4 value(i) / advance()   (as in mfma_pipe.h)
//   pre-split B: issue(i, plane) / u32x4 value(i, plane) / advance()  -- 8 bf16 of row (t>>2)+64 i at k = 8 (t&3)
template <class Cfg, class LA, class LB>
__device__ __forceinline__ void split_mainloop(unsigned char* lds, int steps, LA& la, LB& lb,
                                               f32x16 (&acc)[Cfg::TM][Cfg::TN]) {
  constexpr int TM = Cfg::TM, TN = Cfg::TN, NA = Cfg::NA, NB = Cfg::NB;
  constexpr int A_PL = Cfg::A_PL, B_PL = Cfg::B_PL, BUF = Cfg::BUF;
  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int wm = wave / Cfg::WN, wn = wave % Cfg::WN, l31 = lane & 31, hh = lane >> 5;

  auto stage_a = [&](unsigned char* buf, int i) {       // (split +) store row i of A, then reload it
    if constexpr (Cfg::APRE) {
      const int o = ((t >> 2) + Cfg::RPP_PRE * i) * SP_ROW_BYTES + (t & 3) * 16;
#pragma unroll
      for (int p = 0; p < 3; ++p) *reinterpret_cast<u32x4*>(buf + p * A_PL + o) = la.value(i, p);
#pragma unroll
      for (int p = 0; p < 3; ++p) la.issue(i, p);
    } else {
      const Split3 s = split3(la.value(i, 0));
      const int o = ((t >> 3) + Cfg::RPP_F32 * i) * SP_ROW_BYTES + (t & 7) * 8;
      *reinterpret_cast<bf16x4*>(buf + o) = s.h;
      *reinterpret_cast<bf16x4*>(buf + A_PL + o) = s.m;
      *reinterpret_cast<bf16x4*>(buf + 2 * A_PL + o) = s.l;
      la.issue(i, 0);
    }
  };
  auto stage_b = [&](unsigned char* buf, int i) {
    unsigned char* b = buf + 3 * A_PL;
    if constexpr (Cfg::BPRE) {
      const int o = ((t >> 2) + Cfg::RPP_PRE * i) * SP_ROW_BYTES + (t & 3) * 16;
#pragma unroll
      for (int p = 0; p < 3; ++p) *reinterpret_cast<u32x4*>(b + p * B_PL + o) = lb.value(i, p);
#pragma unroll
      for (int p = 0; p < 3; ++p) lb.issue(i, p);
    } else {
      const Split3 s = split3(lb.value(i, 0));
      const int o = ((t >> 3) + Cfg::RPP_F32 * i) * SP_ROW_BYTES + (t & 7) * 8;
      *reinterpret_cast<bf16x4*>(b + o) = s.h;
      *reinterpret_cast<bf16x4*>(b + B_PL + o) = s.m;
      *reinterpret_cast<bf16x4*>(b + 2 * B_PL + o) = s.l;
      lb.issue(i, 0);
    }
  };
  auto issue_all = [&]() {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      if constexpr (Cfg::APRE) {
#pragma unroll
        for (int p = 0; p < 3; ++p) la.issue(i, p);
      } else {
        la.issue(i, 0);
      }
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      if constexpr (Cfg::BPRE) {
#pragma unroll
        for (int p = 0; p < 3; ++p) lb.issue(i, p);
      } else {
        lb.issue(i, 0);
      }
    }
  };
  // prologue: tile 0 -> buffer 0 (stage_* also re-issues: that is tile 1 after advance)
  issue_all();
  la.advance(); lb.advance();
#pragma unroll
  for (int i = 0; i < NA; ++i) stage_a(lds, i);
#pragma unroll
  for (int i = 0; i < NB; ++i) stage_b(lds, i);
  la.advance(); lb.advance();
  __syncthreads();

  const int aoff = (wm * (Cfg::BM / Cfg::WM) + l31) * SP_ROW_BYTES + 16 * hh;
  const int boff = 3 * A_PL + (wn * (Cfg::BN / Cfg::WN) + l31) * SP_ROW_BYTES + 16 * hh;
  bf16x8 fa[2][3][TM], fb[2][3][TN];
  auto read_frags = [&](const unsigned char* buf, int c, int slot) {
#pragma unroll
    for (int p = 0; p < 3; ++p) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
        fa[slot][p][i] = *reinterpret_cast<const bf16x8*>(buf + p * A_PL + aoff + i * 32 * SP_ROW_BYTES + 32 * c);
#pragma unroll
      for (int i = 0; i < TN; ++i)
        fb[slot][p][i] = *reinterpret_cast<const bf16x8*>(buf + p * B_PL + boff + i * 32 * SP_ROW_BYTES + 32 * c);
    }
  };
  read_frags(lds, 0, 0);
  // partial products, leading terms first: (A plane, B plane)
  constexpr int PA[9] = {0, 0, 1, 0, 1, 2, 1, 2, 2};
  constexpr int PB[9] = {0, 1, 0, 2, 1, 0, 2, 1, 2};
  for (int s = 0; s < steps; ++s) {
    unsigned char* cur = lds + (s & 1) * BUF;
    unsigned char* nxt = lds + ((s + 1) & 1) * BUF;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      if (c == 0) read_frags(cur, 1, 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < Cfg::NPROD; ++q) {
#pragma unroll
        for (int im = 0; im < TM; ++im)
#pragma unroll
          for (int in_ = 0; in_ < TN; ++in_)
            acc[im][in_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[c][PA[q]][im], fb[c][PB[q]][in_], acc[im][in_], 0, 0, 0);
        const int slot = c * Cfg::NPROD + q;          // one staging piece per slot
        if (slot < NA + NB) {
          __builtin_amdgcn_sched_barrier(0);
          if (slot < NA) stage_a(nxt, slot); else stage_b(nxt, slot - NA);
          __builtin_amdgcn_sched_barrier(0);
        }
        constexpr int BARRIER_SLOT = (NA + NB - 1 > Cfg::NPROD) ? NA + NB - 1 : Cfg::NPROD;   // in chunk 1, after the last piece
        if (slot == BARRIER_SLOT) {
          // all reads of `cur` are issued (chunk 1's fragments were read during chunk 0) and all writes
          // of `nxt` are issued: barrier, then fetch the next tile's first fragments under the rest
          __builtin_amdgcn_sched_barrier(0);
          __syncthreads();
          read_frags(nxt, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    la.advance(); lb.advance();
  }
}

// Pre-split plane loader for a row-major operand: planes[p] + row*ld + k (bf16), rows >= nrows are zeros.
template <int NR, int RPP = 64>
struct PlaneRowLoader {
  const __bf16* p[3];
  size_t base[NR];
  u32x4 r[NR][3];
  unsigned ok;
  int k0, K;
  __device__ __forceinline__ PlaneRowLoader(const __bf16* planes, size_t plane_stride, int ld, int row0, int nrows, int K_)
      : ok(0), k0(0), K(K_) {
    p[0] = planes; p[1] = planes + plane_stride; p[2] = planes + 2 * plane_stride;
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      const int row = row0 + (threadIdx.x >> 2) + RPP * i;
      ok |= (unsigned)(row < nrows) << i;
      base[i] = (size_t)min(row, nrows - 1) * ld + (threadIdx.x & 3) * 8;
    }
  }
  __device__ __forceinline__ void issue(int i, int pl) { r[i][pl] = *reinterpret_cast<const u32x4*>(p[pl] + base[i] + k0); }
  __device__ __forceinline__ u32x4 value(int i, int pl) const {
    const u32x4 z = {0u, 0u, 0u, 0u};
    return ((ok >> i) & 1u) ? r[i][pl] : z;
  }
  __device__ __forceinline__ void advance() { k0 += 32; if (k0 >= K) k0 = 0; }
};
