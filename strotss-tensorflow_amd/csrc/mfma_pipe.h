// Pipelined f32-MFMA main loop shared by the direct 3x3 conv (conv.hip) and the K-contiguous GEMMs
// (gemm.hip: Winograd-domain GEMMs, cosine cost matrices, moment backward).
//
// Block tile BM x BN, K-step 32, WM x WN waves (each (BM/WM) x (BN/WN) as 32x32 MFMA tiles).
//   * double-buffered LDS ([row][36] images, conflict-free ds_read_b128 / ds_write_b128);
//   * per K-step ONE barrier, placed before the last 8 MFMAs of the step: by then every wave has
//     issued all its reads of the current buffer and all its writes of the next one, so the first
//     fragments of the next tile are fetched under the remaining MFMAs;
//   * MFMA fragments are register double-buffered (k-group t+1 is read before group t's MFMAs);
//   * the staging work of a step (LDS writes of tile s+1, global loads of tile s+2) is cut into 4
//     pieces pinned between the MFMA groups with sched_barrier: it runs in the shadow of the
//     64-cycle f32 MFMAs.  Loads are branch-free (clamped address); rows/taps outside the operand
//     are zeroed at the LDS store one step later, so no load is waited for at issue.
//
// Global loads run TWO K-steps ahead of the LDS store that consumes them (two register slots; measured:
// removing the loads altogether is worth 18 %, i.e. their latency was exposed at distance one).
// Loader protocol (functor L, owns its staging registers, two slots):
//   L.issue(i, slot) : issue the raw 16-byte load of staged row i of the NEXT tile into `slot`;
//   L.value(i, slot) : the f32x4 to put in LDS for row i of the tile held in `slot` (this is where
//                      out-of-range rows/taps become zeros);
//   L.advance()      : move to the following tile; loads past the last tile must stay in bounds
//                      (wrap) -- their data is never used.
#pragma once
#include <type_traits>

#include "mfma_tile.h"

// XCD-aware workgroup id (guide T1): hardware deals consecutive workgroup ids round-robin over the 8
// XCDs (each with a private 4 MiB L2).  Remap so that every XCD owns a CONTIGUOUS range of the logical
// tile order -- neighbouring tiles (same A panel, adjacent B panels, halo rows) then share an L2.
// Bijective for any workgroup count.  Speed only: results do not depend on placement.
__device__ __forceinline__ unsigned xcd_swizzle(unsigned id, unsigned n) {
  const unsigned xcd = id & 7u, slot = id >> 3, q = n >> 3, r = n & 7u;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
}

template <int BM_, int BN_, int WM_, int WN_>
struct PipeCfg {
  static constexpr int BM = BM_, BN = BN_, WM = WM_, WN = WN_;
  static constexpr int NT = 64 * WM_ * WN_;          // threads
  static constexpr int RPP = NT / 8;                 // rows staged per pass (8 lanes x 16 B per row)
  static constexpr int NA = BM_ / RPP, NB = BN_ / RPP;
  static constexpr int TM = BM_ / WM_ / 32, TN = BN_ / WN_ / 32;
  static constexpr int A_FL = BM_ * KC_LD, B_FL = BN_ * KC_LD;
  static constexpr int LDS_FLOATS = 2 * (A_FL + B_FL);
  static_assert(NA >= 1 && NB >= 1 && TM >= 1 && TN >= 1, "tile too small for this wave grid");
};

template <class Cfg>
struct PipeAccMap {   // accumulator element -> (row, col) in the block tile (32x32 MFMA C/D layout)
  int row_base, col;
  __device__ __forceinline__ PipeAccMap() {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    row_base = (wave / Cfg::WN) * (Cfg::BM / Cfg::WM) + 4 * (lane >> 5);
    col = (wave % Cfg::WN) * (Cfg::BN / Cfg::WN) + (lane & 31);
  }
  __device__ __forceinline__ int row(int im, int reg) const { return row_base + im * 32 + (reg & 3) + 8 * (reg >> 2); }
  __device__ __forceinline__ int colof(int in) const { return col + in * 32; }
};

template <class Cfg, class LA, class LB>
__device__ __forceinline__ void pipe_mainloop(float* lds, int steps, LA& la, LB& lb,
                                              f32x16 (&acc)[Cfg::TM][Cfg::TN]) {
  constexpr int TM = Cfg::TM, TN = Cfg::TN, NA = Cfg::NA, NB = Cfg::NB, A_FL = Cfg::A_FL, B_FL = Cfg::B_FL;
  const int t = threadIdx.x, c4 = t & 7, r0 = t >> 3;
  const int lane = t & 63, wave = t >> 6;
  const int wm = wave / Cfg::WN, wn = wave % Cfg::WN, l31 = lane & 31, hh = lane >> 5;
  auto store_a = [&](float* buf, int i, int slot) {
    *reinterpret_cast<f32x4*>(&buf[(r0 + Cfg::RPP * i) * KC_LD + c4 * 4]) = la.value(i, slot);
  };
  auto store_b = [&](float* buf, int i, int slot) {
    *reinterpret_cast<f32x4*>(&buf[A_FL + (r0 + Cfg::RPP * i) * KC_LD + c4 * 4]) = lb.value(i, slot);
  };
  auto issue_tile = [&](int slot) {
#pragma unroll
    for (int i = 0; i < NA; ++i) la.issue(i, slot);
#pragma unroll
    for (int i = 0; i < NB; ++i) lb.issue(i, slot);
    la.advance(); lb.advance();
  };
  // prologue: tile 0 -> LDS buffer 0; tile 1 -> register slot 1; tile 2 -> register slot 0.
  // (the global loads run TWO K-steps ahead of the LDS store that consumes them)
  issue_tile(0);
#pragma unroll
  for (int i = 0; i < NA; ++i) store_a(lds, i, 0);
#pragma unroll
  for (int i = 0; i < NB; ++i) store_b(lds, i, 0);
  issue_tile(1);
  issue_tile(0);
  __syncthreads();

  const int arow = (wm * (Cfg::BM / Cfg::WM) + l31) * KC_LD + 4 * hh;
  const int brow = A_FL + (wn * (Cfg::BN / Cfg::WN) + l31) * KC_LD + 4 * hh;
  f32x4 fa[2][TM], fb[2][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) fa[0][i] = *reinterpret_cast<const f32x4*>(&lds[arow + i * 32 * KC_LD]);
#pragma unroll
  for (int i = 0; i < TN; ++i) fb[0][i] = *reinterpret_cast<const f32x4*>(&lds[brow + i * 32 * KC_LD]);

  // one K-step: MFMAs on `cur`; tile s+1 (register slot SLOT) -> `nxt`; tile s+3 -> register slot SLOT
  auto kstep = [&](const float* cur, float* nxt, auto slot_tag) {
    constexpr int SLOT = decltype(slot_tag)::value;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      if (g < 3) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
          fa[(g + 1) & 1][i] = *reinterpret_cast<const f32x4*>(&cur[arow + i * 32 * KC_LD + 8 * (g + 1)]);
#pragma unroll
        for (int i = 0; i < TN; ++i)
          fb[(g + 1) & 1][i] = *reinterpret_cast<const f32x4*>(&cur[brow + i * 32 * KC_LD + 8 * (g + 1)]);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int im = 0; im < TM; ++im)
#pragma unroll
          for (int in_ = 0; in_ < TN; ++in_)
            acc[im][in_] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[g & 1][im][j], fb[g & 1][in_][j], acc[im][in_], 0, 0, 0);
        if (j == 0) {
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int i = g; i < NA; i += 4) { store_a(nxt, i, SLOT); la.issue(i, SLOT); }
          __builtin_amdgcn_sched_barrier(0);
        }
        if (j == 1) {
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int i = g; i < NB; i += 4) { store_b(nxt, i, SLOT); lb.issue(i, SLOT); }
          __builtin_amdgcn_sched_barrier(0);
          if (g == 3) {
            __syncthreads();
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[0][i] = *reinterpret_cast<const f32x4*>(&nxt[arow + i * 32 * KC_LD]);
#pragma unroll
            for (int i = 0; i < TN; ++i) fb[0][i] = *reinterpret_cast<const f32x4*>(&nxt[brow + i * 32 * KC_LD]);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
    }
    la.advance(); lb.advance();
  };
  float* buf0 = lds;
  float* buf1 = lds + (A_FL + B_FL);
  int s = 0;
  for (; s + 1 < steps; s += 2) {
    kstep(buf0, buf1, std::integral_constant<int, 1>{});
    kstep(buf1, buf0, std::integral_constant<int, 0>{});
  }
  if (s < steps) kstep(buf0, buf1, std::integral_constant<int, 1>{});
}

// Plain row-major K-contiguous operand: element (row, k) at p[row*ld + k]; rows >= nrows are zeros.
template <class Cfg, int NROWS_STAGED>
struct RowMajorLoader {
  const float* p;
  size_t base[NROWS_STAGED];
  f32x4 r[2][NROWS_STAGED];
  unsigned ok;
  int k0, K;
  __device__ __forceinline__ RowMajorLoader(const float* ptr, int ld, int row0, int nrows, int K_) : p(ptr), ok(0), k0(0), K(K_) {
    const int c4 = threadIdx.x & 7, r0 = threadIdx.x >> 3;
#pragma unroll
    for (int i = 0; i < NROWS_STAGED; ++i) {
      const int row = row0 + r0 + Cfg::RPP * i;
      ok |= (unsigned)(row < nrows) << i;
      base[i] = (size_t)min(row, nrows - 1) * ld + c4 * 4;
    }
  }
  __device__ __forceinline__ void issue(int i, int slot) { r[slot][i] = *reinterpret_cast<const f32x4*>(p + base[i] + k0); }
  __device__ __forceinline__ f32x4 value(int i, int slot) const {
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    return ((ok >> i) & 1u) ? r[slot][i] : z4;
  }
  __device__ __forceinline__ void advance() { k0 += 32; if (k0 >= K) k0 = 0; }
};
