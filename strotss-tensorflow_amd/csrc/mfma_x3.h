// f32 GEMM cores on the bf16 MFMA by exact 3-way operand splitting ("bf16x3 operands, 6 products").
//
// gfx950 runs v_mfma_f32_32x32x2_f32 at 1/16 of the bf16 MFMA rate.  Every f32 value splits EXACTLY into
// three bf16 values, x = h + m + l (h = bf16(x), m = bf16(x-h), l = bf16(x-h-m): 3 x 8 mantissa bits +
// signs cover the 24-bit f32 mantissa), so an f32 product is
//     x*y = hh + (hm + mh) + (hl + mm + lh) + [ml + lm + ll],
// where each partial product is EXACT in f32 (8 x 8 bits) and the bracket is below 2^-24 relative.  The
// six leading partial products on v_mfma_f32_32x32x16_bf16 with f32 accumulation give f32-class accuracy
// (tests/test_hip_ops.py::test_x3_gemm_accuracy: error against fp64 at the native f32-MFMA path's level)
// at 6/16 of the f32-MFMA cost.  This is NOT a reduced-precision mode: inputs, outputs and accumulators
// are f32; the split is a change of number representation, not a rounding.
//
// Operand format ("x3 panels"): both operands arrive PRE-SPLIT from their producers (the feature-row norm pass, the
// centring / transposing pass, the epilogue of a preceding GEMM, the Winograd input transform and weight pack), K-blocked so that what a workgroup stages per K-step
// is contiguous:
//     element (kb, plane, row, k)  at  base + ((kb * 3 + plane) * rows + row) * 32 + k      (bf16)
// with kb = K-block of 32, plane in {h, m, l} (an operand whose values are exact in bf16 may carry the h plane only:
// element (kb, row, k) at base + (kb * rows + row) * 32 + k).  A 128-row tile of one plane and one K-block is 8 KiB of
// consecutive bytes: every staging load instruction of a wave reads 1 KiB contiguous, there is no split
// arithmetic, no zero-select and no address arithmetic in the main loop (rows past the operand's end are
// clamped at set-up; their products land in accumulator rows the epilogue masks).
//
// Block tile 128x128 (or 64x64), 256 threads = 2x2 waves (64x64 | 32x32 per wave as 32x32 MFMA tiles), K-step 32
// = 2 MFMA k-chunks x 6 products x 4 | 1 tiles = 48 | 12 MFMAs per wave (see the main loop below).
#pragma once
#include <type_traits>

#include "mfma_pipe.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

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

// x3 panel addressing (elements): see the header comment.
__host__ __device__ __forceinline__ size_t x3_panel_elems(size_t rows, size_t K) { return 3 * rows * K; }
// Stores the split of 4 consecutive k (k0 % 4 == 0) of one row.
__device__ __forceinline__ void x3_store4(__bf16* base, size_t rows, size_t row, int k0, const f32x4 v) {
  const Split3 s = split3(v);
  const size_t o = ((size_t)(k0 >> 5) * 3 * rows + row) * 32 + (k0 & 31);
  *reinterpret_cast<bf16x4*>(base + o) = s.h;
  *reinterpret_cast<bf16x4*>(base + o + rows * 32) = s.m;
  *reinterpret_cast<bf16x4*>(base + o + 2 * rows * 32) = s.l;
}

// Main loop: operands go global -> LDS by LDS-DMA (global_load_lds_dwordx4), three-stage ring.
// LDS stage = 6 images (A h,m,l; B h,m,l) of [128 rows][64 B] (one K-block of 32 bf16 per row), UNPADDED
// because an LDS-DMA wave instruction writes 1 KiB linearly (16 rows); bank conflicts of the fragment reads
// are removed by an XOR swizzle of the four 16-byte slots of a row, slot' = slot ^ ((row >> 2) & 3), applied
// on the SOURCE address of the DMA and on the ds_read address (same involution on both sides).
// Per K-step s: MFMAs on stage s % 3, the DMA of tile s+2 issued between the MFMAs of the first k-chunk into
// stage (s+2) % 3 (free since the barrier of step s-1), one `s_waitcnt vmcnt(12)` (tile s+1 landed, tile
// s+2 stays in flight ACROSS the barrier) + lgkmcnt(0) + raw s_barrier in the second k-chunk, then the first
// fragments of tile s+1.  No staging registers, no ds_write, no VALU in the loop besides 4 address adds.
// NPB_ = planes of the B operand: 3 (general f32 values) or 1 (values exact in bf16, e.g. a sign matrix: B = h, its
// m and l planes would be zero -- three partial products instead of six, a third of the B bytes).
template <int B_, int NPB_ = 3>
struct X3Cfg {               // square block tile B x B, B = 128 (64 x 64 per wave) or 64 (32 x 32 per wave)
  static_assert(B_ == 128 || B_ == 64, "x3 main loop: 128 x 128 or 64 x 64 tiles");
  static_assert(NPB_ == 3 || (NPB_ == 1 && B_ == 128), "single-plane B: 128 x 128 tiles only");
  static constexpr int NPB = NPB_, NPROD = NPB_ == 3 ? 6 : 3;
  static constexpr int BM = B_, BN = B_, WM = 2, WN = 2, NT = 256;
  static constexpr int T = B_ / 64;                      // 32x32 MFMA tiles per wave and dimension
  static constexpr int TM = T, TN = T;
  static constexpr int PL = B_ * 64;                     // bytes of one plane image ([B rows][64 B])
  static constexpr int STAGE = (3 + NPB_) * PL;          // 48 KiB | 24 KiB (three B planes)
  static constexpr int LDS_BYTES = 3 * STAGE;            // 144 KiB (one workgroup per CU) | 72 KiB (two)
  static constexpr int G = B_ / 64;                      // 16-row DMA groups per wave, operand and plane
  static constexpr int NP = (3 + NPB_) * G;              // DMA pieces (1 KiB wave instructions) per wave and K-step
  static constexpr bool K16 = false;
};

#define X3_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

template <int G, int NPL = 3>
struct X3Operand {          // one x3 panel operand (NPL planes per K-block) as seen by one wave's DMA lanes
  const char* base;          // batch base, bytes
  unsigned off[G];           // byte offset of this lane's 16-byte source chunk within a plane panel, per row group
  unsigned plane;            // bytes per plane panel (rows * 64)
  unsigned cur;              // byte offset of the current K-block (3 planes per K-block)
  __device__ __forceinline__ X3Operand(const __bf16* p, int rows, int row0) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    base = reinterpret_cast<const char*>(p);
    plane = (unsigned)rows * 64u;
    cur = 0;
    const int chunk = (lane & 3) ^ ((lane >> 4) & 3);      // source-side swizzle: (row >> 2) & 3 == (lane >> 4) & 3
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const int row = min(row0 + (G * wave + g) * 16 + (lane >> 2), rows - 1);
      off[g] = (unsigned)row * 64u + (unsigned)chunk * 16u;
    }
  }
  __device__ __forceinline__ void dma(int pl, int g, unsigned char* lds_dst) const {
    const char* src = base + (size_t)(cur + pl * plane) + off[g];
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
  }
  __device__ __forceinline__ void advance() { cur += NPL * plane; }
};

template <class Cfg>
__device__ __forceinline__ void x3_mainloop(unsigned char* lds, int steps, X3Operand<Cfg::G, 3>& oa,
                                            X3Operand<Cfg::G, Cfg::NPB>& ob, f32x16 (&acc)[Cfg::T][Cfg::T]) {
  constexpr int PL = Cfg::PL, STAGE = Cfg::STAGE, T = Cfg::T, G = Cfg::G, NP = Cfg::NP, HALF = Cfg::BM / 2;
  constexpr int NPB = Cfg::NPB, NPROD = Cfg::NPROD;
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, hh = lane >> 5;
  // DMA piece j of a K-step (NP per wave): A planes first (3 G pieces), then B's; row group j % G
  auto dma_piece = [&](int j, unsigned char* stage) {
#ifdef X3_ABL_NO_DMA                                  // tools/x3_gemm_ablate.hip: time the loop without its loads
    return;
#endif
    const int pl = j / G, g = j % G;                 // pl 0..2: A planes, 3..: B planes
    unsigned char* dst = stage + pl * PL + (G * wave + g) * 1024;
    if (pl < 3) oa.dma(pl, g, dst); else ob.dma(pl - 3, g, dst);
  };
  auto dma_tile = [&](unsigned char* stage) {
#pragma unroll
    for (int j = 0; j < NP; ++j) dma_piece(j, stage);
    oa.advance(); ob.advance();
  };
  auto wait_one_tile_in_flight = [&]() {
    if constexpr (NP == 12) X3_WAIT_VM(12); else if constexpr (NP == 8) X3_WAIT_VM(8); else X3_WAIT_VM(6);
  };
  // fragment read offsets within a stage: row-major 64-byte rows, slot = (2 kc + hh) ^ ((row >> 2) & 3)
  const int f = (l31 >> 2) & 3;
  const int a_rd = (wm * HALF + l31) * 64 + ((hh ^ f) << 4);              // kc = 0; kc = 1 is this ^ 32
  const int b_rd = 3 * PL + (wn * HALF + l31) * 64 + ((hh ^ f) << 4);
  bf16x8 fa[2][3][T], fb[2][NPB][T];
  auto read_frags = [&](const unsigned char* stage, int kc, int slot) {
#ifdef X3_ABL_NO_FRAG
    if (stage != lds) return;
#endif
    const unsigned char* pa = stage + (a_rd ^ (kc << 5));
    const unsigned char* pb = stage + (b_rd ^ (kc << 5));
#pragma unroll
    for (int p = 0; p < 3; ++p) {
#pragma unroll
      for (int i = 0; i < T; ++i) {
        fa[slot][p][i] = *reinterpret_cast<const bf16x8*>(pa + p * PL + i * 2048);
        if (p < NPB) fb[slot][p][i] = *reinterpret_cast<const bf16x8*>(pb + p * PL + i * 2048);
      }
    }
  };
  // partial products (A plane, B plane), smallest terms first; single-plane B: (l, h), (m, h), (h, h)
  constexpr int PA[6] = {2, NPB == 3 ? 0 : 1, NPB == 3 ? 1 : 0, 1, 0, 0};
  constexpr int PB[6] = {0, NPB == 3 ? 2 : 0, NPB == 3 ? 1 : 0, 0, 1, 0};
  constexpr int STRIDE = NPROD * T * T / NP;      // one DMA piece behind every STRIDE MFMAs of the first k-chunk
  static_assert(STRIDE >= 1, "more DMA pieces than MFMAs in a k-chunk");

  unsigned char* s_cur = lds;
  unsigned char* s_nxt = lds + STAGE;
  unsigned char* s_nn = lds + 2 * STAGE;
  dma_tile(s_cur);
  if (steps > 1) { dma_tile(s_nxt); wait_one_tile_in_flight(); } else { X3_WAIT_VM(0); }
  __builtin_amdgcn_s_barrier();
  read_frags(s_cur, 0, 0);

  // MODE 2: tile s+2 exists (DMA it, leave it in flight); 1: tile s+1 is the last (wait for it); 0: last step
  auto kstep = [&](auto mode_tag) {
    constexpr int MODE = decltype(mode_tag)::value;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
#pragma unroll
      for (int q = 0; q < NPROD; ++q) {
        if (c == 0 && q == 1) {          // second k-chunk's fragments: behind the first product group
          __builtin_amdgcn_sched_barrier(0);
          read_frags(s_cur, 1, 1);
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int im = 0; im < T; ++im) {
#pragma unroll
          for (int in_ = 0; in_ < T; ++in_) {
#ifndef X3_ABL_NO_MFMA
            acc[im][in_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[c][PA[q]][im], fb[c][PB[q]][in_], acc[im][in_], 0, 0, 0);
#else
            acc[im][in_][0] += (float)fa[c][PA[q]][im][0] + (float)fb[c][PB[q]][in_][0];
#endif
            const int done = (q * T + im) * T + in_ + 1;       // MFMAs issued in this k-chunk
            if (MODE == 2 && c == 0 && done % STRIDE == 0 && done / STRIDE <= NP) {
              __builtin_amdgcn_sched_barrier(0);
              dma_piece(done / STRIDE - 1, s_nn);
              __builtin_amdgcn_sched_barrier(0);
            }
          }
        }
        if (c == 1 && q == 0 && MODE != 0) {
          __builtin_amdgcn_sched_barrier(0);
          if (MODE == 2) wait_one_tile_in_flight(); else X3_WAIT_VM(0);
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifndef X3_ABL_NO_BARRIER
          __builtin_amdgcn_s_barrier();
#endif
          asm volatile("" ::: "memory");
          read_frags(s_nxt, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    if (MODE == 2) { oa.advance(); ob.advance(); }
    unsigned char* tmp = s_cur; s_cur = s_nxt; s_nxt = s_nn; s_nn = tmp;
  };
  int s = 0;
  for (; s + 2 < steps; ++s) kstep(std::integral_constant<int, 2>{});
  if (steps > 1) kstep(std::integral_constant<int, 1>{});
  kstep(std::integral_constant<int, 0>{});
}

// ---------------------------------------------------------------------------------------------------------------------
// 128 x 128 tiles, TWO workgroups per CU: K-step 16 (ONE MFMA k-chunk), stage = 6 images of [128 rows][32 B] = 24 KiB,
// three-stage ring = 72 KiB.  The 32-KiB-step form above holds 144 KiB and runs alone on its CU: its prologue (first
// tiles from L2 / HBM), its epilogue (64 KiB of stores) and every barrier are exposed -- tools/x3_gemm_ablate.hip: the MFMA
// stream alone, no loads at all, takes 79 of the 92 us of a 36 x (1024 x 512 x 512) product whose MFMAs are 46 us of work.
// With two co-resident workgroups (two waves per SIMD) one computes while the other fills, stores or waits; the operand
// bytes per MFMA stay those of the 128 tile (half the 64 x 64 form's).
// A DMA wave instruction (64 lanes x 16 B) covers 32 rows x 32 B: wave w stages rows [32 w, 32 w + 32) of each image.
// Rows are 32 B; the two 16-byte slots of a row are swapped in rows with (row >> 3) & 1 (on the DMA's source address and
// on the fragment read), so 16 consecutive lanes of a ds_read_b128 (16 rows, one k-half) cover all 64 banks once.
// Per step s: the 6 DMA pieces of tile s+2 behind the first MFMAs (its stage is free since the barrier of step s-1),
// MFMAs 0..15 on the fragments of tile s, `s_waitcnt vmcnt(6)` (tile s+1 landed, tile s+2 stays in flight) + barrier,
// the fragments of tile s+1 into the other register slot, MFMAs 16..23.
template <int NPB_ = 3>
struct X3CfgK16 {
  static_assert(NPB_ == 3, "K16 ring: three B planes");
  static constexpr int NPB = NPB_, NPROD = 6;
  static constexpr int BM = 128, BN = 128, WM = 2, WN = 2, NT = 256;
  static constexpr int T = 2, TM = 2, TN = 2;
  static constexpr int PL = 128 * 32;                    // bytes of one plane image ([128 rows][32 B])
  static constexpr int STAGE = 6 * PL;                   // 24 KiB
  static constexpr int LDS_BYTES = 3 * STAGE;            // 72 KiB: two workgroups per CU
  static constexpr int G = 1, NP = 6;
  static constexpr bool K16 = true;
};

struct X3OperandK16 {
  const char* base;
  unsigned off;              // this lane's 16-byte chunk of its row: row * 64 + (lane & 1) * 16
  unsigned plane;            // bytes per plane panel (rows * 64)
  unsigned cur;              // byte offset of the current K-block's first plane (+ 32 in its second half)
  __device__ __forceinline__ X3OperandK16(const __bf16* p, int rows, int row0) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    base = reinterpret_cast<const char*>(p);
    plane = (unsigned)rows * 64u;
    cur = 0;
    const int row = min(row0 + wave * 32 + (lane >> 1), rows - 1);
    // LDS slot (lane & 1) of tile row (lane >> 1) holds source chunk slot ^ ((row >> 3) & 1): see the fragment reads
    off = (unsigned)row * 64u + (unsigned)((lane & 1) ^ ((lane >> 4) & 1)) * 16u;
  }
  __device__ __forceinline__ void dma(int pl, unsigned char* lds_dst) const {
    const char* src = base + (size_t)(cur + pl * plane) + off;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
  }
  // half 0 -> 1: the other 32 bytes of the same rows; half 1 -> 0: next K-block
  __device__ __forceinline__ void advance(int half) { cur += half ? 3u * plane - 32u : 32u; }
};

template <class Cfg> struct X3Ops {
  using A = std::conditional_t<Cfg::K16, X3OperandK16, X3Operand<Cfg::G, 3>>;
  using B = std::conditional_t<Cfg::K16, X3OperandK16, X3Operand<Cfg::G, Cfg::NPB>>;
};

// The first two tiles' DMA of a product (steps >= 2, even).  Separate from the loop: a persistent workgroup can put the NEXT
// product's first loads in flight before it stores the current one (tools/x3_gemm_ablate.hip; measured: no gain).
template <class Cfg>
__device__ __forceinline__ void x3_k16_prologue(unsigned char* lds, X3OperandK16& oa, X3OperandK16& ob) {
#ifndef X3_ABL_NO_DMA
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    unsigned char* stage = lds + h * Cfg::STAGE;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      unsigned char* dst = stage + j * Cfg::PL + wave * 1024;
      if (j < 3) oa.dma(j, dst); else ob.dma(j - 3, dst);
    }
    oa.advance(h); ob.advance(h);
  }
#endif
}

// After x3_k16_prologue.  Stores issued between the prologue and this loop (a persistent kernel's epilogue) would not
// break the first wait: loads return in order among themselves, so "at most 6 operations outstanding" still means that
// at most the 6 youngest LOADS (tile 1's) are, i.e. tile 0 has landed.
template <class Cfg>
__device__ __forceinline__ void x3_mainloop_k16(unsigned char* lds, int steps, X3OperandK16& oa, X3OperandK16& ob,
                                                f32x16 (&acc)[2][2]) {
  constexpr int PL = Cfg::PL, STAGE = Cfg::STAGE;
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, hh = lane >> 5;
  auto dma_piece = [&](int j, unsigned char* stage) {        // j 0..2: A planes, 3..5: B planes; this wave's 32 rows
#ifdef X3_ABL_NO_DMA
    return;
#endif
    unsigned char* dst = stage + j * PL + wave * 1024;
    if (j < 3) oa.dma(j, dst); else ob.dma(j - 3, dst);
  };
  int half = 0;
  const int sw = (hh ^ ((l31 >> 3) & 1)) << 4;
  const int a_rd = (wm * 64 + l31) * 32 + sw;
  const int b_rd = 3 * PL + (wn * 64 + l31) * 32 + sw;
  bf16x8 fa[2][3][2], fb[2][3][2];
  auto read_frags = [&](const unsigned char* stage, int slot) {
#ifdef X3_ABL_NO_FRAG
    if (stage != lds) return;
#endif
#pragma unroll
    for (int p = 0; p < 3; ++p) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        fa[slot][p][i] = *reinterpret_cast<const bf16x8*>(stage + a_rd + p * PL + i * 1024);
        fb[slot][p][i] = *reinterpret_cast<const bf16x8*>(stage + b_rd + p * PL + i * 1024);
      }
    }
  };
  constexpr int PA[6] = {2, 0, 1, 1, 0, 0};
  constexpr int PB[6] = {0, 2, 1, 0, 1, 0};

  unsigned char* s_cur = lds;
  unsigned char* s_nxt = lds + STAGE;
  unsigned char* s_nn = lds + 2 * STAGE;
  X3_WAIT_VM(6);
  __builtin_amdgcn_s_barrier();
  read_frags(s_cur, 0);

  // MODE 2: tile s+2 exists; 1: tile s+1 is the last; 0: last step.  SLOT: register slot of this step's fragments
  auto kstep = [&](auto mode_tag, auto slot_tag) {
    constexpr int MODE = decltype(mode_tag)::value, SLOT = decltype(slot_tag)::value;
#pragma unroll
    for (int q = 0; q < 6; ++q) {
#pragma unroll
      for (int im = 0; im < 2; ++im) {
#pragma unroll
        for (int in_ = 0; in_ < 2; ++in_) {
#ifndef X3_ABL_NO_MFMA
          acc[im][in_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[SLOT][PA[q]][im], fb[SLOT][PB[q]][in_], acc[im][in_], 0, 0, 0);
#else
          acc[im][in_][0] += (float)fa[SLOT][PA[q]][im][0] + (float)fb[SLOT][PB[q]][in_][0];
#endif
          const int done = (q * 2 + im) * 2 + in_ + 1;
          if (MODE == 2 && done <= 6) {
            __builtin_amdgcn_sched_barrier(0);
            dma_piece(done - 1, s_nn);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
      if (q == 3 && MODE != 0) {
        __builtin_amdgcn_sched_barrier(0);
        if (MODE == 2) X3_WAIT_VM(6); else X3_WAIT_VM(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifndef X3_ABL_NO_BARRIER
        __builtin_amdgcn_s_barrier();
#endif
        asm volatile("" ::: "memory");
        read_frags(s_nxt, SLOT ^ 1);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (MODE == 2) { oa.advance(half); ob.advance(half); half ^= 1; }
    unsigned char* tmp = s_cur; s_cur = s_nxt; s_nxt = s_nn; s_nn = tmp;
  };
  using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
  // steps is even (K % 32 == 0): two steps per trip keep the register slots static
  int s = 0;
  for (; s + 3 < steps; s += 2) { kstep(I2{}, I0{}); kstep(I2{}, I1{}); }
  // s + 2 == steps here
  kstep(I1{}, I0{});
  kstep(I0{}, I1{});
}

// Default "no prefetch" protocol for epilogues: derive from X3NoPrefetch<Derived> (CRTP) to get Pre / prefetch / the
// 7-argument apply forwarding to the plain apply(r, c, v).
template <class D>
struct X3NoPrefetch {
  template <int T> struct Pre {};
  template <class P> __device__ __forceinline__ void prefetch(P&, int, int, int, int, int) const {}
  template <class P> __device__ __forceinline__ float apply(P&, int, int, int, int r, int c, float v) const {
    return static_cast<const D*>(this)->apply(r, c, v);
  }
};

// C[z] (M x N) = A[z] B[z]^T on x3 panels; 1-D launch of cdiv(M,B) * cdiv(N,B) * batch workgroups in XCD-aware
// order (N-tile fastest, then M-tile, then batch: an XCD's 32 CUs share one batch's panels in their L2).
// Epilogues as in gemm.hip (apply / finish; SYMM epilogues also value / mirror and a run-time `symm` switch: then
// the LAUNCH holds only the tiles on or above the diagonal, gx (gx + 1) / 2 workgroups, and every tile below is
// written as the transpose of the one above, bitwise symmetric).
// One workgroup's tile of such a product: `bid` of `nblk` workgroups (the launch's own, or one problem's share of a grouped
// launch: gemm_x3_group3_kernel below).  `lds` holds >= Cfg::LDS_BYTES, 1024-byte aligned.
template <class Cfg, class Epi, class Mirror>
__device__ __forceinline__ void gemm_x3_tile(unsigned char* lds, unsigned bid, unsigned nblk, const __bf16* __restrict__ A, int M,
                                             long long strideA, const __bf16* __restrict__ B, int N, long long strideB, int K,
                                             const Epi& epi_in, const Mirror& mirror, int xbh, int xbw) {
  Epi epi = epi_in;                                 // (set_batch moves its pointers)
  const unsigned gx = (N + Cfg::BN - 1) / Cfg::BN, gy = (M + Cfg::BM - 1) / Cfg::BM;
  unsigned tile = xcd_swizzle(bid, nblk);
  unsigned bz = 0;
  int m0, n0;
  bool tri = false;
  if constexpr (Epi::SYMM) tri = epi.symm != 0;
  if (tri) {
    // symmetric product (M == N): the launch holds only the gx (gx + 1) / 2 tiles on or above the diagonal of every
    // batch entry, enumerated row by row -- tile row i = tiles (i, i .. gx - 1)
    const unsigned ntri = gx * (gx + 1) / 2;
    bz = tile / ntri;
    tile -= bz * ntri;
    unsigned i = 0, len = gx;
    while (tile >= len) { tile -= len; --len; ++i; }
    m0 = i * Cfg::BM; n0 = (i + tile) * Cfg::BN;
  } else {
    bz = tile / (gx * gy);
    const unsigned rem = tile - bz * (gx * gy);
    if (xbh > 0) {
      // 2-D XCD blocking: the launch order gives every XCD a contiguous run of xbh * xbw tile ids; lay each run out as
      // an xbh x xbw BLOCK of the tile grid instead of xbh * xbw / gx whole rows, so the XCD's L2 streams xbh A panels
      // and xbw B panels (16 x 16 tiles: 4 + 8 = 12 panels instead of 2 + 16 = 18)
      const unsigned per = (unsigned)(xbh * xbw), blk = rem / per, in = rem - blk * per;
      const unsigned bpr = gx / (unsigned)xbw;                     // blocks per block-row
      const unsigned ty = (blk / bpr) * xbh + in / xbw, tx = (blk % bpr) * xbw + in % xbw;
      m0 = ty * Cfg::BM; n0 = tx * Cfg::BN;
    } else {
      m0 = (rem / gx) * Cfg::BM; n0 = (rem % gx) * Cfg::BN;
    }
  }
  epi.set_batch(bz);
  typename X3Ops<Cfg>::A oa(A + (long long)bz * strideA, M, m0);
  typename X3Ops<Cfg>::B ob(B + (long long)bz * strideB, N, n0);
  f32x16 acc[Cfg::T][Cfg::T];
#pragma unroll
  for (int i = 0; i < Cfg::T; ++i)
#pragma unroll
    for (int j = 0; j < Cfg::T; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  PipeAccMap<Cfg> map;
  // Epilogues that read memory (read-modify-write outputs) may fetch their operands HERE, before the main loop: with one
  // workgroup per CU nothing else hides those loads.  Pre = per-lane register image, one slot per accumulator element.
  typename Epi::template Pre<Cfg::T> pre;
#pragma unroll
  for (int im = 0; im < Cfg::T; ++im)
#pragma unroll
    for (int in = 0; in < Cfg::T; ++in)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg)
        epi.prefetch(pre, im, in, reg, m0 + map.row(im, reg), n0 + map.colof(in));
  if constexpr (Cfg::K16) { x3_k16_prologue<Cfg>(lds, oa, ob); x3_mainloop_k16<Cfg>(lds, K >> 4, oa, ob, acc); }
  else x3_mainloop<Cfg>(lds, K >> 5, oa, ob, acc);
  float local = 0.f;
#pragma unroll
  for (int im = 0; im < Cfg::T; ++im)
#pragma unroll
    for (int in = 0; in < Cfg::T; ++in)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg)
        local += epi.apply(pre, im, in, reg, m0 + map.row(im, reg), n0 + map.colof(in), acc[im][in][reg]);
  if constexpr (Epi::SYMM) {
    static_assert(!Epi::SYMM || Cfg::LDS_BYTES >= Cfg::BM * (Cfg::BN + 1) * 4, "mirror tile");
    // SYMM epilogues on this core store only the upper triangle themselves: diagonal tiles mirror too
    if (epi.symm && n0 >= m0) mirror(epi, pre, reinterpret_cast<float*>(lds), m0, n0, M, N, acc, map);
  }
  __syncthreads();
  epi.finish(reinterpret_cast<float*>(lds), local);
}

template <class Cfg, class Epi, class Mirror>
__global__ __launch_bounds__(Cfg::NT) void gemm_x3_kernel(const __bf16* __restrict__ A, int M, long long strideA,
                                                          const __bf16* __restrict__ B, int N, long long strideB,
                                                          int K, Epi epi, Mirror mirror, int xbh = 0, int xbw = 0) {
  __shared__ __attribute__((aligned(1024))) unsigned char lds[Cfg::LDS_BYTES];
  gemm_x3_tile<Cfg, Epi, Mirror>(lds, blockIdx.x, gridDim.x, A, M, strideA, B, N, strideB, K, epi, mirror, xbh, xbw);
}

// THREE independent products in ONE launch (round 4): the forward cost / covariance matrices of a train step's loss section
// (run_strotss.py:33-40, nn/losses.py:39-80 of the reference) are mutually independent, each underfills the chip on its
// own (272, 256 and 171 workgroups for 512 two-per-CU slots) and a dependent launch costs ~5 us end to end.  Problem p owns
// the workgroups [first_p, first_p + n_p) of the grid, first_p a multiple of 8 so that a problem's local workgroup ids keep
// the launch's XCD round-robin (xcd_swizzle); the padding workgroups exit.  Every problem keeps its own tile
// configuration, epilogue and product order: results are bit for bit those of three separate launches.
template <class Epi>
struct X3Problem {
  const __bf16* A; int M; long long strideA;
  const __bf16* B; int N; long long strideB;
  int K; Epi epi; int xbh, xbw;
  unsigned n;                                       // workgroups of this problem
};
__host__ __device__ __forceinline__ unsigned x3_pad8(unsigned n) { return (n + 7u) & ~7u; }
template <class C0, class E0, class M0, class C1, class E1, class M1, class C2, class E2, class M2>
__global__ __launch_bounds__(256) void gemm_x3_group3_kernel(X3Problem<E0> p0, X3Problem<E1> p1, X3Problem<E2> p2) {
  static_assert(C0::NT == 256 && C1::NT == 256 && C2::NT == 256, "grouped x3 launch: 256-thread workgroups");
  constexpr int L01 = C0::LDS_BYTES > C1::LDS_BYTES ? C0::LDS_BYTES : C1::LDS_BYTES;
  constexpr int LDSB = L01 > C2::LDS_BYTES ? L01 : C2::LDS_BYTES;
  __shared__ __attribute__((aligned(1024))) unsigned char lds[LDSB];
  unsigned b = blockIdx.x;
  if (b < x3_pad8(p0.n)) {
    if (b < p0.n) { M0 m; gemm_x3_tile<C0, E0, M0>(lds, b, p0.n, p0.A, p0.M, p0.strideA, p0.B, p0.N, p0.strideB, p0.K, p0.epi, m, p0.xbh, p0.xbw); }
    return;
  }
  b -= x3_pad8(p0.n);
  if (b < x3_pad8(p1.n)) {
    if (b < p1.n) { M1 m; gemm_x3_tile<C1, E1, M1>(lds, b, p1.n, p1.A, p1.M, p1.strideA, p1.B, p1.N, p1.strideB, p1.K, p1.epi, m, p1.xbh, p1.xbw); }
    return;
  }
  b -= x3_pad8(p1.n);
  if (b < p2.n) { M2 m; gemm_x3_tile<C2, E2, M2>(lds, b, p2.n, p2.A, p2.M, p2.strideA, p2.B, p2.N, p2.strideB, p2.K, p2.epi, m, p2.xbh, p2.xbw); }
}

// Row-major f32 (rows x ld, K <= ld columns used, K % 32 == 0) -> x3 panels, batched over blockIdx.y.
// Lane order: 8 lanes cover one row's 32 k (128 B read), consecutive rows follow: 512 contiguous bytes per plane
// and wave store.
static __global__ __launch_bounds__(256) void x3_split_rows_kernel(const float* __restrict__ x, int rows, int ld, int K,
                                                            long long stride_in, __bf16* __restrict__ out,
                                                            long long stride_out) {
  x += (long long)blockIdx.y * stride_in;
  out += (long long)blockIdx.y * stride_out;
  const size_t total = (size_t)rows * (K >> 2);
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const int k4 = (int)(e & 7);
    const size_t rr = e >> 3;
    const int row = (int)(rr % rows), kb = (int)(rr / rows);
    const int k0 = kb * 32 + k4 * 4;
    x3_store4(out, rows, row, k0, *reinterpret_cast<const f32x4*>(x + (size_t)row * ld + k0));
  }
}

