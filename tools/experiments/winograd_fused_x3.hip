// Winograd F(4x4, 3x3) in ONE persistent kernel with the 36 transform-domain products on the bf16 MFMA by exact
// 3-way operand splitting (mfma_x3.h): the successor of winograd_fused.hip's f32-MFMA kernel for the same layers
// (nn/model.py:44-48 of the reference: block1_conv2 ... block3_conv3, forward and data-gradient).
//
// Why: on gfx950 the f32 MFMA (v_mfma_f32_32x32x2_f32) runs at 1/16 of the bf16 rate AND blocks the SIMD's vector issue
// for its whole 64 cycles, so the f32 kernel pays its input transform (VALU) in MFMA time: 3900 cycles per 8-channel
// phase for 2304 cycles of MFMA.  Here every f32 value x of V = B^T d B and of U = G g G^T is the exact sum of three
// bf16 values h + m + l, a product of two bf16 values is exact in f32, and the six leading partial products
//     hh + hm + mh + mm + lh + hl        (the dropped ml + lm + ll are below 2^-24 relative)
// are accumulated in f32 by v_mfma_f32_32x32x16_bf16: f32-class results (same error class as the f32 MFMA, tested
// against the float64 oracle side by side) at 6/16 of the MFMA time -- and the bf16 MFMA holds the vector issue for 8
// of its 32 cycles only, so the transform of the next chunk runs UNDER the products of the current one.
//
// K = 16 of one MFMA = two operand planes of the chunk's 8 input channels side by side ("plane concatenation"):
//     lanes 0-31 carry k = 0..7  = 8 channels of one plane,  lanes 32-63 carry k = 8..15 = 8 channels of another:
//     MFMA 1:  A = [a_l | a_h]   B = [b_h | b_l]   ->  lh + hl
//     MFMA 2:  A = [a_m | a_m]   B = [b_h | b_m]   ->  mh + mm
//     MFMA 3:  A = [a_h | a_h]   B = [b_h | b_m]   ->  hh + hm        (smallest terms first)
// A comes from LDS (V planes, any lane may read any plane: duplicates are free), B from global memory: per (position,
// 32-cout group, chunk) the planes h, m, l of U lie as 3 x 512 contiguous bytes (strotss_conv3x3_winograd_pack_x3),
// X = [h | m] is ONE linear 1 KB wave load, Y = [h | l] a second one whose lower half re-reads h from L1
// (BLGP, which would broadcast a lane half for free, is ignored by this instruction on gfx950: tools/mfma_blgp_probe.hip).
//
// Work item, persistent grid, XCD ownership, the chunk pipeline (patch of chunk c+3 in flight, patch c+2 registers ->
// LDS, transform of chunk c+1, products of chunk c, ONE barrier per phase) and the output transform are those of
// winograd_fused.hip.  What differs is WHO transforms: there waves 0-3 do (thread = tile x channel pair x 3 of the 6
// output rows); here ALL 12 waves do, thread = (tile, channel pair, ONE output row): measured (tools/fused_x3_ablate.hip),
// one transforming wave per SIMD needs ~3700 cycles per phase for its ~330 vector + 84 LDS instructions -- a lone wave
// issues a vector instruction every 4+ cycles and waits out every LDS round trip -- whereas three waves per SIMD
// interleave (2 cycles per instruction) and each holds 12 row values instead of 36.  The price is LDS reads: every
// output row re-reads its 3-4 input rows.  LDS: V planes 2 x 54 KB ([plane][position][tile][8 ch] bf16, 16-byte rows,
// conflict-free for the 4-byte transform stores and the 16-byte fragment reads alike), raw patches 2 x 24 KB; the
// epilogue's three exchange columns alias V[1] (two) and raw[0] (one), both free at an item's end.
#include <stdlib.h>

#include "internal.h"
#include "mfma_x3.h"

// timing ablations for tools/fused_x3_ablate.hip (results are wrong when any is defined)
#ifdef X3_NO_TRANSFORM
#define ABL_T(x) do {} while (0)
#else
#define ABL_T(x) x
#endif
#ifdef X3_NO_MFMA
#define ABL_M(x)
#else
#define ABL_M(x) x
#endif
#ifdef X3_NO_U
#define ABL_U(x) do {} while (0)
#else
#define ABL_U(x) x
#endif
#ifdef X3_NO_RAW
#define ABL_R(x) do {} while (0)
#else
#define ABL_R(x) x
#endif

namespace {

constexpr int F_TR = 4, F_TC = 8;                  // Winograd tiles per work item: rows x cols
constexpr int F_TILES = F_TR * F_TC;               // 32 = M of the MFMA tile
constexpr int F_PH = 4 * F_TR + 2, F_PW = 4 * F_TC + 2;   // input patch 18 x 34
constexpr int F_NPX = F_PH * F_PW;                 // 612
constexpr int F_KC = 8;                            // input channels per chunk
constexpr int F_PS = 10;                           // LDS floats per patch pixel: tile stride 40 floats, conflict-free
constexpr int F_RAW = 6 * F_TILES * 32;             // floats of one patch buffer (612 px x 10 + dummy slot) = one exchange column
constexpr int X_VROW = 16;                         // bytes of one (plane, position, tile) row: 8 channels bf16
constexpr int X_VPOS = F_TILES * X_VROW;           // 512 B per (plane, position)
constexpr int X_VPLANE = 36 * X_VPOS;              // 18 KB per plane
constexpr int X_VBUF = 3 * X_VPLANE;               // 54 KB per V buffer
constexpr int X_LDS = 2 * X_VBUF + 2 * F_RAW * 4;  // 159744 B
constexpr int F_NT = 768;                          // threads: 12 waves, 3 positions each
constexpr int F_NLOAD = (F_NPX * 2 + F_NT - 1) / F_NT;    // float4 loads per thread per chunk (2)
constexpr int X_UNIT = 3 * 32 * 16;                // bytes of U per (position, cout group, chunk): planes h, m, l
static_assert(X_LDS <= 160 * 1024, "LDS");
static_assert(F_RAW >= F_NPX * F_PS + 4 && X_VBUF >= 2 * F_RAW * 4, "epilogue columns alias V[1] (two) and raw[0] (one)");

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

struct ItemRef { int goff[F_NLOAD]; unsigned okm; int g; };

// exact 3-way split of a channel pair: three dwords of 2 x bf16 (mfma_x3.h split3, two values at a time)
__device__ __forceinline__ void split_pair(const f32x2 v, unsigned& h, unsigned& m, unsigned& l) {
  const bf16x2 hb = {(__bf16)v[0], (__bf16)v[1]};
  const f32x2 r1 = {v[0] - (float)hb[0], v[1] - (float)hb[1]};
  const bf16x2 mb = {(__bf16)r1[0], (__bf16)r1[1]};
  const f32x2 r2 = {r1[0] - (float)mb[0], r1[1] - (float)mb[1]};
  const bf16x2 lb = {(__bf16)r2[0], (__bf16)r2[1]};
  h = __builtin_bit_cast(unsigned, hb); m = __builtin_bit_cast(unsigned, mb); l = __builtin_bit_cast(unsigned, lb);
}

// row R of B^T applied down a column: t = sum_k BT[R][k] d[k]  (Lavin & Gray, points 0, +-1, +-2, inf); LD(k) reads d[k]
template <int R, class LD>
__device__ __forceinline__ f32x2 bt_row(LD&& ld) {
  if constexpr (R == 0) { const f32x2 d0 = ld(0), d2 = ld(2), d4 = ld(4); return 4.f * d0 - 5.f * d2 + d4; }
  if constexpr (R == 1) { const f32x2 d1 = ld(1), d2 = ld(2), d3 = ld(3), d4 = ld(4); return -4.f * d1 - 4.f * d2 + d3 + d4; }
  if constexpr (R == 2) { const f32x2 d1 = ld(1), d2 = ld(2), d3 = ld(3), d4 = ld(4); return 4.f * d1 - 4.f * d2 - d3 + d4; }
  if constexpr (R == 3) { const f32x2 d1 = ld(1), d2 = ld(2), d3 = ld(3), d4 = ld(4); return -2.f * d1 - d2 + 2.f * d3 + d4; }
  if constexpr (R == 4) { const f32x2 d1 = ld(1), d2 = ld(2), d3 = ld(3), d4 = ld(4); return 2.f * d1 - d2 - 2.f * d3 + d4; }
  if constexpr (R == 5) { const f32x2 d1 = ld(1), d3 = ld(3), d5 = ld(5); return 4.f * d1 - 5.f * d3 + d5; }
}

// in: (H, W, K) NHWC, K % 32 == 0; U: x3 planes per (position, 32-cout group, 8-channel chunk), see the pack kernel;
// out / mask: (H, W, Cout), Cout % 32 == 0.   !MASK: out = relu ? max(Y + bias, 0) : Y + bias;  MASK: out = mask > 0 ? Y : 0.
template <bool MASK>
__global__ __launch_bounds__(F_NT) void winograd43_fused_x3_kernel(const float* __restrict__ in, int H, int W, int K,
                                                                   const unsigned char* __restrict__ U, int Cout,
                                                                   const float* __restrict__ bias,
                                                                   const float* __restrict__ mask, int relu,
                                                                   float* __restrict__ out, float* __restrict__ pool,
                                                                   unsigned char* __restrict__ pool_code, int RW, int NG,
                                                                   int nitems) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[X_LDS];
  unsigned char* const Vb = lds;                                        // V planes [2]
  float* const Rb = reinterpret_cast<float*>(lds + 2 * X_VBUF);         // raw[2]
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);      // scalar: the per-phase address arithmetic runs on the SALU
  const int l31 = lane & 31, hh = lane >> 5;

  // items of this workgroup: XCD x = blockIdx % 8 owns [x * per, (x + 1) * per), its workgroups interleave them
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslot = gridDim.x >> 3;
  const int per = (nitems + 7) >> 3;
  const int item_end = min(nitems, (xcd + 1) * per);
  int item = xcd * per + slot;
  if (item >= item_end) return;                    // whole workgroup

  // ---- patch loader of one item: element e = t + 768 i  ->  pixel e >> 1, channel quad e & 1
  auto setup = [&](int it, ItemRef& r) {
    const int region = it / NG;
    r.g = it - region * NG;
    const int ry = region / RW, rx = region - ry * RW;
    r.okm = 0;
#pragma unroll
    for (int i = 0; i < F_NLOAD; ++i) {
      const int e = t + F_NT * i;
      const int px = e >> 1, c4 = e & 1;
      const int py = px / F_PW, pxx = px - py * F_PW;
      const int gy = ry * (4 * F_TR) - 1 + py, gx = rx * (4 * F_TC) - 1 + pxx;
      const bool ok = (e < F_NPX * 2) && gy >= 0 && gy < H && gx >= 0 && gx < W;
      r.okm |= (unsigned)ok << i;
      r.goff[i] = ok ? (gy * W + gx) * K + c4 * 4 : 0;   // out-of-image: load pixel 0, zeroed at the LDS store
    }
  };
  auto load_raw = [&](const ItemRef& r, int kc, f32x4 (&st)[F_NLOAD]) {
#pragma unroll
    for (int i = 0; i < F_NLOAD; ++i) st[i] = *reinterpret_cast<const f32x4*>(in + r.goff[i] + kc * F_KC);
  };
  auto store_raw = [&](float* raw, unsigned okm, const f32x4 (&st)[F_NLOAD]) {   // branch-free (dummy slot at the end)
#pragma unroll
    for (int i = 0; i < F_NLOAD; ++i) {
      const int e = t + F_NT * i;
      const int o = (e < F_NPX * 2) ? (e >> 1) * F_PS + (e & 1) * 4 : F_NPX * F_PS;
      const bool ok = (okm >> i) & 1u;
      f32x2 lo = {ok ? st[i][0] : 0.f, ok ? st[i][1] : 0.f}, hi = {ok ? st[i][2] : 0.f, ok ? st[i][3] : 0.f};
      *reinterpret_cast<f32x2*>(raw + o) = lo;      // 40-byte pixel stride: 8-byte aligned only
      *reinterpret_cast<f32x2*>(raw + o + 2) = hi;
    }
  };
  // ---- input transform, ALL waves: thread = (tile, channel PAIR, output row xf_r) of V = B^T d B -- waves 2r and
  // 2r + 1 take row r, so the row is wave-uniform and its coefficients are compile-time -- packed-f32 arithmetic on
  // the pair, then the exact split of the 6 values of the row into their h, m, l planes.
  const int xf_it = t & 127, xf_r = __builtin_amdgcn_readfirstlane(t >> 7);
  const int xf_tile = xf_it >> 2, xf_pr = xf_it & 3;
  const int xf_src = ((4 * (xf_tile >> 3)) * F_PW + 4 * (xf_tile & 7)) * F_PS + 2 * xf_pr;
  const int xf_dst = (6 * xf_r) * X_VPOS + xf_tile * X_VROW + xf_pr * 4;
  auto transform_down = [&](const float* raw, f32x2 (&tr)[6]) {     // tr[q] = (B^T d)[xf_r][q]
    const float* src = raw + xf_src;
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      auto ld = [&](int k) { return *reinterpret_cast<const f32x2*>(src + (k * F_PW + q) * F_PS); };
      switch (xf_r) {                               // uniform
        case 0: tr[q] = bt_row<0>(ld); break;
        case 1: tr[q] = bt_row<1>(ld); break;
        case 2: tr[q] = bt_row<2>(ld); break;
        case 3: tr[q] = bt_row<3>(ld); break;
        case 4: tr[q] = bt_row<4>(ld); break;
        default: tr[q] = bt_row<5>(ld); break;
      }
    }
  };
  auto transform_across = [&](unsigned char* V, const f32x2 (&tr)[6]) {   // the 6 positions of row xf_r
    const f32x2 d0 = tr[0], d1 = tr[1], d2 = tr[2], d3 = tr[3], d4 = tr[4], d5 = tr[5];
    f32x2 v[6];
    v[0] = 4.f * d0 - 5.f * d2 + d4;
    v[1] = -4.f * d1 - 4.f * d2 + d3 + d4;
    v[2] = 4.f * d1 - 4.f * d2 - d3 + d4;
    v[3] = -2.f * d1 - d2 + 2.f * d3 + d4;
    v[4] = 2.f * d1 - d2 - 2.f * d3 + d4;
    v[5] = 4.f * d1 - 5.f * d3 + d5;
    unsigned char* dst = V + xf_dst;
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      unsigned h, m, l;
#ifdef X3_NO_SPLIT
      h = __builtin_bit_cast(unsigned, v[q][0]); m = __builtin_bit_cast(unsigned, v[q][1]); l = h ^ m;
#else
      split_pair(v[q], h, m, l);
#endif
#ifdef X3_NO_VSTORE
      if (h == 0x12345678u && m == 0x9abcdef0u && l == 0x0fedcba9u)
#endif
      {
        *reinterpret_cast<unsigned*>(dst + q * X_VPOS) = h;
        *reinterpret_cast<unsigned*>(dst + q * X_VPOS + X_VPLANE) = m;
        *reinterpret_cast<unsigned*>(dst + q * X_VPOS + 2 * X_VPLANE) = l;
      }
    }
  };
  // ---- U fragments of positions wave, wave + 12, wave + 24 for (group g, chunk kc): X = [h | m], Y = [h | l]
  const int nchunk = K / F_KC;                      // even, >= 4
  const size_t b_step = (size_t)12 * NG * nchunk * X_UNIT;
  const int x_off = lane * 16, y_off = l31 * 16 + (hh ? 1024 : 0);
  auto load_u1 = [&](int g, int kc, int j, bf16x8& bxj, bf16x8& byj) {
    const unsigned char* bp = U + (((size_t)wave * NG + g) * nchunk + kc) * X_UNIT + j * b_step;
    bxj = *reinterpret_cast<const bf16x8*>(bp + x_off);
    byj = *reinterpret_cast<const bf16x8*>(bp + y_off);
  };
  // A fragments of position p from a V buffer: [a_h | a_h], [a_m | a_m], [a_l | a_h]
  const int a_h = l31 * X_VROW, a_m = X_VPLANE + l31 * X_VROW, a_lh = (hh ? 0 : 2 * X_VPLANE) + l31 * X_VROW;

  ItemRef cur, nxt;
  setup(item, cur);
  f32x4 S0[F_NLOAD], S1[F_NLOAD];                   // patches in flight: S[c & 1] is loaded in phase c with chunk c + 3
  bf16x8 bx[3], by[3];                             // U fragments of the CURRENT chunk (reloaded behind its last MFMA)
  // ---- prologue: raw[0] = chunk 0, raw[1] = chunk 1, V[0] = T(chunk 0), S1 = chunk 2 in flight, b = U(chunk 0)
  load_raw(cur, 0, S0);
  load_raw(cur, 1, S1);
  store_raw(Rb, cur.okm, S0);
  store_raw(Rb + F_RAW, cur.okm, S1);
#pragma unroll
  for (int j = 0; j < 3; ++j) load_u1(cur.g, 0, j, bx[j], by[j]);
  load_raw(cur, 2, S1);
  __syncthreads();
  {
    f32x2 tr[6];
    transform_down(Rb, tr);
    transform_across(Vb, tr);
  }
  __syncthreads();

  for (;;) {
    f32x16 acc[3];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const int next = item + nslot;
    const bool more = next < item_end;              // uniform
    if (more) setup(next, nxt); else nxt = cur;     // past the last item the pipeline re-reads it (results unused)

    // stream offset d from chunk kc of the current item -> (item ref, chunk)
#define CHUNK_AT(d, ref, kk)                             \
    const bool wrap_##d = kc + d >= nchunk;              \
    const ItemRef& ref = wrap_##d ? nxt : cur;           \
    const int kk = kc + d - (wrap_##d ? nchunk : 0);
#define READ_A(j, al, am, ah)                                                                                         \
    const bf16x8 al = *reinterpret_cast<const bf16x8*>(Vc + (wave + 12 * (j)) * X_VPOS + a_lh);                       \
    const bf16x8 am = *reinterpret_cast<const bf16x8*>(Vc + (wave + 12 * (j)) * X_VPOS + a_m);                        \
    const bf16x8 ah = *reinterpret_cast<const bf16x8*>(Vc + (wave + 12 * (j)) * X_VPOS + a_h);
    /* the three MFMAs of a unit, then -- behind the last one that reads them, in-order issue -- the loads of the  */
    /* next chunk's U into the SAME registers: a whole phase to land, no second register set                       */
#define MFMA_UNIT(j, al, am, ah)                                                                                      \
    ABL_M(acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, by[j], acc[j], 0, 0, 0);)     /* lh + hl */            \
    ABL_M(acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bx[j], acc[j], 0, 0, 0);)     /* mh + mm */            \
    ABL_M(acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bx[j], acc[j], 0, 0, 0);)     /* hh + hm */            \
    __builtin_amdgcn_sched_barrier(0);                                                                                \
    ABL_U(load_u1(r1.g, k1, j, bx[j], by[j]));                                                                        \
    __builtin_amdgcn_sched_barrier(0);
    /* One phase.  The bf16 MFMA holds the SIMD's vector issue for 8 of its 32 cycles: each wave alternates its three */
    /* MFMA units with the stages of its share of the NEXT chunk's transform, so that while one wave waits for the   */
    /* matrix pipe the other two of its SIMD issue vector work.                                                       */
#define PHASE(PAR, SLOAD, SSTORE)                                                                                     \
    {                                                                                                                 \
      CHUNK_AT(1, r1, k1) CHUNK_AT(3, r3, k3)                                                                         \
      const unsigned okm2 = (kc + 2 >= nchunk) ? nxt.okm : cur.okm;                                                   \
      const unsigned char* Vc = Vb + PAR * X_VBUF;                                                                    \
      f32x2 tr[6];                                                                                                    \
      READ_A(0, al0, am0, ah0)                                                                                        \
      READ_A(1, al1, am1, ah1)                                                                                        \
      __builtin_amdgcn_sched_barrier(0);                                                                              \
      MFMA_UNIT(0, al0, am0, ah0)                                                                                     \
      ABL_T(transform_down(Rb + (1 - PAR) * F_RAW, tr));                                                              \
      __builtin_amdgcn_sched_barrier(0);                                                                              \
      READ_A(2, al2, am2, ah2)                                                                                        \
      MFMA_UNIT(1, al1, am1, ah1)                                                                                     \
      ABL_T(transform_across(Vb + (1 - PAR) * X_VBUF, tr));                                                           \
      __builtin_amdgcn_sched_barrier(0);                                                                              \
      MFMA_UNIT(2, al2, am2, ah2)                                                                                     \
      ABL_R(load_raw(r3, k3, SLOAD));                                                                                 \
      __builtin_amdgcn_sched_barrier(0);                                                                              \
      ABL_R(store_raw(Rb + PAR * F_RAW, okm2, SSTORE));                                                               \
      __syncthreads();                                                                                                \
      ++kc;                                                                                                           \
    }
    for (int kc = 0; kc < nchunk;) {
      PHASE(0, S0, S1)
      PHASE(1, S1, S0)
    }
#undef PHASE
#undef MFMA_UNIT
#undef READ_A
#undef CHUNK_AT

    // ---- output transform by columns q of the 6x6 positions: waves q and q + 6 hold its rows {0,2,4} / {1,3,5}
    const int region = item / NG;
    const int ry = region / RW, rx = region - ry * RW;
    const int y0 = ry * (4 * F_TR), x0 = rx * (4 * F_TC);
    float Y[2][4][4];                               // starts at the bias (none for the data-gradient)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      float bv = 0.f;
      if constexpr (!MASK) { if (bias && t < 512) bv = bias[cur.g * 32 + (t & 31)]; }
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int q = 0; q < 4; ++q) Y[i][r][q] = bv;
    }
    // Three columns per round in V[1] (read by the last phase's MFMAs: nchunk is even, an item ends on an odd phase)
    // and raw[0] (transformed in the last phase); V[0] and raw[1] already hold the next item's first chunks.
    const int myq = wave % 6, myr = wave / 6;
    float* const colbuf[3] = {reinterpret_cast<float*>(Vb + X_VBUF), reinterpret_cast<float*>(Vb + X_VBUF) + F_RAW, Rb};
#pragma unroll
    for (int round = 0; round < 2; ++round) {
      if (myq / 3 == round) {
        float* base = colbuf[0];
        if (myq % 3 == 1) base = colbuf[1];
        if (myq % 3 == 2) base = colbuf[2];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          float* dst = base + ((myr + 2 * j) * F_TILES + 4 * hh) * 32 + l31;
#pragma unroll
          for (int r = 0; r < 16; ++r) dst[((r & 3) + 8 * (r >> 2)) * 32] = acc[j][r];
        }
      }
      __syncthreads();
      if (t < 512) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int e = t + 512 * i;
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            const int q = 3 * round + k;
            const float* src = colbuf[k] + (e >> 5) * 32 + (e & 31);      // [row][tile][cout]
            float m[6];
#pragma unroll
            for (int r = 0; r < 6; ++r) m[r] = src[r * (F_TILES * 32)];
            float sv[4];
            sv[0] = m[0] + m[1] + m[2] + m[3] + m[4];
            sv[1] = m[1] - m[2] + 2.f * m[3] - 2.f * m[4];
            sv[2] = m[1] + m[2] + 4.f * m[3] + 4.f * m[4];
            sv[3] = m[1] - m[2] + 8.f * m[3] - 8.f * m[4] + m[5];
            // Y[r][:] += s[r] * A^T[:, q]
            constexpr float AT[4][6] = {{1.f, 1.f, 1.f, 1.f, 1.f, 0.f}, {0.f, 1.f, -1.f, 2.f, -2.f, 0.f},
                                        {0.f, 1.f, 1.f, 4.f, 4.f, 0.f}, {0.f, 1.f, -1.f, 8.f, -8.f, 1.f}};
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
              for (int c = 0; c < 4; ++c)
                if (AT[c][q] != 0.f) Y[i][r][c] += AT[c][q] * sv[r];
          }
        }
      }
      __syncthreads();
    }
    if (t < 512) {
      const float lo = (!MASK && relu) ? 0.f : -INFINITY;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int e = t + 512 * i;
        const int cl = e & 31, tile = e >> 5;
        const int ty = tile >> 3, tx = tile & 7;
        const int co = cur.g * 32 + cl;
        const int yb = y0 + 4 * ty, xb = x0 + 4 * tx;
        // Every memory-dependent value (bias: folded into Y before the rounds; ReLU mask: one batch of clamped
        // loads condensed to a bit mask) is resolved before the first store, so the stores carry no s_waitcnt.
        const bool tile_in = yb < H && xb < W, full = yb + 3 < H && xb + 3 < W;
        const size_t ob = tile_in ? ((size_t)yb * W + xb) * Cout + co : (size_t)co;
        float* op = out + ob;
        unsigned keep = 0xffffu;
        if constexpr (MASK) {
          const float* mp = mask + ob;
          float mk[4][4];
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              const bool ok = tile_in && yb + r < H && xb + c < W;
              mk[r][c] = mp[ok ? (r * W + c) * Cout : 0];
            }
          keep = 0;
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) keep |= (mk[r][c] > 0.f ? 1u : 0u) << (4 * r + c);
        }
        if (full) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c)
              op[(r * W + c) * Cout] = ((keep >> (4 * r + c)) & 1u) ? fmaxf(Y[i][r][c], lo) : 0.f;
        } else if (tile_in) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c)
              if (yb + r < H && xb + c < W)
                op[(r * W + c) * Cout] = ((keep >> (4 * r + c)) & 1u) ? fmaxf(Y[i][r][c], lo) : 0.f;
        }
        if constexpr (!MASK) {
          // fused 2x2/2 max-pool of the activations just written (the tile's 4x4 outputs hold 2x2 windows);
          // windows are emitted only where they lie inside the image (floor pooling)
          if (pool) {
            const int PH = H >> 1, PW = W >> 1, py0 = yb >> 1, px0 = xb >> 1;
#pragma unroll
            for (int pr = 0; pr < 2; ++pr)
#pragma unroll
              for (int pc = 0; pc < 2; ++pc)
                if (py0 + pr < PH && px0 + pc < PW) {
                  const float v4[4] = {Y[i][2 * pr][2 * pc], Y[i][2 * pr][2 * pc + 1], Y[i][2 * pr + 1][2 * pc],
                                       Y[i][2 * pr + 1][2 * pc + 1]};
                  int best = 0;
                  float bv = v4[0];
#pragma unroll
                  for (int q4 = 1; q4 < 4; ++q4)
                    if (v4[q4] > bv) { bv = v4[q4]; best = q4; }
                  const size_t po = ((size_t)(py0 + pr) * PW + px0 + pc) * Cout + co;
                  pool[po] = fmaxf(bv, lo);
                  if (pool_code) pool_code[po] = (unsigned char)(bv > 0.f ? best : 4);   // argmax code of maxpool2_fwd
                }
          }
        }
      }
    }
    if (!more) break;
    item = next;
    cur = nxt;                                      // (the round loop ended on a barrier: the exchange columns are free)
  }
}

// u (36, rows, k) f32 -> planes per (position p, group r / 32, chunk c / 8): [plane h | m | l][32 rows][8 channels] bf16
__global__ __launch_bounds__(256) void winograd43_pack_x3_kernel(const float* __restrict__ u, int rows, int k, size_t total,
                                                                __bf16* __restrict__ up) {
  const int ng = rows / 32, nc = k / F_KC;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const int c = (int)(e % k);
    const size_t pr = e / k;
    const int r = (int)(pr % rows), p = (int)(pr / rows);
    const float x = u[e];
    const __bf16 h = (__bf16)x;
    const float r1 = x - (float)h;
    const __bf16 m = (__bf16)r1;
    const __bf16 l = (__bf16)(r1 - (float)m);
    const size_t o = ((((size_t)p * ng + r / 32) * nc + c / 8) * 3 * 32 + r % 32) * 8 + c % 8;
    up[o] = h; up[o + 32 * 8] = m; up[o + 2 * 32 * 8] = l;
  }
}

}  // namespace

int st_winograd43_pack_x3(const float* u_prk, int rows, int k, void* u_packed, hipStream_t st) {
  const size_t total = (size_t)36 * rows * k;
  hipLaunchKernelGGL(winograd43_pack_x3_kernel, dim3((unsigned)min((size_t)4096, (total + 255) / 256)), dim3(256), 0, st,
                     u_prk, rows, k, total, reinterpret_cast<__bf16*>(u_packed));
  ST_LAUNCH_RET();
}

// STROTSS_WINO_FUSED_X3: 1 = this kernel (pre-split V planes in LDS, pre-split U planes from memory) wherever the fused
// form runs; 0 (default) = winograd_fused.hip, whose X3 form splits both operands in registers instead.  This kernel
// is the SLOWER one (block1_conv2 forward at 1024^2: 350-450 us against 338 us for the f32-MFMA kernel): its
// bottleneck is not arithmetic but the cache-line request rate of a CU -- U as three bf16 planes is 1.5-2x the bytes
// of the f32 fragments (72 KB + 20 KB of patch per phase and CU = ~1350 line requests for ~1000 cycles of products)
// and the V planes are 1.5x the LDS bytes; tools/fused_x3_ablate.hip holds the ablations.
bool st_winograd43_fused_x3_enabled() {
  static int on = -1;
  if (on < 0) {
    const char* e = getenv("STROTSS_WINO_FUSED_X3"); on = e ? atoi(e) : 0;
    const char* x = getenv("STROTSS_X3"); if (x && atoi(x) == 0) on = 0;
  }
  return on != 0;
}

int st_winograd43_fused_x3(const float* in, int h, int w, int cin, const void* U, const float* bias, int cout,
                           const float* mask, int relu, float* out, float* pool_out, unsigned char* pool_code,
                           hipStream_t st) {
  if (cin % 32 != 0 || cout % 32 != 0) return STROTSS_EALIGN;
  if ((size_t)h * w * cin >= ((size_t)1 << 30) || (size_t)h * w * cout >= ((size_t)1 << 30)) return STROTSS_EALIGN;
  const int TH = (h + 3) / 4, TW = (w + 3) / 4;
  const int RH = (TH + F_TR - 1) / F_TR, RW = (TW + F_TC - 1) / F_TC;
  const int NG = cout / 32, nitems = RH * RW * NG;
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return STROTSS_EINVAL;
    cus = prop.multiProcessorCount / 8 * 8;
    if (cus < 8) cus = 8;
  }
  int grid = cus;                                  // persistent: one 156 KB-LDS workgroup per CU, a multiple of 8
  while (grid > 8 && grid / 2 >= nitems) grid /= 2;
  const unsigned char* Up = reinterpret_cast<const unsigned char*>(U);
  if (mask)
    hipLaunchKernelGGL(winograd43_fused_x3_kernel<true>, dim3((unsigned)grid), dim3(F_NT), 0, st, in, h, w, cin, Up, cout,
                       bias, mask, relu, out, pool_out, pool_code, RW, NG, nitems);
  else
    hipLaunchKernelGGL(winograd43_fused_x3_kernel<false>, dim3((unsigned)grid), dim3(F_NT), 0, st, in, h, w, cin, Up, cout,
                       bias, mask, relu, out, pool_out, pool_code, RW, NG, nitems);
  ST_LAUNCH_RET();
}
