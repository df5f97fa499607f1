// Winograd F(4x4, 3x3) in ONE persistent kernel for the layers whose three-kernel form (winograd.hip) is bound by
// its transform traffic (nn/model.py:44-48 of the reference: block1_conv2, block2_conv1/2, ...).
//
// The three-kernel form moves V = B^T d B and M = V U^T -- each 2.25x the activation -- through HBM: 2.4 GB per
// launch at 1024^2 x 64 for 0.54 GB of algorithmic traffic.  Here a workgroup (768 threads, one per CU) owns work
// items of 4x8 Winograd tiles (16x32 output pixels) x 32 output channels and keeps everything on chip.  The input
// channels stream through in chunks of 8, software-pipelined over the chunk stream of ALL the workgroup's items:
//
//   phase c:  waves 0-3: input transform of chunk c+1:  raw[(c+1)&1] (18x34 px x 8 ch)  ->  V[(c+1)&1] = B^T d B
//             all waves: M[p] += V[c&1][p] U[p]^T for their 3 of the 36 positions (4 x v_mfma_f32_32x32x2_f32 each)
//             all waves: [issue] U fragments of chunk c+1 (L2 -> registers), input patch of chunk c+3 (HBM -> registers)
//             all waves: patch of chunk c+2 (in flight since phase c-1): registers -> raw[c&1]
//             ONE barrier
//
// so the VALU-heavy transform of one wave per SIMD runs under the other waves' MFMAs, and global loads have one
// (U, L2) or two (patch, HBM) phases to land.  After an item's last chunk the accumulators go through a small LDS
// exchange buffer one transform-domain column at a time (waves w and w+6 hold column w % 6): Y = A^T M A is
// accumulated in registers, then bias/ReLU or the ReLU mask, then the stores -- the prefetched V / raw of the next
// item stay in place.  LDS: V 2 x 36 KB (XOR-swizzled 32-byte rows), raw 2 x 24 KB, exchange 24 KB.
#include <stdlib.h>

#include "internal.h"
#include "mfma_pipe.h"

namespace {

#ifdef FUSED_PROF
__device__ long long fused_prof[2][64][8];
#define PROF(slot) do { if (blockIdx.x == 0 && lane == 0 && (wave == 0 || wave == 4) && prof_ph < 64) fused_prof[wave >> 2][prof_ph][slot] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define PROF(slot) do {} while (0)
#endif
// timing ablations for tools/fused_phase_timing.hip (results are wrong when defined)
#ifdef FUSED_NO_TRANSFORM
#define ABL_T(x) do {} while (0)
#else
#define ABL_T(x) x
#endif
#ifdef FUSED_NO_MFMA
#define MFMA_STEP(av, bv, cv) cv[0] += (av) + (bv)
#else
#define MFMA_STEP(av, bv, cv) cv = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, cv, 0, 0, 0)
#endif
#ifndef FUSED_BT_SHARED
#define FUSED_BT_SHARED 1          // 0: every output of B^T d as its own expression (the pre-round-3 form, for A/B builds)
#endif
#ifdef FUSED_NO_RAW
#define ABL_R(x) do {} while (0)
#else
#define ABL_R(x) x
#endif
// finer ablations of the patch path: only its global loads / only its LDS stores compiled out
#ifdef FUSED_NO_RAW_LOAD
#define ABL_RL(x) do {} while (0)
#else
#define ABL_RL(x) x
#endif
#ifdef FUSED_NO_RAW_STORE
#define ABL_RS(x) do {} while (0)
#else
#define ABL_RS(x) x
#endif

constexpr int F_TR = 4, F_TC = 8;                  // Winograd tiles per work item: rows x cols
constexpr int F_TILES = F_TR * F_TC;               // 32 = M of the MFMA tile
constexpr int F_PH = 4 * F_TR + 2, F_PW = 4 * F_TC + 2;   // input patch 18 x 34
constexpr int F_NPX = F_PH * F_PW;                 // 612
constexpr int F_KC = 8;                            // input channels per chunk
constexpr int F_PS = 10;                           // LDS floats per patch pixel: tile stride 40 floats, conflict-free
constexpr int F_RAW = 6 * F_TILES * 32;             // one patch buffer (612 px x 10 + dummy slot), sized to double as an
                                                   // exchange column in the epilogue
constexpr int F_V = 36 * F_TILES * F_KC;           // one V buffer: [p][tile][8], 16-byte halves XOR-swizzled by tile
constexpr int F_SX = 6 * F_TILES * 32;             // exchange buffer: one column of positions x 32 tiles x 32 couts
constexpr int F_LDS_FLOATS = 2 * F_V + 2 * F_RAW + F_SX;
constexpr int F_NT = 768;                          // threads: 12 waves, 3 positions each
constexpr int F_NLOAD = (F_NPX * 2 + F_NT - 1) / F_NT;    // float4 loads per thread per chunk (2)
static_assert(F_LDS_FLOATS * 4 <= 160 * 1024, "LDS");
static_assert(F_RAW >= F_NPX * F_PS + 4 && F_V >= F_SX, "epilogue columns alias the patch / V buffers");

__device__ __forceinline__ void bt6s(float (&d)[6]) {      // in-place B^T d (Lavin & Gray, points 0, +-1, +-2, inf)
  const float d0 = d[0], d1 = d[1], d2 = d[2], d3 = d[3], d4 = d[4], d5 = d[5];
  d[0] = 4.f * d0 - 5.f * d2 + d4;
  d[1] = -4.f * d1 - 4.f * d2 + d3 + d4;
  d[2] = 4.f * d1 - 4.f * d2 - d3 + d4;
  d[3] = -2.f * d1 - d2 + 2.f * d3 + d4;
  d[4] = 2.f * d1 - d2 - 2.f * d3 + d4;
  d[5] = 4.f * d1 - 5.f * d3 + d5;
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

struct ItemRef { int goff[F_NLOAD]; unsigned okm; int g; };

// in: (H, W, K) NHWC, K % 32 == 0; U: fragment-major (36, Cout, K) weights; out / mask: (H, W, Cout), Cout % 32 == 0.
// !MASK: out = relu ? max(Y + bias, 0) : Y + bias;  MASK: out = mask > 0 ? Y : 0;  ACC (with MASK): out += that (the taps of
// the layer were scattered into `out` before the backward pass: "pre-scatter", nn/model.py).
// Work item = (region r of 16x32 output pixels, row-major) x (group g of 32 couts): item = r * NG + g.  The grid is
// persistent; workgroup b walks the items of its XCD's contiguous range, so the NG groups of a region (same input
// patch) run on neighbouring CUs of one XCD at the same time and share its L2.
// The 36 products on the bf16 MFMA with the operands split exactly in registers (round 2: 353 us against 338 us for
// block1_conv2 forward, the splits' vector instructions take the place of the shorter MFMAs) and a pre-split-planes
// variant live in tools/experiments/ with their ablation tool; DESIGN.md 4 has the measurements.
template <bool MASK, bool ACC = false>
__global__ __launch_bounds__(F_NT) void winograd43_fused_kernel(const float* __restrict__ in, int H, int W, int K,
                                                                const float* __restrict__ U, int Cout,
                                                                const float* __restrict__ bias,
                                                                const float* __restrict__ mask, int relu,
                                                                float* __restrict__ out, float* __restrict__ pool, unsigned char* __restrict__ pool_code, int RW,
                                                                int NG, int nitems, const unsigned* __restrict__ bits_in,
                                                                unsigned* __restrict__ bits_out) {
  __shared__ __attribute__((aligned(16))) float lds[F_LDS_FLOATS];
  float* const Vb = lds;                            // V[2]
  float* const Rb = lds + 2 * F_V;                  // raw[2]
  float* const Sx = lds + 2 * F_V + 2 * F_RAW;      // exchange
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
  // ---- input transform, all 12 waves: thread = (tile, channel PAIR) of item t & 127 and ONE output row
  // r = t >> 7 (wave-uniform) of V = B^T d B.  On gfx950 the f32 MFMA and the VALU do not co-execute (measured:
  // tools/mfma_valu_overlap.hip -- a v_fma loop in one wave and a v_mfma_f32_32x32x2 loop in another wave of the
  // same SIMD take the SUM of their times), so every VALU instruction is paid in MFMA time: two channels per lane
  // make the whole transform packed-f32 (v_pk_fma/add/mul) without register shuffles, a third of the scalar count.
  auto transform = [&](const float* raw, float* V) {     // threads 0..255: (tile, channel pair, half of the rows)
    const int it = t & 127, half = (t >> 7) & 1;
    const int ttile = it >> 2, pr = it & 3;
    const int tty = ttile >> 3, ttx = ttile & 7;
    const float* src = raw + ((4 * tty) * F_PW + 4 * ttx) * F_PS + 2 * pr;
    auto ld = [&](int k, int q) { return *reinterpret_cast<const f32x2*>(src + (k * F_PW + q) * F_PS); };
    float* dst = V + (size_t)(18 * half) * (F_TILES * F_KC) + ttile * F_KC + ((((pr >> 1) ^ (ttile >> 3)) & 1) << 2) +
                 ((pr & 1) << 1);
    // B^T d with the common subexpressions written out (the compiler may not reassociate floating point):
    //   v1, v2 = (d4 - 4 d2) +- (d3 - 4 d1);   v3, v4 = (d4 - d2) +- 2 (d3 - d1):  12 packed operations instead of 16
    auto row_out = [&](int r, const f32x2 (&tr)[6]) {
      const f32x2 d0 = tr[0], d1 = tr[1], d2 = tr[2], d3 = tr[3], d4 = tr[4], d5 = tr[5];
      f32x2 v[6];
#if FUSED_BT_SHARED
      const f32x2 t0 = d4 - 4.f * d2, t1 = d3 - 4.f * d1, u0 = d4 - d2, u1 = d3 - d1;
      v[0] = 4.f * d0 - 5.f * d2 + d4;
      v[1] = t0 + t1;
      v[2] = t0 - t1;
      v[3] = u0 + 2.f * u1;
      v[4] = u0 - 2.f * u1;
      v[5] = 4.f * d1 - 5.f * d3 + d5;
#else
      v[0] = 4.f * d0 - 5.f * d2 + d4;
      v[1] = -4.f * d1 - 4.f * d2 + d3 + d4;
      v[2] = 4.f * d1 - 4.f * d2 - d3 + d4;
      v[3] = -2.f * d1 - d2 + 2.f * d3 + d4;
      v[4] = 2.f * d1 - d2 - 2.f * d3 + d4;
      v[5] = 4.f * d1 - 5.f * d3 + d5;
#endif
#pragma unroll
      for (int q = 0; q < 6; ++q) *reinterpret_cast<f32x2*>(dst + (6 * r + q) * (F_TILES * F_KC)) = v[q];
    };
    f32x2 ta[6], tb[6], tc[6];
    if (half == 0) {                                // rows 0, 1, 2 from input rows 0..4
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        const f32x2 d0 = ld(0, q), d1 = ld(1, q), d2 = ld(2, q), d3 = ld(3, q), d4 = ld(4, q);
        ta[q] = 4.f * d0 - 5.f * d2 + d4;
#if FUSED_BT_SHARED
        const f32x2 t0 = d4 - 4.f * d2, t1 = d3 - 4.f * d1;
        tb[q] = t0 + t1;
        tc[q] = t0 - t1;
#else
        tb[q] = -4.f * d1 - 4.f * d2 + d3 + d4;
        tc[q] = 4.f * d1 - 4.f * d2 - d3 + d4;
#endif
      }
    } else {                                        // rows 3, 4, 5 from input rows 1..5
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        const f32x2 d1 = ld(1, q), d2 = ld(2, q), d3 = ld(3, q), d4 = ld(4, q), d5 = ld(5, q);
#if FUSED_BT_SHARED
        const f32x2 u0 = d4 - d2, u1 = d3 - d1;
        ta[q] = u0 + 2.f * u1;
        tb[q] = u0 - 2.f * u1;
#else
        ta[q] = -2.f * d1 - d2 + 2.f * d3 + d4;
        tb[q] = 2.f * d1 - d2 - 2.f * d3 + d4;
#endif
        tc[q] = 4.f * d1 - 5.f * d3 + d5;
      }
    }
    row_out(0, ta); row_out(1, tb); row_out(2, tc);
  };
  // U fragments: positions wave, wave + 12, wave + 24; B = U[p][32 g + l31][8 kc + 4 hh ..] from the fragment-major
  // copy (strotss_conv3x3_winograd_pack): one fully coalesced 1 KB wave load per (position, group, chunk).  From the
  // (P, Cout, K) layout every load would touch 32 cache lines for 32 bytes each, and the per-XCD L2 request rate
  // (all 32 CUs stream the same weights) becomes the bound.
  const int nchunk = K / F_KC;                      // even, >= 4
  const size_t b_step = (size_t)12 * NG * nchunk * 256;
  auto load_u = [&](int g, int kc, f32x4 (&b)[3]) {
    const float* bp = U + (((size_t)wave * NG + g) * nchunk + kc) * 256 + lane * 4;
#pragma unroll
    for (int j = 0; j < 3; ++j) b[j] = *reinterpret_cast<const f32x4*>(bp + j * b_step);
  };
  // A = V[p][tile l31][4 (hh ^ swizzle) ..]
  const int a_off = (wave * F_TILES + l31) * F_KC + (((hh ^ (l31 >> 3)) & 1) << 2);
  constexpr int a_step = 12 * F_TILES * F_KC;

  int prof_ph = 0; (void)prof_ph;
  ItemRef cur, nxt;
  setup(item, cur);
  f32x4 S0[F_NLOAD], S1[F_NLOAD];                   // patches in flight: S[c & 1] is loaded in phase c with chunk c + 3
  f32x4 b0[3], b1[3];                              // U fragments: b0 for even chunks, b1 for odd ones
  // ---- prologue: raw[0] = chunk 0, raw[1] = chunk 1, V[0] = T(chunk 0), S1 = chunk 2 in flight, b = U(chunk 0)
  load_raw(cur, 0, S0);
  load_raw(cur, 1, S1);
  store_raw(Rb, cur.okm, S0);
  store_raw(Rb + F_RAW, cur.okm, S1);
  load_u(cur.g, 0, b0);
  load_raw(cur, 2, S1);
  __syncthreads();
  if (t < 256) transform(Rb, Vb);
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
#define PHASE(PAR, SLOAD, SSTORE, BCUR, BNEXT)                                                                            \
    {                                                                                                         \
      PROF(0);                                                                                                \
      CHUNK_AT(1, r1, k1) CHUNK_AT(3, r3, k3)                                                                 \
      const unsigned okm2 = (kc + 2 >= nchunk) ? nxt.okm : cur.okm;                                           \
      PROF(1);                                                                                                \
      const float* Vc = Vb + PAR * F_V + a_off;                                                               \
      f32x4 a[3];                                                                                             \
      /* the middle wave of each SIMD runs its MFMAs first, the other two transform first: the phase's VALU */ \
      /* and MFMA halves of different waves then overlap instead of all waves fighting for the same pipe  */   \
      if (t < 256) ABL_T(transform(Rb + (1 - PAR) * F_RAW, Vb + (1 - PAR) * F_V));                            \
      PROF(2);                                                                                                \
      _Pragma("unroll") for (int j = 0; j < 3; ++j) a[j] = *reinterpret_cast<const f32x4*>(Vc + j * a_step);  \
      _Pragma("unroll") for (int j = 0; j < 3; ++j)                                                           \
        _Pragma("unroll") for (int s = 0; s < 4; ++s)                                                         \
          MFMA_STEP(a[j][s], BCUR[j][s], acc[j]);                                                             \
      __builtin_amdgcn_sched_barrier(0);                                                                      \
      __builtin_amdgcn_sched_barrier(0);                                                                      \
      PROF(3);                                                                                                \
      /* next chunk's U fragments: issued AFTER this wave's MFMAs -- at the phase start their address arithmetic   */ \
      /* (64-bit scalar multiplies) and load issue kept every wave of the SIMD off the matrix pipe for ~250 cycles */ \
      load_u(r1.g, k1, BNEXT);                                                                                \
      /* the patch loads (32 cache lines per wave instruction) go here, not to the phase start: 24 of them */ \
      /* queued in the CU's address unit right after the barrier stall every wave behind them             */ \
      ABL_R(ABL_RL(load_raw(r3, k3, SLOAD)));                                                                 \
      __builtin_amdgcn_sched_barrier(0);                                                                      \
      ABL_R(ABL_RS(store_raw(Rb + PAR * F_RAW, okm2, SSTORE)));                                               \
      PROF(4);                                                                                                \
      __syncthreads();                                                                                        \
      PROF(5);                                                                                                \
      ++prof_ph;                                                                                              \
      ++kc;                                                                                                   \
    }
    for (int kc = 0; kc < nchunk;) {
      PHASE(0, S0, S1, b0, b1)
      PHASE(1, S1, S0, b1, b0)
    }
#undef PHASE
#undef CHUNK_AT

    PROF(0);
    // ---- output transform by columns q of the 6x6 positions: waves q and q + 6 hold its rows {0,2,4} / {1,3,5}
    const int region = item / NG;
    const int ry = region / RW, rx = region - ry * RW;
    const int y0 = ry * (4 * F_TR), x0 = rx * (4 * F_TC);
    // Column arithmetic and stores: thread t < 512 owns tile t >> 4 and the TWO adjacent output channels 2 (t & 15), +1 --
    // its LDS reads, mask / sign-word loads and stores are 8 bytes wide (half the instructions of one channel per thread
    // and two tiles: the store issue, ~7 cycles per wave instruction, was the longest part of the epilogue)
    const int etile = (t >> 4) & 31, ecp = t & 15;
    float Y[2][4][4];                               // [channel of the pair]; starts at the bias (none for the data-gradient)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      float bv = 0.f;
      if constexpr (!MASK) { if (bias && t < 512) bv = bias[cur.g * 32 + 2 * ecp + i]; }
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int q = 0; q < 4; ++q) Y[i][r][q] = bv;
    }
    // Three columns per round: besides Sx, V[1] (read by the last phase's MFMAs) and raw[0] (transformed in the
    // last phase) are free here -- nchunk is even, so an item always ends on an odd phase.
    const int myq = wave % 6, myr = wave / 6;
    float* const colbuf[3] = {Sx, Vb + F_V, Rb};
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
      if (round == 0) PROF(1); else PROF(3);
      if (t < 512) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const int q = 3 * round + k;
          const float* src = colbuf[k] + etile * 32 + 2 * ecp;           // [row][tile][cout]: this thread's channel pair
          f32x2 m[6];
#pragma unroll
          for (int r = 0; r < 6; ++r) m[r] = *reinterpret_cast<const f32x2*>(src + r * (F_TILES * 32));
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            float sv[4];
            sv[0] = m[0][i] + m[1][i] + m[2][i] + m[3][i] + m[4][i];
            sv[1] = m[1][i] - m[2][i] + 2.f * m[3][i] - 2.f * m[4][i];
            sv[2] = m[1][i] + m[2][i] + 4.f * m[3][i] + 4.f * m[4][i];
            sv[3] = m[1][i] - m[2][i] + 8.f * m[3][i] - 8.f * m[4][i] + m[5][i];
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
      if (round == 0) PROF(2); else PROF(4);
    }
    if (t < 512) {
      const float lo = (!MASK && relu) ? 0.f : -INFINITY;
      const int ty = etile >> 3, tx = etile & 7;
      const int co = cur.g * 32 + 2 * ecp;                   // even: every access below is 8-byte aligned (Cout is even)
      const int yb = y0 + 4 * ty, xb = x0 + 4 * tx;
      // Every memory-dependent value (bias: folded into Y before the rounds; ReLU mask: one batch of clamped
      // loads condensed to a bit mask) is resolved before the first store, so the stores carry no s_waitcnt --
      // with a load pending the compiler puts vmcnt(0) in front of each predicated store and they serialise.
      const bool tile_in = yb < H && xb < W, full = yb + 3 < H && xb + 3 < W;
      const size_t ob = tile_in ? ((size_t)yb * W + xb) * Cout + co : (size_t)co;
      float* op = out + ob;
      unsigned keep[2] = {0xffffu, 0xffffu};
      // sign words of the tile grid (include/strotss_hip.h: relu_bits): one word per (tile, channel), byte r = row r
      const size_t bo = ((size_t)(yb >> 2) * ((W + 3) >> 2) + (xb >> 2)) * Cout + co;
      if (MASK && bits_in) {                       // 8 bytes instead of 2 x 16 activations
        const u32x2 wv = *reinterpret_cast<const u32x2*>(bits_in + (tile_in ? bo : 0));
#pragma unroll
        for (int i = 0; i < 2; ++i)
          keep[i] = (wv[i] & 0xfu) | ((wv[i] >> 4) & 0xf0u) | ((wv[i] >> 8) & 0xf00u) | ((wv[i] >> 12) & 0xf000u);
      } else if constexpr (MASK) {
        const float* mp = mask + ob;
        f32x2 mk[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const bool ok = tile_in && yb + r < H && xb + c < W;
            mk[r][c] = *reinterpret_cast<const f32x2*>(mp + (ok ? (r * W + c) * Cout : 0));
          }
        keep[0] = keep[1] = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            keep[0] |= (mk[r][c][0] > 0.f ? 1u : 0u) << (4 * r + c);
            keep[1] |= (mk[r][c][1] > 0.f ? 1u : 0u) << (4 * r + c);
          }
      }
      f32x2 old[ACC ? 4 : 1][ACC ? 4 : 1];            // ACC: what the output holds, loaded (clamped) before the first store
      if constexpr (ACC) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const bool ok = tile_in && yb + r < H && xb + c < W;
            old[r][c] = *reinterpret_cast<const f32x2*>(op + (ok ? (r * W + c) * Cout : 0));
          }
      }
      auto outv = [&](int r, int c) {
        f32x2 v;
        v[0] = ((keep[0] >> (4 * r + c)) & 1u) ? fmaxf(Y[0][r][c], lo) : 0.f;
        v[1] = ((keep[1] >> (4 * r + c)) & 1u) ? fmaxf(Y[1][r][c], lo) : 0.f;
        if constexpr (ACC) { v[0] += old[r][c][0]; v[1] += old[r][c][1]; }
        return v;
      };
      if (full) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int c = 0; c < 4; ++c) *reinterpret_cast<f32x2*>(op + (r * W + c) * Cout) = outv(r, c);
      } else if (tile_in) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int c = 0; c < 4; ++c)
            if (yb + r < H && xb + c < W) *reinterpret_cast<f32x2*>(op + (r * W + c) * Cout) = outv(r, c);
      }
      if constexpr (!MASK) {
        if (bits_out && tile_in) {
          u32x2 wv = {0u, 0u};
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              wv[0] |= (Y[0][r][c] > 0.f ? 1u : 0u) << (8 * r + c);
              wv[1] |= (Y[1][r][c] > 0.f ? 1u : 0u) << (8 * r + c);
            }
          *reinterpret_cast<u32x2*>(bits_out + bo) = wv;
        }
        // fused 2x2/2 max-pool of the activations just written (the tile's 4x4 outputs hold 2x2 windows);
        // windows are emitted only where they lie inside the image (floor pooling)
        if (pool) {
          const int PH = H >> 1, PW = W >> 1, py0 = yb >> 1, px0 = xb >> 1;
#pragma unroll
          for (int pr = 0; pr < 2; ++pr)
#pragma unroll
            for (int pc = 0; pc < 2; ++pc)
              if (py0 + pr < PH && px0 + pc < PW) {
                f32x2 pv;
                unsigned short code2 = 0;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                  const float v4[4] = {Y[i][2 * pr][2 * pc], Y[i][2 * pr][2 * pc + 1], Y[i][2 * pr + 1][2 * pc],
                                       Y[i][2 * pr + 1][2 * pc + 1]};
                  int best = 0;
                  float bv = v4[0];
#pragma unroll
                  for (int q4 = 1; q4 < 4; ++q4)
                    if (v4[q4] > bv) { bv = v4[q4]; best = q4; }
                  pv[i] = fmaxf(bv, lo);
                  code2 |= (unsigned short)((bv > 0.f ? best : 4) << (8 * i));   // argmax code of maxpool2_fwd
                }
                const size_t po = ((size_t)(py0 + pr) * PW + px0 + pc) * Cout + co;
                *reinterpret_cast<f32x2*>(pool + po) = pv;
                if (pool_code) *reinterpret_cast<unsigned short*>(pool_code + po) = code2;
              }
        }
      }
    }
    PROF(5); PROF(6); ++prof_ph;
    if (!more) break;
    item = next;
    cur = nxt;
  }
}

__global__ __launch_bounds__(256) void winograd43_pack_kernel(const float* __restrict__ u, int rows, int k, size_t total,
                                                             float* __restrict__ up) {
  const int ng = rows / 32, nc = k / F_KC;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const int c = (int)(e % k);
    const size_t pr = e / k;
    const int r = (int)(pr % rows), p = (int)(pr / rows);
    const size_t o = (((((size_t)p * ng + r / 32) * nc + c / 8) * 2 + (c % 8) / 4) * 32 + r % 32) * 4 + c % 4;
    up[o] = u[e];
  }
}

}  // namespace

int st_winograd43_pack(const float* u_prk, int rows, int k, float* u_packed, hipStream_t st) {
  const size_t total = (size_t)36 * rows * k;
  hipLaunchKernelGGL(winograd43_pack_kernel, dim3((unsigned)min((size_t)4096, (total + 255) / 256)), dim3(256), 0, st,
                     u_prk, rows, k, total, u_packed);
  ST_LAUNCH_RET();
}

// Policy (measured per layer, tools/conv_bench.py): the fused kernel wins where the three-kernel form is bound by
// its transform traffic -- up to 256 output channels -- and there are at least 128 work items (measured at the
// 256 / 512 px scales: 128 and 256 beat 512 and 64 by 3-5 % of the step; STROTSS_WINO_FUSED_MIN_ITEMS overrides).
// STROTSS_WINO_FUSED: 0 = never, 1 (default) = that policy, 2 = every layer it supports.
bool st_winograd43_fused_enabled(int h, int w, int cout) {
  static int on = -1;
  if (on < 0) { const char* e = getenv("STROTSS_WINO_FUSED"); on = e ? atoi(e) : 1; }
  if (on == 0 || cout % 32 != 0) return false;
  if (on >= 2) return true;
  const int TH = (h + 3) / 4, TW = (w + 3) / 4;
  const long items = (long)((TH + F_TR - 1) / F_TR) * ((TW + F_TC - 1) / F_TC) * (cout / 32);
  static long min_items = -1, max_cout = 256;
  if (min_items < 0) {
    const char* e = getenv("STROTSS_WINO_FUSED_MIN_ITEMS"); min_items = e ? atol(e) : 128;
    const char* c = getenv("STROTSS_WINO_FUSED_MAX_COUT"); if (c) max_cout = atol(c);
  }
  return cout <= max_cout && items >= min_items;
}

int st_winograd43_fused(const float* in, int h, int w, int cin, const float* U, const float* bias, int cout,
                        const float* mask, int relu, float* out, float* pool_out, unsigned char* pool_code,
                        const unsigned* bits_in, unsigned* bits_out, hipStream_t st, int accumulate) {
  if (cin % 32 != 0 || cout % 32 != 0) return STROTSS_EALIGN;
  if (accumulate && !(mask || bits_in)) return STROTSS_EINVAL;       // only the data-gradient form adds to its output
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
  int grid = cus;                                  // persistent: one 147 KB-LDS workgroup per CU, a multiple of 8
  while (grid > 8 && grid / 2 >= nitems) grid /= 2;
#define LAUNCH_FUSED(...) hipLaunchKernelGGL((winograd43_fused_kernel<__VA_ARGS__>), dim3((unsigned)grid), dim3(F_NT), 0, st, in, h, w, \
                                             cin, U, cout, bias, mask, relu, out, pool_out, pool_code, RW, NG, nitems, bits_in, bits_out)
  if (accumulate) LAUNCH_FUSED(true, true);
  else if (mask || bits_in) LAUNCH_FUSED(true);
  else LAUNCH_FUSED(false);
#undef LAUNCH_FUSED
  ST_LAUNCH_RET();
}
