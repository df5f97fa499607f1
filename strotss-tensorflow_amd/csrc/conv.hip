// VGG16 trunk kernels (nn/model.py:44-55): 3x3 SAME conv + bias + ReLU as an implicit GEMM on the
// fp32 MFMA tile engine (forward and data-gradient; the net is frozen so there is no weight
// gradient), the 3-channel first layer and its pixel gradient, and 2x2 max-pool fwd/bwd.
// Layout NHWC: a pixel's channels are contiguous, so the GEMM's A operand (pixels x Cin-chunk)
// is gathered as whole 128-byte channel segments and K = (tap, ci) needs no im2col buffer.
#include <stdlib.h>

#include "internal.h"
#include "mfma_pipe.h"

typedef float f32x2 __attribute__((ext_vector_type(2)));

namespace {

// out[p, co] = epilogue( sum_{tap,ci} in[p + tap offset, ci] * wt[tap][co][ci] )
//   fwd  : + bias[co], ReLU                        (bias != null, relu = 1, mask = null)
//   dgrad: * (mask[p,co] > 0) when mask != null    (bias = null, relu = 0)
template <int BM, int BN>
__global__ __launch_bounds__(256) void conv3x3_mfma_kernel(const float* __restrict__ in, int H, int W,
                                                           int Cin, const float* __restrict__ wt,
                                                           const float* __restrict__ bias, int Cout,
                                                           const float* __restrict__ mask,
                                                           float* __restrict__ out, int relu) {
  __shared__ __attribute__((aligned(16))) float lds[OperandLds<BM>::kc_floats + OperandLds<BN>::kc_floats];
  float* ldsA = lds;
  float* ldsB = lds + OperandLds<BM>::kc_floats;
  const int HW = H * W;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int t = threadIdx.x, c4 = t & 7, r0 = t >> 3;

  // pixel coordinates of the rows this thread stages
  int py[BM / 32], px[BM / 32];
#pragma unroll
  for (int i = 0; i < BM / 32; ++i) {
    const int p = m0 + r0 + 32 * i;
    if (p < HW) { py[i] = p / W; px[i] = p - py[i] * W; }
    else { py[i] = -4; px[i] = -4; }   // always out of bounds -> zero rows
  }
  const int kchunks = Cin >> 5;
  const int steps = 9 * kchunks;

  f32x4 ra[BM / 32], rb[BN / 32];
  auto load = [&](int s) {
    const int tap = s / kchunks;
    const int ci0 = (s - tap * kchunks) << 5;
    const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
#pragma unroll
    for (int i = 0; i < BM / 32; ++i) {
      const int yy = py[i] + dy, xx = px[i] + dx;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (yy >= 0 && yy < H && xx >= 0 && xx < W)
        v = *reinterpret_cast<const f32x4*>(&in[((size_t)yy * W + xx) * Cin + ci0 + c4 * 4]);
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < BN / 32; ++i) {
      const int co = n0 + r0 + 32 * i;   // Cout % BN == 0 (checked by the launcher)
      rb[i] = *reinterpret_cast<const f32x4*>(&wt[((size_t)tap * Cout + co) * Cin + ci0 + c4 * 4]);
    }
  };

  f32x16 acc[BM / 64][BN / 64];
  acc_zero<BM, BN>(acc);
  load(0);
  for (int s = 0; s < steps; ++s) {
    __syncthreads();
    lds_store_kc<BM>(ldsA, ra);
    lds_store_kc<BN>(ldsB, rb);
    __syncthreads();
    if (s + 1 < steps) load(s + 1);
    mma_kstep<BM, BN, true, true>(ldsA, ldsB, acc);
  }

  AccMap<BM, BN> map;
#pragma unroll
  for (int in_ = 0; in_ < BN / 64; ++in_) {
    const int co = n0 + map.colof(in_);
    const float b = bias ? bias[co] : 0.f;
#pragma unroll
    for (int im = 0; im < BM / 64; ++im)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int p = m0 + map.row(im, reg);
        if (p < HW) {
          float v = acc[im][in_][reg] + b;
          if (relu) v = fmaxf(v, 0.f);
          const size_t o = (size_t)p * Cout + co;
          if (mask) v = (mask[o] > 0.f) ? v : 0.f;
          out[o] = v;
        }
      }
  }
}

// ---- pipelined kernel (mfma_pipe.h): implicit-GEMM loaders for the 3x3 gather -----------------
template <class Cfg>
struct ConvALoader {   // A(pixel, k) = in[pixel + tap offset][ci]; out-of-image taps are zeros
  const float* in;
  int base[Cfg::NA];
  unsigned tapmask[Cfg::NA];
  f32x4 r[2][Cfg::NA];
  unsigned okbits[2];
  int tap, ci0, off, Cin, W;
  __device__ __forceinline__ ConvALoader(const float* in_, int H, int W_, int Cin_, int m0) : in(in_), tap(0), ci0(0), Cin(Cin_), W(W_) {
    okbits[0] = okbits[1] = 0;
    const int c4 = threadIdx.x & 7, r0 = threadIdx.x >> 3;
    const int HW = H * W_;
#pragma unroll
    for (int i = 0; i < Cfg::NA; ++i) {
      const int p = m0 + r0 + Cfg::RPP * i;
      unsigned mk = 0;
      int pc = 0;
      if (p < HW) {
        pc = p;
        const int y = p / W_, x = p - y * W_;
#pragma unroll
        for (int tp = 0; tp < 9; ++tp) {
          const int yy = y + tp / 3 - 1, xx = x + tp % 3 - 1;
          if (yy >= 0 && yy < H && xx >= 0 && xx < W_) mk |= 1u << tp;
        }
      }
      base[i] = pc * Cin_ + c4 * 4;
      tapmask[i] = mk;
    }
    off = (-W_ - 1) * Cin_;
  }
  __device__ __forceinline__ void seek(int step) {          // start at K-step `step` = (tap, 32-channel chunk)
    const int kch = Cin >> 5;
    tap = step / kch; ci0 = (step - tap * kch) << 5;
    off = ((tap / 3 - 1) * W + (tap - (tap / 3) * 3 - 1)) * Cin + ci0;
  }
  __device__ __forceinline__ void issue(int i, int slot) {
    const bool ok = (tapmask[i] >> tap) & 1u;
    r[slot][i] = *reinterpret_cast<const f32x4*>(in + (ok ? base[i] + off : base[i]));
    okbits[slot] = (okbits[slot] & ~(1u << i)) | ((unsigned)ok << i);
  }
  __device__ __forceinline__ f32x4 value(int i, int slot) const {
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    return ((okbits[slot] >> i) & 1u) ? r[slot][i] : z4;
  }
  __device__ __forceinline__ void advance() {
    ci0 += 32; off += 32;
    if (ci0 == Cin) {
      ci0 = 0; ++tap;
      if (tap == 9) tap = 0;      // past the end: wrap, the (unused) extra loads stay in bounds
      off = ((tap / 3 - 1) * W + (tap - (tap / 3) * 3 - 1)) * Cin;
    }
  }
};
template <class Cfg>
struct ConvBLoader {   // B(co, k) = wt[tap][co][ci]
  const float* wt;
  int base[Cfg::NB];
  f32x4 r[2][Cfg::NB];
  int tap, ci0, off, Cin, CoutCin;
  __device__ __forceinline__ ConvBLoader(const float* wt_, int Cin_, int Cout, int n0) : wt(wt_), tap(0), ci0(0), off(0), Cin(Cin_), CoutCin(Cout * Cin_) {
    const int c4 = threadIdx.x & 7, r0 = threadIdx.x >> 3;
#pragma unroll
    for (int i = 0; i < Cfg::NB; ++i) base[i] = (n0 + r0 + Cfg::RPP * i) * Cin_ + c4 * 4;
  }
  __device__ __forceinline__ void seek(int step) {
    const int kch = Cin >> 5;
    tap = step / kch; ci0 = (step - tap * kch) << 5;
    off = tap * CoutCin + ci0;
  }
  __device__ __forceinline__ void issue(int i, int slot) { r[slot][i] = *reinterpret_cast<const f32x4*>(wt + base[i] + off); }
  __device__ __forceinline__ f32x4 value(int i, int slot) const { return r[slot][i]; }
  __device__ __forceinline__ void advance() {
    ci0 += 32; off += 32;
    if (ci0 == Cin) { ci0 = 0; ++tap; if (tap == 9) tap = 0; off = tap * CoutCin; }
  }
};

template <class Cfg>
__global__ __launch_bounds__(Cfg::NT) void conv3x3_mfma_pipe_kernel(const float* __restrict__ in, int H, int W,
                                                                    int Cin, const float* __restrict__ wt,
                                                                    const float* __restrict__ bias, int Cout,
                                                                    const float* __restrict__ mask,
                                                                    float* __restrict__ out, int relu, int accumulate) {
  __shared__ __attribute__((aligned(16))) float lds[Cfg::LDS_FLOATS];
  const int HW = H * W;
  const unsigned gx = Cout / Cfg::BN;                       // 1-D launch, XCD-aware tile order (N-tile fastest)
  const unsigned tile = xcd_swizzle(blockIdx.x, gridDim.x);
  const int m0 = (tile / gx) * Cfg::BM, n0 = (tile % gx) * Cfg::BN;
  ConvALoader<Cfg> la(in, H, W, Cin, m0);
  ConvBLoader<Cfg> lb(wt, Cin, Cout, n0);
  f32x16 acc[Cfg::TM][Cfg::TN];
#pragma unroll
  for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
    for (int j = 0; j < Cfg::TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  pipe_mainloop<Cfg>(lds, 9 * (Cin >> 5), la, lb, acc);

  PipeAccMap<Cfg> map;
#pragma unroll
  for (int in_ = 0; in_ < Cfg::TN; ++in_) {
    const int co = n0 + map.colof(in_);
    const float b = bias ? bias[co] : 0.f;
#pragma unroll
    for (int im = 0; im < Cfg::TM; ++im) {
      float mv[16];
      if (mask) {      // uniform branch; the 16 loads inside are issued back to back, clamped in bounds
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const int p = min(m0 + map.row(im, reg), HW - 1);
          mv[reg] = mask[(size_t)p * Cout + co];
        }
      } else {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) mv[reg] = 1.f;
      }
      float ov[16];
      if (accumulate) {  // (data-gradient into a buffer that already holds the layer's scattered taps: loads first, as the mask's)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const int p = min(m0 + map.row(im, reg), HW - 1);
          ov[reg] = out[(size_t)p * Cout + co];
        }
      } else {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) ov[reg] = 0.f;
      }
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int p = m0 + map.row(im, reg);
        if (p < HW) {
          float v = acc[im][in_][reg] + b;
          if (relu) v = fmaxf(v, 0.f);
          v = (mv[reg] > 0.f) ? v : 0.f;
          out[(size_t)p * Cout + co] = accumulate ? ov[reg] + v : v;
        }
      }
    }
  }
}

// ---- split-K form for SMALL maps (the 64-256 px scales: 4x4 ... 32x32 pixel maps).  There the whole layer is a few
// 64 x 64 output tiles and the three-kernel Winograd form is three dependent, latency-bound launches of 5-9 us each;
// here `nsplit` workgroups per tile take consecutive ranges of the 9 * Cin / 32 K-steps (a 512 -> 512 layer on 4 x 4
// pixels: 8 tiles x 32 splits = 256 workgroups, each streams 37 KB of weights), write their partial tiles, and the
// finish kernel adds them in split order (fixed: bitwise reproducible) with bias / ReLU / mask.
template <class Cfg>
__global__ __launch_bounds__(Cfg::NT) void conv3x3_mfma_splitk_kernel(const float* __restrict__ in, int H, int W, int Cin,
                                                                      const float* __restrict__ wt, int Cout, int nsplit,
                                                                      float* __restrict__ part) {
  __shared__ __attribute__((aligned(16))) float lds[Cfg::LDS_FLOATS];
  const int HW = H * W;
  const unsigned gx = Cout / Cfg::BN, gy = (HW + Cfg::BM - 1) / Cfg::BM, ntile = gx * gy;
  const unsigned id = xcd_swizzle(blockIdx.x, gridDim.x);
  const unsigned sp = id / ntile, tile = id - sp * ntile;       // split slowest: an XCD's run shares one K range
  const int m0 = (tile / gx) * Cfg::BM, n0 = (tile % gx) * Cfg::BN;
  const int S = 9 * (Cin >> 5);
  const int s0 = (int)((long long)S * sp / nsplit), s1 = (int)((long long)S * (sp + 1) / nsplit);
  ConvALoader<Cfg> la(in, H, W, Cin, m0);
  ConvBLoader<Cfg> lb(wt, Cin, Cout, n0);
  la.seek(s0); lb.seek(s0);
  f32x16 acc[Cfg::TM][Cfg::TN];
#pragma unroll
  for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
    for (int j = 0; j < Cfg::TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  pipe_mainloop<Cfg>(lds, s1 - s0, la, lb, acc);
  PipeAccMap<Cfg> map;
  float* P = part + (size_t)sp * HW * Cout;
#pragma unroll
  for (int in_ = 0; in_ < Cfg::TN; ++in_)
#pragma unroll
    for (int im = 0; im < Cfg::TM; ++im)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int p = m0 + map.row(im, reg);
        if (p < HW) P[(size_t)p * Cout + n0 + map.colof(in_)] = acc[im][in_][reg];
      }
}
// out = epilogue(sum_s part[s]): + bias, ReLU (forward) or * (mask > 0) (data-gradient); 4 channels per thread
__global__ __launch_bounds__(256) void conv_splitk_finish_kernel(const f32x4* __restrict__ part, int nsplit, size_t total4,
                                                                 int cout4, const f32x4* __restrict__ bias,
                                                                 const f32x4* __restrict__ mask, int relu,
                                                                 f32x4* __restrict__ out, int accumulate) {
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total4; e += (size_t)gridDim.x * 256) {
    // the partial tiles are added in split order (bitwise reproducible), eight loads in flight at a time: a thread's chain
    // of up to 72 dependent L2 round trips was the whole 9 us of this kernel on the 8 x 8 and 4 x 4 maps
    f32x4 v = part[e];
    int sp = 1;
    for (; sp + 8 <= nsplit; sp += 8) {
      f32x4 p8[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) p8[u] = part[(size_t)(sp + u) * total4 + e];
#pragma unroll
      for (int u = 0; u < 8; ++u) v = v + p8[u];
    }
    for (; sp < nsplit; ++sp) v = v + part[(size_t)sp * total4 + e];
    if (bias) v = v + bias[e % cout4];
    if (relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
    if (mask) {
      const f32x4 m = mask[e];
      v[0] = m[0] > 0.f ? v[0] : 0.f; v[1] = m[1] > 0.f ? v[1] : 0.f; v[2] = m[2] > 0.f ? v[2] : 0.f; v[3] = m[3] > 0.f ? v[3] : 0.f;
    }
    out[e] = accumulate ? out[e] + v : v;
  }
}
// The same finish for a layer that is followed by the 2x2/2 max-pool (block ends): thread = one pooled element (4 channels) --
// it finishes the four pixels of its window (partial tiles added in split order, + bias, ReLU: the bits of the plain finish),
// stores them, and writes their maximum and its argmax code exactly as maxpool2_fwd_kernel would from `out`.  An odd last
// row / column lies in no window: finished and stored, not pooled.  One launch less per block at the split-K scales.
__global__ __launch_bounds__(256) void conv_splitk_finish_pool_kernel(const f32x4* __restrict__ part, int nsplit, size_t total4,
                                                                      int H, int W, int C4, const f32x4* __restrict__ bias,
                                                                      f32x4* __restrict__ out, f32x4* __restrict__ pool,
                                                                      unsigned* __restrict__ code) {
  const int Ho = H >> 1, Wo = W >> 1, Wx = Wo + (W & 1), rows = Ho + (H & 1);
  const int row_elems = Wx * C4;
  for (int oy = blockIdx.y; oy < rows; oy += gridDim.y)
    for (int er = blockIdx.x * 256 + threadIdx.x; er < row_elems; er += gridDim.x * 256) {
      const int ox = er / C4, c = er - ox * C4;
      size_t o[4];
      bool ok[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {                                // q = (dy << 1) | dx
        const int y = 2 * oy + (q >> 1), x = 2 * ox + (q & 1);
        ok[q] = y < H && x < W;
        o[q] = ok[q] ? ((size_t)y * W + x) * C4 + c : (size_t)c;   // clamped: loaded unconditionally, stored if ok
      }
      f32x4 v[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = part[o[q]];
      int sp = 1;
      for (; sp + 4 <= nsplit; sp += 4) {                          // 4 pixels x 4 partial tiles in flight, split order per pixel
        f32x4 p4[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int q = 0; q < 4; ++q) p4[q][u] = part[(size_t)(sp + u) * total4 + o[q]];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] = v[q] + p4[q][u];
      }
      for (; sp < nsplit; ++sp)
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = v[q] + part[(size_t)sp * total4 + o[q]];
      const f32x4 b = bias[c];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        v[q] = v[q] + b;
        v[q][0] = fmaxf(v[q][0], 0.f); v[q][1] = fmaxf(v[q][1], 0.f); v[q][2] = fmaxf(v[q][2], 0.f); v[q][3] = fmaxf(v[q][3], 0.f);
        if (ok[q]) out[o[q]] = v[q];
      }
      if (oy < Ho && ox < Wo) {
        f32x4 m;
        unsigned packed = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          m[k] = fmaxf(fmaxf(v[0][k], v[1][k]), fmaxf(v[2][k], v[3][k]));
          int best = 0;
          float bv = v[0][k];
#pragma unroll
          for (int q = 1; q < 4; ++q)
            if (v[q][k] > bv) { bv = v[q][k]; best = q; }
          packed |= (unsigned)(bv > 0.f ? best : 4) << (8 * k);
        }
        const size_t e = ((size_t)oy * Wo + ox) * C4 + c;
        pool[e] = m;
        if (code) code[e] = packed;
      }
    }
}
// The finish of a data-gradient whose layer's input came from the 2x2/2 max-pool, written THROUGH the pool's adjoint: thread =
// one pooled element, v = its partial tiles added in split order (what the plain finish would store as the pooled gradient),
// then the 2 x 2 window of the (H, W) gradient of the layer in front of the pool from the argmax codes, as
// maxpool2_bwd_code_kernel does -- the pooled gradient is never stored, one launch less per block.
__global__ __launch_bounds__(256) void conv_splitk_finish_unpool_kernel(const f32x4* __restrict__ part, int nsplit, size_t total4,
                                                                        int H, int W, int C4, const unsigned* __restrict__ code,
                                                                        f32x4* __restrict__ gin, int accumulate) {
  const int Ho = H >> 1, Wo = W >> 1, Wx = Wo + (W & 1);
  const int row_elems = Wx * C4;
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  for (int oy = blockIdx.y; oy < Ho + (H & 1); oy += gridDim.y) {
    const bool pooled_row = oy < Ho;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < row_elems; e += gridDim.x * 256) {
      const int ox = e / C4, c = e - ox * C4;
      const bool pooled = pooled_row && ox < Wo;
      f32x4 r[4] = {z, z, z, z};
      if (pooled) {
        const size_t po = ((size_t)oy * Wo + ox) * C4 + c;
        const unsigned cd = code[po];
        f32x4 go = part[po];
        int sp = 1;
        for (; sp + 8 <= nsplit; sp += 8) {
          f32x4 p8[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) p8[u] = part[(size_t)(sp + u) * total4 + po];
#pragma unroll
          for (int u = 0; u < 8; ++u) go = go + p8[u];
        }
        for (; sp < nsplit; ++sp) go = go + part[(size_t)sp * total4 + po];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const unsigned me = (cd >> (8 * k)) & 0xffu;
#pragma unroll
          for (int q = 0; q < 4; ++q) r[q][k] = me == (unsigned)q ? go[k] : 0.f;
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int y = 2 * oy + (q >> 1), x = 2 * ox + (q & 1);
        if (y < H && x < W) {
          const size_t o = ((size_t)y * W + x) * C4 + c;
          gin[o] = accumulate ? gin[o] + r[q] : r[q];
        }
      }
    }
  }
}
// number of K splits for a (h, w, cin, cout) layer, 0 = one-pass kernel: only when the layer is at most 128 tiles of 64 x 64
static int conv_splits(int H, int W, int Cin, int Cout) {
  static int on = -1;
  if (on < 0) { const char* e = getenv("STROTSS_CONV_SPLITK"); on = e ? atoi(e) : 1; }
  const long tiles = (long)cdiv((int64_t)H * W, 64) * (Cout / 64);
  const int S = 9 * (Cin >> 5);
  if (!on || tiles > 128) return 0;
  int n = (int)(256 / tiles);
  if (n > S / 2) n = S / 2;                          // at least two K-steps per workgroup
  return n >= 2 ? n : 0;
}

static int conv_variant() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("STROTSS_CONV_VARIANT");
    v = e ? atoi(e) : 3;   // 1 = simple two-barrier kernel (kept for A/B), otherwise the pipelined one
  }
  return v;
}

static int conv_waves() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("STROTSS_CONV_WAVES");
    v = e ? atoi(e) : 4;
  }
  return v;
}

template <int BM, int BN>
int launch_conv(const float* in, int H, int W, int Cin, const float* wt, const float* bias, int Cout,
                const float* mask, float* out, int relu, hipStream_t s, int accumulate) {
  dim3 grid(Cout / BN, cdiv((int64_t)H * W, BM));
  if (conv_variant() == 1) {
    if (accumulate) return STROTSS_EINVAL;             // (the two-barrier A/B kernel overwrites)
    hipLaunchKernelGGL((conv3x3_mfma_kernel<BM, BN>), grid, dim3(256), 0, s, in, H, W, Cin, wt, bias, Cout,
                       mask, out, relu);
    ST_LAUNCH_RET();
  }
  const dim3 grid1(grid.x * grid.y);
  if constexpr (BM == 128) {
    if (conv_waves() == 8) {
      using Cfg = PipeCfg<BM, BN, (BN == 128 ? 2 : 4), (BN == 128 ? 4 : 2)>;
      hipLaunchKernelGGL((conv3x3_mfma_pipe_kernel<Cfg>), grid1, dim3(Cfg::NT), 0, s, in, H, W, Cin, wt, bias, Cout,
                         mask, out, relu, accumulate);
      ST_LAUNCH_RET();
    }
  }
  {
    using Cfg = PipeCfg<BM, BN, 2, 2>;
    hipLaunchKernelGGL((conv3x3_mfma_pipe_kernel<Cfg>), grid1, dim3(Cfg::NT), 0, s, in, H, W, Cin, wt, bias, Cout,
                       mask, out, relu, accumulate);
  }
  ST_LAUNCH_RET();
}

// what the finish of a split-K layer also does at a block end (the pooling launch it replaces): forward -- the pooled copy
// and its argmax codes; data-gradient -- the result goes through the pool's adjoint into the (2h.., 2w..) gradient in front
struct FinishPool {
  float* pool = nullptr; unsigned char* code = nullptr;                               // forward
  const unsigned char* ucode = nullptr; float* ugin = nullptr; int uH = 0, uW = 0;    // data-gradient
};

int conv_dispatch(const float* in, int H, int W, int Cin, const float* wt, const float* bias, int Cout,
                  const float* mask, float* out, int relu, hipStream_t s, void* workspace = nullptr, size_t workspace_bytes = 0,
                  int accumulate = 0, const FinishPool* fp = nullptr) {
  const int64_t M = (int64_t)H * W;
  const int nsplit = conv_splits(H, W, Cin, Cout);
  if (nsplit && workspace && workspace_bytes >= (size_t)nsplit * M * Cout * sizeof(float)) {
    using Cfg = PipeCfg<64, 64, 2, 2>;
    const unsigned tiles = (unsigned)(cdiv(M, 64) * (Cout / 64));
    hipLaunchKernelGGL((conv3x3_mfma_splitk_kernel<Cfg>), dim3(tiles * nsplit), dim3(Cfg::NT), 0, s, in, H, W, Cin, wt, Cout,
                       nsplit, (float*)workspace);
    const size_t total4 = (size_t)M * Cout / 4;
    if (fp && fp->pool) {
      const int C4 = Cout / 4, rows = (H >> 1) + (H & 1), row_elems = ((W >> 1) + (W & 1)) * C4;
      hipLaunchKernelGGL(conv_splitk_finish_pool_kernel, dim3((unsigned)min(64, cdiv(row_elems, 256)), (unsigned)rows), dim3(256), 0, s,
                         (const f32x4*)workspace, nsplit, total4, H, W, C4, (const f32x4*)bias, (f32x4*)out, (f32x4*)fp->pool,
                         (unsigned*)fp->code);
      ST_LAUNCH_RET();
    }
    if (fp && fp->ugin) {
      const int C4 = Cout / 4, rows = (fp->uH >> 1) + (fp->uH & 1), row_elems = ((fp->uW >> 1) + (fp->uW & 1)) * C4;
      hipLaunchKernelGGL(conv_splitk_finish_unpool_kernel, dim3((unsigned)min(64, cdiv(row_elems, 256)), (unsigned)rows), dim3(256), 0,
                         s, (const f32x4*)workspace, nsplit, total4, fp->uH, fp->uW, C4, (const unsigned*)fp->ucode,
                         (f32x4*)fp->ugin, accumulate);
      ST_LAUNCH_RET();
    }
    hipLaunchKernelGGL(conv_splitk_finish_kernel, dim3((unsigned)min((size_t)2048, (total4 + 255) / 256)), dim3(256), 0, s,
                       (const f32x4*)workspace, nsplit, total4, Cout / 4, (const f32x4*)bias, (const f32x4*)mask, relu,
                       (f32x4*)out, accumulate);
    ST_LAUNCH_RET();
  }
  if (fp) return STROTSS_EINVAL;                      // only the split-K form pools in its finish
  if (Cout % 128 == 0 && cdiv(M, 128) * (Cout / 128) >= 512)
    return launch_conv<128, 128>(in, H, W, Cin, wt, bias, Cout, mask, out, relu, s, accumulate);
  if (cdiv(M, 128) * (Cout / 64) >= 512)
    return launch_conv<128, 64>(in, H, W, Cin, wt, bias, Cout, mask, out, relu, s, accumulate);
  return launch_conv<64, 64>(in, H, W, Cin, wt, bias, Cout, mask, out, relu, s, accumulate);
}

// ---------------------------------------------------------------- first layer (Cin = 3)
// (img - mean)/std is applied to in-bounds taps only: Keras zero-pads the PREPROCESSED tensor.
// cout/16 lanes per pixel, 16 output channels each -> a pixel's cout floats are stored contiguously.
__global__ __launch_bounds__(256) void conv3x3_c3_fwd_kernel(const float* __restrict__ img, int H, int W,
                                                             const float* __restrict__ w_kio,
                                                             const float* __restrict__ bias, int cout,
                                                             f32x4 mean_istd0, f32x4 mean_istd1,
                                                             float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float wsm[];   // 27 * cout
  for (int i = threadIdx.x; i < 27 * cout; i += 256) wsm[i] = w_kio[i];
  __syncthreads();
  const int tpp = cout >> 4;
  const int ppb = 256 / tpp;
  const int sub = threadIdx.x % tpp;
  const int p = blockIdx.x * ppb + threadIdx.x / tpp;
  if (p >= H * W) return;
  const int y = p / W, x = p - y * W;
  const float mean[3] = {mean_istd0[0], mean_istd0[1], mean_istd0[2]};
  const float istd[3] = {mean_istd1[0], mean_istd1[1], mean_istd1[2]};
  float v[27];
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
    const bool ok = yy >= 0 && yy < H && xx >= 0 && xx < W;
    const float* q = img + (ok ? ((size_t)yy * W + xx) * 3 : 0);
#pragma unroll
    for (int c = 0; c < 3; ++c) v[tap * 3 + c] = ok ? (q[c] - mean[c]) * istd[c] : 0.f;
  }
  const int co0 = sub * 16;
  f32x4 acc[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) acc[g] = *reinterpret_cast<const f32x4*>(&bias[co0 + 4 * g]);
#pragma unroll
  for (int k = 0; k < 27; ++k) {
    const float a = v[k];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 w4 = *reinterpret_cast<const f32x4*>(&wsm[k * cout + co0 + 4 * g]);
      acc[g] += a * w4;
    }
  }
  float* o = out + (size_t)p * cout + co0;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    f32x4 r = acc[g];
    r[0] = fmaxf(r[0], 0.f); r[1] = fmaxf(r[1], 0.f); r[2] = fmaxf(r[2], 0.f); r[3] = fmaxf(r[3], 0.f);
    *reinterpret_cast<f32x4*>(o + 4 * g) = r;
  }
}

// gimg[p, ci] (+)= (1/std[ci]) * sum_{tap', co} gout[p + tap' offset, co] * w_tic[tap'][ci][co]
__global__ __launch_bounds__(256) void conv3x3_c3_dgrad_kernel(const float* __restrict__ gout, int H, int W,
                                                               int cout, const float* __restrict__ w_tic,
                                                               f32x4 istd, float* __restrict__ gimg,
                                                               int accumulate) {
  extern __shared__ __attribute__((aligned(16))) float wsm[];   // 27 * cout, [tap][ci][co]
  for (int i = threadIdx.x; i < 27 * cout; i += 256) wsm[i] = w_tic[i];
  __syncthreads();
  const int tpp = cout >> 4;
  const int ppb = 256 / tpp;
  const int sub = threadIdx.x % tpp;
  int p = blockIdx.x * ppb + threadIdx.x / tpp;
  const bool live = p < H * W;
  if (!live) p = H * W - 1;            // keep every lane in the shuffles below
  const int y = p / W, x = p - y * W;
  const int co0 = sub * 16;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
    if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
      const float* q = gout + ((size_t)yy * W + xx) * cout + co0;
      const float* w0 = &wsm[(tap * 3 + 0) * cout + co0];
      const float* w1 = &wsm[(tap * 3 + 1) * cout + co0];
      const float* w2 = &wsm[(tap * 3 + 2) * cout + co0];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 gv = *reinterpret_cast<const f32x4*>(q + 4 * g);
        const f32x4 a = *reinterpret_cast<const f32x4*>(w0 + 4 * g);
        const f32x4 b = *reinterpret_cast<const f32x4*>(w1 + 4 * g);
        const f32x4 c = *reinterpret_cast<const f32x4*>(w2 + 4 * g);
        s0 += gv[0] * a[0] + gv[1] * a[1] + gv[2] * a[2] + gv[3] * a[3];
        s1 += gv[0] * b[0] + gv[1] * b[1] + gv[2] * b[2] + gv[3] * b[3];
        s2 += gv[0] * c[0] + gv[1] * c[1] + gv[2] * c[2] + gv[3] * c[3];
      }
    }
  }
  for (int o = 1; o < tpp; o <<= 1) {
    s0 += __shfl_xor(s0, o, 64);
    s1 += __shfl_xor(s1, o, 64);
    s2 += __shfl_xor(s2, o, 64);
  }
  if (live && sub == 0) {
    float* g = gimg + (size_t)p * 3;
    const float a0 = s0 * istd[0], a1 = s1 * istd[1], a2 = s2 * istd[2];
    if (accumulate) { g[0] += a0; g[1] += a1; g[2] += a2; }
    else { g[0] = a0; g[1] = a1; g[2] = a2; }
  }
}

// ---- first layer on the MFMA (cout == 64): one wave = 32 pixels x 64 channels, K = 27 (+1 zero) as 14
// v_mfma_f32_32x32x2_f32 per 32-channel half; B (taps x channels) sits in LDS.
// A workgroup owns 128 consecutive pixels of one image row per trip.  The f32 MFMA does not co-issue with the VALU on
// gfx950, so every address / bounds / preprocess instruction of the A gather is paid in MFMA time: the 3 x 130 pixel
// patch is therefore staged ONCE per trip, preprocessed and zero-padded, in LDS (one bounds check and one preprocess per
// patch element instead of nine), and the 14 A values of a lane are plain ds_reads at compile-time offsets.
// The wave's 32 pixels x 64 channels are 8 KiB of CONSECUTIVE output: through LDS (row stride 68 floats: the two lane
// halves, 4 rows apart, land 16 banks apart) into eight contiguous 1-KiB dwordx4 stores.  HBM-bound on its 256-byte rows.
#define C3F_RS 68       // LDS row stride of the output staging, floats
#define C3F_SEG 128     // pixels per workgroup trip
#define C3F_PW (C3F_SEG + 2)
__global__ __launch_bounds__(256) void conv3x3_c3_fwd_mfma_kernel(const float* __restrict__ img, int H, int W,
                                                                  const float* __restrict__ w_kio,
                                                                  const float* __restrict__ bias,
                                                                  f32x4 mean_, f32x4 istd_, float* __restrict__ out,
                                                                  unsigned char* __restrict__ bits_out) {
  __shared__ float wsm[28 * 64];
  __shared__ float patch[3 * C3F_PW * 3];
  __shared__ __attribute__((aligned(16))) float stage[4 * 32 * C3F_RS];
  for (int i = threadIdx.x; i < 28 * 64; i += 256) wsm[i] = (i < 27 * 64) ? w_kio[i] : 0.f;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
  const int segs_x = (W + C3F_SEG - 1) / C3F_SEG;
  const int trips = H * segs_x;
  const float b0 = bias[l31], b1 = bias[32 + l31];
  float* st = stage + wave * (32 * C3F_RS);
  // The patch of trip t+1 is fetched into registers while trip t computes: its HBM latency hides behind the MFMAs and the
  // output stores instead of standing between two barriers.  Which patch element a thread stages (row, pixel, channel,
  // its mean and 1/std) does not depend on the trip: worked out once, so a trip costs a thread two adds, the bounds
  // test and one multiply-add per element (the f32 MFMA blocks the vector issue: every instruction here is MFMA time).
  constexpr int NPF = (3 * C3F_PW * 3 + 255) / 256;    // patch elements per thread
  int prow[NPF], poff[NPF];
  float pm[NPF], pis[NPF], pv[NPF];
#pragma unroll
  for (int i = 0; i < NPF; ++i) {
    const int e = threadIdx.x + 256 * i;
    const int r = e / (C3F_PW * 3), rem = e - r * (C3F_PW * 3);
    const int px = rem / 3, ch = rem - px * 3;
    prow[i] = e < 3 * C3F_PW * 3 ? r - 1 : -(1 << 20);                  // (past the patch: never inside the image)
    poff[i] = (px - 1) * 3 + ch;                                        // float offset from pixel x0 of row y + r - 1
    pm[i] = ch == 0 ? mean_[0] : (ch == 1 ? mean_[1] : mean_[2]);
    pis[i] = ch == 0 ? istd_[0] : (ch == 1 ? istd_[1] : istd_[2]);
  }
  auto inside = [&](int i, int y, int x0) {
    const int yy = y + prow[i], x3 = x0 * 3 + poff[i];
    return yy >= 0 && yy < H && x3 >= 0 && x3 < W * 3;
  };
  auto fetch = [&](int trip) {                         // loads only: nothing here waits for them
    const int y = trip / segs_x, x0 = (trip - y * segs_x) * C3F_SEG;
#pragma unroll
    for (int i = 0; i < NPF; ++i)
      pv[i] = img[inside(i, y, x0) ? (size_t)(y + prow[i]) * W * 3 + x0 * 3 + poff[i] : 0];
  };
  if ((int)blockIdx.x < trips) fetch(blockIdx.x);
  for (int trip = blockIdx.x; trip < trips; trip += gridDim.x) {
    const int y = trip / segs_x, x0 = (trip - y * segs_x) * C3F_SEG;
    __syncthreads();                                   // the previous trip's A reads are done (first trip: wsm visible)
#pragma unroll
    for (int i = 0; i < NPF; ++i) {
      const int e = threadIdx.x + 256 * i;
      if (e < 3 * C3F_PW * 3) patch[e] = inside(i, y, x0) ? (pv[i] - pm[i]) * pis[i] : 0.f;
    }
    __syncthreads();
    if (trip + (int)gridDim.x < trips) fetch(trip + gridDim.x);
    const int xw = x0 + wave * 32;                     // this wave's 32 pixels
    if (xw >= W) continue;
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = b0; acc1[r] = b1; }
    // this lane's k = 2*s2 + hh -> (tap, ch) = (k / 3, k % 3) -> patch offset (tap / 3) * PW * 3 + (tap % 3) * 3 + ch
    // = k + (k / 9) * (PW * 3 - 9); relative to the lane's pixel: + (wave * 32 + l31) * 3
    const float* pl = patch + (wave * 32 + l31) * 3;
#pragma unroll
    for (int s2 = 0; s2 < 14; ++s2) {
      constexpr int ROWJ = C3F_PW * 3 - 9;
      const int o0 = 2 * s2 + ((2 * s2) / 9) * ROWJ, o1 = 2 * s2 + 1 + ((2 * s2 + 1) / 9) * ROWJ;
      const float av = pl[(hh && s2 != 13) ? o1 : o0];
      const float a = (s2 == 13 && hh) ? 0.f : av;     // k = 27 is the zero tap
      const float w0 = wsm[(2 * s2 + hh) * 64 + l31], w1 = wsm[(2 * s2 + hh) * 64 + 32 + l31];
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, w0, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, w1, acc1, 0, 0, 0);
    }
    if (bits_out) {
      // sign words of the F(4x4) tile grid (include/strotss_hip.h: relu_bits): this image row is byte y % 4 of its tiles'
      // words; registers 4g .. 4g + 3 of a lane are the 4 pixels of tile 2g + hh of the wave's eight, channel l31 (+ 32)
      unsigned char* brow = bits_out + ((size_t)(y >> 2) * ((W + 3) >> 2) + (xw >> 2)) * 256 + (y & 3);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int tl = 2 * g + hh;
        if (xw + 4 * tl < W) {
          unsigned n0 = 0, n1 = 0;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            n0 |= (acc0[4 * g + c] > 0.f ? 1u : 0u) << c;
            n1 |= (acc1[4 * g + c] > 0.f ? 1u : 0u) << c;
          }
          brow[(tl * 64 + l31) * 4] = (unsigned char)n0;
          brow[(tl * 64 + 32 + l31) * 4] = (unsigned char)n1;
        }
      }
    }
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int row = (reg & 3) + 8 * (reg >> 2) + 4 * hh;
      st[row * C3F_RS + l31] = fmaxf(acc0[reg], 0.f);
      st[row * C3F_RS + 32 + l31] = fmaxf(acc1[reg], 0.f);
    }
    // (same wave wrote and reads: no barrier; the compiler orders the LDS accesses with lgkmcnt)
    const int r4 = lane >> 4, c4 = lane & 15;
    float* orow = out + ((size_t)y * W + xw) * 64;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int row = it * 4 + r4;
      const f32x4 v = *reinterpret_cast<const f32x4*>(st + row * C3F_RS + c4 * 4);
      if (xw + row < W) *reinterpret_cast<f32x4*>(orow + (size_t)row * 64 + c4 * 4) = v;
    }
  }
}

// ---- pixel gradient through an LDS halo tile (cout == 64): a block owns 4 x 32 pixels; the 6 x 34 x 64
// patch of gout is staged once (each element is read 9 times from LDS instead of from L1/L2); 4 lanes per
// pixel take 16 channels each, as in the generic kernel.
#define C3D_TH 4
#define C3D_TW 32
#define C3D_PS 68      // padded pixel stride in floats (64 + 4): 16-lane b128 groups hit distinct banks
__global__ __launch_bounds__(512) void conv3x3_c3_dgrad_lds_kernel(const float* __restrict__ gout, int H, int W,
                                                                   const float* __restrict__ w_tic, f32x4 istd,
                                                                   float* __restrict__ gimg, int accumulate) {
  __shared__ __attribute__((aligned(16))) float tile[(C3D_TH + 2) * (C3D_TW + 2) * C3D_PS];
  const int t = threadIdx.x;
  const int tiles_x = (W + C3D_TW - 1) / C3D_TW;
  const int ty0 = (blockIdx.x / tiles_x) * C3D_TH, tx0 = (blockIdx.x % tiles_x) * C3D_TW;
  // stage (TH+2) x (TW+2) pixels x 64 channels, zero outside the image
  constexpr int NPIX = (C3D_TH + 2) * (C3D_TW + 2);
  constexpr int NST = (NPIX * 16 + 511) / 512;     // loads per thread, issued as one batch (clamped, zero-selected)
  f32x4 sv[NST];
  bool sok[NST];
#pragma unroll
  for (int i = 0; i < NST; ++i) {
    const int e = t + 512 * i;
    const int pix = e >> 4, c4 = e & 15;
    const int py = pix / (C3D_TW + 2), px = pix - py * (C3D_TW + 2);
    const int y = ty0 + py - 1, x = tx0 + px - 1;
    sok[i] = e < NPIX * 16 && y >= 0 && y < H && x >= 0 && x < W;
    sv[i] = *reinterpret_cast<const f32x4*>(&gout[sok[i] ? ((size_t)y * W + x) * 64 + c4 * 4 : 0]);
  }
#pragma unroll
  for (int i = 0; i < NST; ++i) {
    const int e = t + 512 * i;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    if (e < NPIX * 16) *reinterpret_cast<f32x4*>(&tile[(e >> 4) * C3D_PS + (e & 15) * 4]) = sok[i] ? sv[i] : z;
  }
  __syncthreads();
  // wave = (pixel half, channel quarter): the quarter is wave-uniform, so its 3 x 16 weights per tap come through
  // the scalar cache as SGPR operands of the FMAs instead of three more LDS reads per staged value (that form was
  // LDS-bandwidth-bound at 134 us for a 1024^2 image); the four quarters of a pixel meet in LDS at the end
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63;
  const int cq = wave & 3, lp = (wave >> 2) * 64 + lane;          // 128 pixels
  const int ly = lp / C3D_TW, lx = lp - ly * C3D_TW;
  const float* wq = w_tic + cq * 16;
  // every accumulator is a PAIR over even / odd channels: (gv[k], gv[k+1]) * (w[k], w[k+1]) are adjacent registers on
  // both sides, so all 432 FMAs of a lane are 216 v_pk_fma_f32 with an SGPR-pair operand and nothing to shuffle
  f32x2 p0 = {0.f, 0.f}, p1 = {0.f, 0.f}, p2 = {0.f, 0.f};
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    const float* q = &tile[((ly + tap / 3) * (C3D_TW + 2) + lx + tap % 3) * C3D_PS + cq * 16];
    const f32x2* w0 = reinterpret_cast<const f32x2*>(wq + (tap * 3 + 0) * 64);
    const f32x2* w1 = reinterpret_cast<const f32x2*>(wq + (tap * 3 + 1) * 64);
    const f32x2* w2 = reinterpret_cast<const f32x2*>(wq + (tap * 3 + 2) * 64);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 gv = *reinterpret_cast<const f32x4*>(q + 4 * g);
      const f32x2 lo = {gv[0], gv[1]}, hi = {gv[2], gv[3]};
      p0 = __builtin_elementwise_fma(lo, w0[2 * g], p0); p0 = __builtin_elementwise_fma(hi, w0[2 * g + 1], p0);
      p1 = __builtin_elementwise_fma(lo, w1[2 * g], p1); p1 = __builtin_elementwise_fma(hi, w1[2 * g + 1], p1);
      p2 = __builtin_elementwise_fma(lo, w2[2 * g], p2); p2 = __builtin_elementwise_fma(hi, w2[2 * g + 1], p2);
    }
  }
  const float s0 = p0[0] + p0[1], s1 = p1[0] + p1[1], s2 = p2[0] + p2[1];
  __shared__ float red[4][C3D_TH * C3D_TW][3];
  red[cq][lp][0] = s0; red[cq][lp][1] = s1; red[cq][lp][2] = s2;
  __syncthreads();
  if (t < C3D_TH * C3D_TW) {
    const int py = t / C3D_TW, px = t - py * C3D_TW;
    const int y = ty0 + py, x = tx0 + px;
    if (y < H && x < W) {
      float* g = gimg + ((size_t)y * W + x) * 3;
      const float a0 = ((red[0][t][0] + red[1][t][0]) + (red[2][t][0] + red[3][t][0])) * istd[0];
      const float a1 = ((red[0][t][1] + red[1][t][1]) + (red[2][t][1] + red[3][t][1])) * istd[1];
      const float a2 = ((red[0][t][2] + red[1][t][2]) + (red[2][t][2] + red[3][t][2])) * istd[2];
      if (accumulate) { g[0] += a0; g[1] += a1; g[2] += a2; }
      else { g[0] = a0; g[1] = a1; g[2] = a2; }
    }
  }
}

// ---------------------------------------------------------------- 2x2/2 VALID max-pool
// code (optional, one byte per pooled element): index 0..3 of the window's FIRST max in scan order (0,0),(0,1),(1,0),(1,1),
// or 4 when that max is not positive (no gradient) -- all the backward pass needs instead of re-reading the activations
__global__ __launch_bounds__(256) void maxpool2_fwd_kernel(const float* __restrict__ in, int H, int W, int C4,
                                                           float* __restrict__ out, unsigned* __restrict__ code) {
  const int Ho = H >> 1, Wo = W >> 1;
  const int row_elems = Wo * C4;
  const f32x4* src = reinterpret_cast<const f32x4*>(in);
  f32x4* dst = reinterpret_cast<f32x4*>(out);
  // one workgroup row per pooled row, 32-bit index arithmetic (no 64-bit division per element)
  for (int oy = blockIdx.y; oy < Ho; oy += gridDim.y)
  for (int er = blockIdx.x * 256 + threadIdx.x; er < row_elems; er += gridDim.x * 256) {
    const int ox = er / C4, c = er - ox * C4;
    const size_t e = (size_t)oy * row_elems + er;
    const size_t b = ((size_t)(2 * oy) * W + 2 * ox) * C4 + c;
    const f32x4 v00 = src[b], v01 = src[b + C4], v10 = src[b + (size_t)W * C4], v11 = src[b + (size_t)W * C4 + C4];
    f32x4 m;
#pragma unroll
    for (int k = 0; k < 4; ++k) m[k] = fmaxf(fmaxf(v00[k], v01[k]), fmaxf(v10[k], v11[k]));
    dst[e] = m;
    if (code) {
      unsigned packed = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float v[4] = {v00[k], v01[k], v10[k], v11[k]};
        int best = 0;
        float bv = v[0];
#pragma unroll
        for (int q = 1; q < 4; ++q)
          if (v[q] > bv) { bv = v[q]; best = q; }
        packed |= (unsigned)(bv > 0.f ? best : 4) << (8 * k);
      }
      code[e] = packed;
    }
  }
}
// backward from the forward pass's argmax codes: reads gout + 1 byte per pooled element instead of the activations
// One thread per POOLED element (4 channels): code word and gradient are read once and the 2 x 2 window is written from
// registers; one workgroup row per pooled row, 32-bit index arithmetic (the element-per-thread form spent its time in
// four 64-bit divisions per element and fetched every pooled value four times).  An odd last row / column of the input
// lies outside every window: it gets zeros (blockIdx.y == Ho, ox == Wo).
__global__ __launch_bounds__(256) void maxpool2_bwd_code_kernel(const unsigned* __restrict__ code, int H, int W, int C4,
                                                                const float* __restrict__ gout,
                                                                float* __restrict__ gin, int accumulate) {
  const int Ho = H >> 1, Wo = W >> 1;
  const int Wx = Wo + (W & 1);                                  // pooled columns + the odd column, if any
  const int row_elems = Wx * C4;
  const f32x4* g = reinterpret_cast<const f32x4*>(gout);
  f32x4* dst = reinterpret_cast<f32x4*>(gin);
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  for (int oy = blockIdx.y; oy < Ho + (H & 1); oy += gridDim.y) {
    const bool pooled_row = oy < Ho;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < row_elems; e += gridDim.x * 256) {
      const int ox = e / C4, c = e - ox * C4;
      const bool pooled = pooled_row && ox < Wo;
      f32x4 r[4] = {z, z, z, z};
      if (pooled) {
        const size_t po = ((size_t)oy * Wo + ox) * C4 + c;
        const unsigned cd = code[po];
        const f32x4 go = g[po];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const unsigned me = (cd >> (8 * k)) & 0xffu;
#pragma unroll
          for (int q = 0; q < 4; ++q) r[q][k] = me == (unsigned)q ? go[k] : 0.f;
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {                              // q = (dy << 1) | dx
        const int y = 2 * oy + (q >> 1), x = 2 * ox + (q & 1);
        if (y < H && x < W) {
          const size_t o = ((size_t)y * W + x) * C4 + c;
          dst[o] = accumulate ? dst[o] + r[q] : r[q];
        }
      }
    }
  }
}
// gin[y,x,c] = (this pixel is the FIRST max of its window, scan order (0,0),(0,1),(1,0),(1,1))
//              ? gout[y/2,x/2,c] : 0, times (act > 0); pixels outside the pooled area get 0.
__global__ __launch_bounds__(256) void maxpool2_bwd_kernel(const float* __restrict__ act, int H, int W, int C4,
                                                           const float* __restrict__ gout,
                                                           float* __restrict__ gin, int accumulate) {
  const int Ho = H >> 1, Wo = W >> 1;
  const size_t total = (size_t)H * W * C4;
  const f32x4* a = reinterpret_cast<const f32x4*>(act);
  const f32x4* g = reinterpret_cast<const f32x4*>(gout);
  f32x4* dst = reinterpret_cast<f32x4*>(gin);
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const int c = (int)(e % C4);
    const size_t pix = e / C4;
    const int x = (int)(pix % W), y = (int)(pix / W);
    const int oy = y >> 1, ox = x >> 1;
    f32x4 r = {0.f, 0.f, 0.f, 0.f};
    if (oy < Ho && ox < Wo) {
      const size_t b = ((size_t)(2 * oy) * W + 2 * ox) * C4 + c;
      const f32x4 v[4] = {a[b], a[b + C4], a[b + (size_t)W * C4], a[b + (size_t)W * C4 + C4]};
      const int me = ((y & 1) << 1) | (x & 1);
      const f32x4 go = g[((size_t)oy * Wo + ox) * C4 + c];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        int best = 0;
        float bv = v[0][k];
#pragma unroll
        for (int q = 1; q < 4; ++q)
          if (v[q][k] > bv) { bv = v[q][k]; best = q; }
        r[k] = (best == me && v[me][k] > 0.f) ? go[k] : 0.f;
      }
    }
    dst[e] = accumulate ? dst[e] + r : r;
  }
}

// sign words (include/strotss_hip.h: relu_bits) from a finished activation tensor: thread = (tile, channel)
__global__ __launch_bounds__(256) void relu_bits_kernel(const float* __restrict__ act, int H, int W, int C, size_t total,
                                                       unsigned* __restrict__ bits) {
  const int TW = (W + 3) >> 2;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const int c = (int)(e % C);
    const size_t tile = e / C;
    const int ty = (int)(tile / TW), tx = (int)(tile - (size_t)ty * TW);
    unsigned wv = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int y = 4 * ty + r, x = 4 * tx + q;
        if (y < H && x < W && act[((size_t)y * W + x) * C + c] > 0.f) wv |= 1u << (8 * r + q);
      }
    bits[e] = wv;
  }
}

}  // namespace

extern "C" {

int strotss_relu_bits(const float* act, int h, int w, int c, unsigned int* relu_bits, void* stream) {
  ST_CHECK_ARG(act && relu_bits && h > 0 && w > 0 && c > 0, STROTSS_EINVAL);
  const size_t total = (size_t)((h + 3) / 4) * ((w + 3) / 4) * c;
  hipLaunchKernelGGL(relu_bits_kernel, dim3((unsigned)min((size_t)8192, (total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     act, h, w, c, total, relu_bits);
  ST_LAUNCH_RET();
}

int strotss_conv3x3_c3_fwd(const float* img, int h, int w, const float* w_kio, const float* bias, int cout,
                           const float* mean3, const float* std3, float* out, unsigned int* relu_bits_out, void* stream) {
  ST_CHECK_ARG(img && w_kio && bias && mean3 && std3 && out && h > 0 && w > 0, STROTSS_EINVAL);
  ST_CHECK_ARG(cout >= 16 && cout <= 256 && (cout & (cout - 1)) == 0, STROTSS_EALIGN);
  const f32x4 m = {mean3[0], mean3[1], mean3[2], 0.f};
  const f32x4 is = {1.0f / std3[0], 1.0f / std3[1], 1.0f / std3[2], 0.f};
  if (cout == 64 && conv_variant() != 1) {
    const int trips = h * cdiv(w, C3F_SEG);
    hipLaunchKernelGGL(conv3x3_c3_fwd_mfma_kernel, dim3(min(2048, trips)), dim3(256), 0, (hipStream_t)stream,
                       img, h, w, w_kio, bias, m, is, out, (unsigned char*)relu_bits_out);
    ST_LAUNCH_RET();
  }
  const int ppb = 256 / (cout >> 4);
  hipLaunchKernelGGL(conv3x3_c3_fwd_kernel, dim3(cdiv((int64_t)h * w, ppb)), dim3(256),
                     27 * cout * sizeof(float), (hipStream_t)stream, img, h, w, w_kio, bias, cout, m, is, out);
  if (relu_bits_out) return strotss_relu_bits(out, h, w, cout, relu_bits_out, stream);
  ST_LAUNCH_RET();
}

size_t strotss_conv3x3_workspace_bytes(int h, int w, int cin, int cout) {
  if (h <= 0 || w <= 0 || cin <= 0 || cin % 32 || cout <= 0 || cout % 64) return 0;
  return (size_t)conv_splits(h, w, cin, cout) * h * w * cout * sizeof(float);
}

int strotss_conv3x3_relu_fwd(const float* in, int h, int w, int cin, const float* w_tok, const float* bias,
                             int cout, float* out, void* workspace, size_t workspace_bytes, void* stream) {
  ST_CHECK_ARG(in && w_tok && bias && out && h > 0 && w > 0, STROTSS_EINVAL);
  ST_CHECK_ARG(cin > 0 && cin % 32 == 0 && cout > 0 && cout % 64 == 0, STROTSS_EALIGN);
  return conv_dispatch(in, h, w, cin, w_tok, bias, cout, nullptr, out, 1, (hipStream_t)stream, workspace, workspace_bytes);
}

int strotss_conv3x3_dgrad(const float* gout, int h, int w, int cout, const float* w_tik, int cin,
                          const float* act_in, float* gin, int accumulate, void* workspace, size_t workspace_bytes,
                          void* stream) {
  ST_CHECK_ARG(gout && w_tik && gin && h > 0 && w > 0, STROTSS_EINVAL);
  ST_CHECK_ARG(cout > 0 && cout % 32 == 0 && cin > 0 && cin % 64 == 0, STROTSS_EALIGN);
  // the transposed convolution is a convolution with K = cout and N = cin
  return conv_dispatch(gout, h, w, cout, w_tik, nullptr, cin, act_in, gin, 0, (hipStream_t)stream, workspace, workspace_bytes,
                       accumulate);
}

int strotss_conv3x3_relu_pool_fwd(const float* in, int h, int w, int cin, const float* w_tok, const float* bias, int cout,
                                  float* out, float* pool_out, unsigned char* pool_code, void* workspace, size_t workspace_bytes,
                                  void* stream) {
  ST_CHECK_ARG(in && w_tok && bias && out && pool_out && workspace && h > 1 && w > 1, STROTSS_EINVAL);
  ST_CHECK_ARG(cin > 0 && cin % 32 == 0 && cout > 0 && cout % 64 == 0, STROTSS_EALIGN);
  FinishPool fp;
  fp.pool = pool_out; fp.code = pool_code;
  return conv_dispatch(in, h, w, cin, w_tok, bias, cout, nullptr, out, 1, (hipStream_t)stream, workspace, workspace_bytes, 0, &fp);
}

int strotss_conv3x3_dgrad_unpool(const float* gout, int h, int w, int cout, const float* w_tik, int cin,
                                 const unsigned char* pool_code, float* gin_full, int full_h, int full_w, int accumulate,
                                 void* workspace, size_t workspace_bytes, void* stream) {
  ST_CHECK_ARG(gout && w_tik && pool_code && gin_full && workspace && h > 0 && w > 0, STROTSS_EINVAL);
  ST_CHECK_ARG((full_h >> 1) == h && (full_w >> 1) == w, STROTSS_EINVAL);
  ST_CHECK_ARG(cout > 0 && cout % 32 == 0 && cin > 0 && cin % 64 == 0, STROTSS_EALIGN);
  FinishPool fp;
  fp.ucode = pool_code; fp.ugin = gin_full; fp.uH = full_h; fp.uW = full_w;
  return conv_dispatch(gout, h, w, cout, w_tik, nullptr, cin, nullptr, nullptr, 0, (hipStream_t)stream, workspace, workspace_bytes,
                       accumulate, &fp);
}

int strotss_conv3x3_c3_dgrad(const float* gout, int h, int w, int cout, const float* w_tic,
                             const float* std3, float* gimg, int accumulate, void* stream) {
  ST_CHECK_ARG(gout && w_tic && std3 && gimg && h > 0 && w > 0, STROTSS_EINVAL);
  ST_CHECK_ARG(cout >= 16 && cout <= 256 && (cout & (cout - 1)) == 0, STROTSS_EALIGN);
  const f32x4 is = {1.0f / std3[0], 1.0f / std3[1], 1.0f / std3[2], 0.f};
  if (cout == 64 && conv_variant() != 1) {
    const int nblk = cdiv(h, C3D_TH) * cdiv(w, C3D_TW);
    hipLaunchKernelGGL(conv3x3_c3_dgrad_lds_kernel, dim3(nblk), dim3(512), 0, (hipStream_t)stream, gout, h, w, w_tic,
                       is, gimg, accumulate);
    ST_LAUNCH_RET();
  }
  const int ppb = 256 / (cout >> 4);
  hipLaunchKernelGGL(conv3x3_c3_dgrad_kernel, dim3(cdiv((int64_t)h * w, ppb)), dim3(256),
                     27 * cout * sizeof(float), (hipStream_t)stream, gout, h, w, cout, w_tic, is, gimg,
                     accumulate);
  ST_LAUNCH_RET();
}

}  // extern "C"

int st_maxpool2_fwd(const float* in, int h, int w, int c, float* out, unsigned char* code, hipStream_t st) {
  return strotss_maxpool2_fwd(in, h, w, c, out, code, (void*)st);
}

extern "C" {

int strotss_maxpool2_fwd(const float* in, int h, int w, int c, float* out, unsigned char* code, void* stream) {
  ST_CHECK_ARG(in && out && h >= 2 && w >= 2 && c > 0, STROTSS_EINVAL);
  ST_CHECK_ARG(c % 4 == 0, STROTSS_EALIGN);
  const dim3 grid((unsigned)min(64, cdiv((w / 2) * (c / 4), 256)), (unsigned)min(h / 2, 16384));
  hipLaunchKernelGGL(maxpool2_fwd_kernel, grid, dim3(256), 0,
                     (hipStream_t)stream, in, h, w, c / 4, out, reinterpret_cast<unsigned*>(code));
  ST_LAUNCH_RET();
}

int strotss_maxpool2_bwd(const float* act, int h, int w, int c, const float* gout, float* gin,
                         const unsigned char* code, int accumulate, void* stream) {
  ST_CHECK_ARG((act || code) && gout && gin && h >= 2 && w >= 2 && c > 0, STROTSS_EINVAL);
  ST_CHECK_ARG(c % 4 == 0, STROTSS_EALIGN);
  const size_t total = (size_t)h * w * (c / 4);
  const dim3 grid((unsigned)min((size_t)8192, (total + 255) / 256));
  if (code) {
    const int row_elems = ((w >> 1) + (w & 1)) * (c / 4), rows = (h >> 1) + (h & 1);
    const dim3 g2((unsigned)min(64, cdiv(row_elems, 256)), (unsigned)min(rows, 16384));
    hipLaunchKernelGGL(maxpool2_bwd_code_kernel, g2, dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const unsigned*>(code), h, w, c / 4, gout, gin, accumulate);
  }
  else
    hipLaunchKernelGGL(maxpool2_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, act, h, w, c / 4, gout, gin, accumulate);
  ST_LAUNCH_RET();
}

}  // extern "C"
