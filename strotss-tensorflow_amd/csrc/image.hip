// HBM-bound image-side kernels of the STROTSS step: TF2 bilinear resize and its adjoint
// (Laplacian fold, strotss_utils.py:139-163), hypercolumn gather / scatter-add
// (Sampling._sample, strotss_utils.py:25-81), multi-tensor RMSprop (run_strotss.py:63,148),
// postprocess (strotss_utils.py:170-175).
#include "internal.h"

namespace {

// TF2 `tf.image.resize(bilinear)` source coordinate, computed in float32 exactly as TF's
// HalfPixelScaler does: (float(i) + 0.5f) * scale - 0.5f   (no FMA contraction).
struct AxisTap { int lo, hi; float lerp; };
__device__ __forceinline__ AxisTap axis_tap(int i, float scale, int in_size) {
  const float src = __fsub_rn(__fmul_rn(__fadd_rn((float)i, 0.5f), scale), 0.5f);
  const float fl = floorf(src);
  AxisTap t;
  t.lo = max((int)fl, 0);
  t.hi = min((int)ceilf(src), in_size - 1);
  t.lerp = __fsub_rn(src, fl);
  return t;
}

// One thread per OUTPUT PIXEL (all C channels), one workgroup row per output row: no 64-bit div/mod per element, the
// two axis taps are computed once per pixel, the row bases once per row.  C = 0: channel count at run time.
template <int C>
__global__ __launch_bounds__(256) void resize_bilinear_kernel(const float* __restrict__ in, int ih, int iw,
                                                              int c_rt, float* __restrict__ out, int oh, int ow,
                                                              float sy, float sx, float alpha,
                                                              const float* __restrict__ add) {
  const int c = C ? C : c_rt;
  for (int oy = blockIdx.y; oy < oh; oy += gridDim.y) {
    const AxisTap ty = axis_tap(oy, sy, ih);
    const float* r0 = in + (size_t)ty.lo * iw * c;
    const float* r1 = in + (size_t)ty.hi * iw * c;
    const size_t orow = (size_t)oy * ow * c;
    for (int ox = blockIdx.x * 256 + threadIdx.x; ox < ow; ox += gridDim.x * 256) {
      const AxisTap tx = axis_tap(ox, sx, iw);
      const int a = tx.lo * c, b = tx.hi * c;
      const size_t o = orow + (size_t)ox * c;
#pragma unroll
      for (int ch = 0; ch < (C ? C : 1); ++ch)
        for (int cc = ch; cc < c; cc += (C ? C : 1)) {
          const float tl = r0[a + cc], tr = r0[b + cc], bl = r1[a + cc], br = r1[b + cc];
          const float top = tl + (tr - tl) * tx.lerp;
          const float bot = bl + (br - bl) * tx.lerp;
          float v = alpha * (top + (bot - top) * ty.lerp);        // fold_pyramid_kernel: the same expression with alpha = 1
          if (add) v += add[o + cc];
          out[o + cc] = v;
        }
    }
  }
}

// ---------------------------------------------------------------- the whole Laplacian fold in ONE launch
// fold_laplacian_pyramid (strotss_utils.py:159-163): ret = x[L-1]; for k = L-2 .. 0: ret = x[k] + up(ret).  As L-1 dependent
// resize launches the fold costs ~5 us per level whatever the image size (5 of the 97 launches of a 64-px step); here one
// workgroup owns a 32 x 32 tile of the image and RECOMPUTES the footprints of its tile in the coarser levels, bottom-up,
// through two LDS buffers: the footprint of a region in the next level is [tap(lo).lo, tap(hi).hi] per axis, at most
// 18 x 18, 11 x 11, 8 x 8 ... pixels for the halving pyramids make_laplacian_pyramid builds.  Same taps, same arithmetic
// and same operation order per pixel as resize_bilinear_kernel<3>.
#define FOLD_TILE 32
#define FOLD_REGION 24                       // footprint side limit of levels >= 1 (checked on the host)
struct FoldLevels { int n; int h[STROTSS_MAX_TENSORS], w[STROTSS_MAX_TENSORS]; const float* v[STROTSS_MAX_TENSORS]; };
__device__ __forceinline__ float bilerp3(float tl, float tr, float bl, float br, float lx, float ly) {
  const float top = tl + (tr - tl) * lx;
  const float bot = bl + (br - bl) * lx;
  return 1.0f * (top + (bot - top) * ly);     // alpha = 1 of resize_bilinear_kernel
}
__global__ __launch_bounds__(256) void fold_pyramid_kernel(FoldLevels p, float* __restrict__ img) {
  __shared__ float buf[2][FOLD_REGION * FOLD_REGION * 3];
  __shared__ int reg[STROTSS_MAX_TENSORS][4];                       // y0, y1, x0, x1 (inclusive) of every level's footprint
  const int t = threadIdx.x;
  if (t == 0) {
    int y0 = blockIdx.y * FOLD_TILE, y1 = min(y0 + FOLD_TILE, p.h[0]) - 1;
    int x0 = blockIdx.x * FOLD_TILE, x1 = min(x0 + FOLD_TILE, p.w[0]) - 1;
    reg[0][0] = y0; reg[0][1] = y1; reg[0][2] = x0; reg[0][3] = x1;
    for (int k = 1; k < p.n; ++k) {
      const float sy = (float)p.h[k] / (float)p.h[k - 1], sx = (float)p.w[k] / (float)p.w[k - 1];
      const int ny0 = axis_tap(y0, sy, p.h[k]).lo, ny1 = axis_tap(y1, sy, p.h[k]).hi;
      const int nx0 = axis_tap(x0, sx, p.w[k]).lo, nx1 = axis_tap(x1, sx, p.w[k]).hi;
      y0 = ny0; y1 = ny1; x0 = nx0; x1 = nx1;
      reg[k][0] = y0; reg[k][1] = y1; reg[k][2] = x0; reg[k][3] = x1;
    }
  }
  __syncthreads();
  // coarsest level: a copy of its footprint
  {
    const int k = p.n - 1, y0 = reg[k][0], x0 = reg[k][2], rh = reg[k][1] - y0 + 1, rw = reg[k][3] - x0 + 1;
    float* dst = buf[k & 1];
    for (int e = t; e < rh * rw; e += 256) {
      const int ry = e / rw, rx = e - ry * rw;
      const float* s = p.v[k] + ((size_t)(y0 + ry) * p.w[k] + x0 + rx) * 3;
      dst[e * 3 + 0] = s[0]; dst[e * 3 + 1] = s[1]; dst[e * 3 + 2] = s[2];
    }
  }
  __syncthreads();
  for (int k = p.n - 2; k >= 0; --k) {
    const int y0 = reg[k][0], x0 = reg[k][2], rh = reg[k][1] - y0 + 1, rw = reg[k][3] - x0 + 1;
    const int cy0 = reg[k + 1][0], cx0 = reg[k + 1][2], cw = reg[k + 1][3] - cx0 + 1;
    const int ih = p.h[k + 1], iw = p.w[k + 1];
    const float sy = (float)ih / (float)p.h[k], sx = (float)iw / (float)p.w[k];
    const float* src = buf[(k + 1) & 1];
    float* dst = k == 0 ? nullptr : buf[k & 1];
    for (int e = t; e < rh * rw; e += 256) {
      const int ry = e / rw, rx = e - ry * rw;
      const int oy = y0 + ry, ox = x0 + rx;
      const AxisTap ty = axis_tap(oy, sy, ih), tx = axis_tap(ox, sx, iw);
      const float* r0 = src + (size_t)(ty.lo - cy0) * cw * 3;
      const float* r1 = src + (size_t)(ty.hi - cy0) * cw * 3;
      const int a = (tx.lo - cx0) * 3, b = (tx.hi - cx0) * 3;
      const size_t o = ((size_t)oy * p.w[k] + ox) * 3;
      const float* add = p.v[k] + o;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        float v = bilerp3(r0[a + c], r0[b + c], r1[a + c], r1[b + c], tx.lerp, ty.lerp);
        v += add[c];
        if (k == 0) img[o + c] = v; else dst[e * 3 + c] = v;
      }
    }
    __syncthreads();
  }
}

// Adjoint as a gather: gin[iy,ix] = sum over the output pixels whose taps touch (iy,ix).  One thread per INPUT pixel
// (all C channels); the candidate output rows / columns and their weights are found once per pixel (at most ADJ_MAX
// per axis in registers: 2x upsampling has <= 6; wider ranges take the generic loop).  Summation order: output rows
// ascending, columns ascending inside a row -- fixed, so the result is bitwise reproducible.
#define ADJ_MAX 8
__device__ __forceinline__ float adj_weight(int o, float scale, int in_size, int i) {
  const AxisTap t = axis_tap(o, scale, in_size);
  return (t.lo == i ? 1.0f - t.lerp : 0.f) + (t.hi == i ? t.lerp : 0.f);
}
// one input pixel (iy, ix) of the adjoint, 3 channels, from any source of output rows: the summation order of
// resize_adjoint_kernel (output rows ascending, columns ascending inside a row, zero weights skipped)
struct AdjGlobal3 {
  const float* p; int ow;
  __device__ __forceinline__ const float* at(int oy, int ox) const { return p + ((size_t)oy * ow + ox) * 3; }
};
struct AdjRegion3 {                          // a rectangle of the output map staged in LDS
  const float* p; int y0, x0, rw;
  __device__ __forceinline__ const float* at(int oy, int ox) const { return p + ((oy - y0) * rw + (ox - x0)) * 3; }
};
__device__ __forceinline__ void adj_range(int i, float inv_scale, int out_size, int& o0, int& o1) {
  // candidates: src(o) in (i-1, i+1)  <=>  o in ((i-0.5)/s - 0.5, (i+1.5)/s - 0.5); +-1 safety margin
  o0 = max(0, (int)floorf(((float)i - 0.5f) * inv_scale - 0.5f) - 1);
  o1 = min(out_size - 1, (int)ceilf(((float)i + 1.5f) * inv_scale - 0.5f) + 1);
}
template <class Src>
__device__ __forceinline__ void adjoint_pixel3(const Src& g, int oh, int ow, int ih, int iw, float sy, float sx, int iy,
                                               int ix, float (&acc)[3]) {
  int oy0, oy1, ox0, ox1;
  adj_range(iy, 1.0f / sy, oh, oy0, oy1);
  adj_range(ix, 1.0f / sx, ow, ox0, ox1);
  acc[0] = acc[1] = acc[2] = 0.f;
  if (ox1 - ox0 < ADJ_MAX) {                 // the column weights once per pixel (2x upsampling has <= 6 candidates)
    float wx[ADJ_MAX];
#pragma unroll
    for (int q = 0; q < ADJ_MAX; ++q) wx[q] = (ox0 + q <= ox1) ? adj_weight(ox0 + q, sx, iw, ix) : 0.f;
    for (int oy = oy0; oy <= oy1; ++oy) {
      const float wy = adj_weight(oy, sy, ih, iy);
      if (wy == 0.f) continue;
      float row[3] = {0.f, 0.f, 0.f};
#pragma unroll
      for (int q = 0; q < ADJ_MAX; ++q)
        if (wx[q] != 0.f) {
          const float* p = g.at(oy, ox0 + q);
#pragma unroll
          for (int ch = 0; ch < 3; ++ch) row[ch] += wx[q] * p[ch];
        }
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) acc[ch] += wy * row[ch];
    }
    return;
  }
  for (int oy = oy0; oy <= oy1; ++oy) {
    const float wy = adj_weight(oy, sy, ih, iy);
    if (wy == 0.f) continue;
    float row[3] = {0.f, 0.f, 0.f};
    for (int ox = ox0; ox <= ox1; ++ox) {
      const float w = adj_weight(ox, sx, iw, ix);
      if (w != 0.f) {
        const float* q = g.at(oy, ox);
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) row[ch] += w * q[ch];
      }
    }
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) acc[ch] += wy * row[ch];
  }
}

// TWO levels of the fold's adjoint in one launch (3 channels): g1 = up^T(g0), g2 = up^T(g1).  A workgroup owns an 8 x 8
// tile of g2 and the 16 x 16 tile of g1 above it; it computes the g1 pixels of its tile plus the few rows and columns
// around it that its g2 tile reads (recomputed, not exchanged), keeps them in LDS, stores its own, then gathers g2 from
// LDS.  Every pixel runs adjoint_pixel3 -- the arithmetic of the level-by-level launches, bit for bit.
#define ADJ2_T2 8
#define ADJ2_REGION 32
__global__ __launch_bounds__(256) void resize_adjoint_pair_kernel(const float* __restrict__ g0, int h0, int w0,
                                                                  float* __restrict__ g1, int h1, int w1,
                                                                  float* __restrict__ g2, int h2, int w2) {
  __shared__ float mid[ADJ2_REGION * ADJ2_REGION * 3];
  const int t = threadIdx.x;
  const float sy1 = (float)h1 / (float)h0, sx1 = (float)w1 / (float)w0;      // level 1 from level 0
  const float sy2 = (float)h2 / (float)h1, sx2 = (float)w2 / (float)w1;      // level 2 from level 1
  // own tiles (either may be empty at the bottom / right edge)
  const int ty0 = blockIdx.y * ADJ2_T2, ty1 = min(ty0 + ADJ2_T2, h2) - 1, tx0 = blockIdx.x * ADJ2_T2, tx1 = min(tx0 + ADJ2_T2, w2) - 1;
  const int oy0 = blockIdx.y * 2 * ADJ2_T2, oy1 = min(oy0 + 2 * ADJ2_T2, h1) - 1;
  const int ox0 = blockIdx.x * 2 * ADJ2_T2, ox1 = min(ox0 + 2 * ADJ2_T2, w1) - 1;
  const bool has2 = ty0 <= ty1 && tx0 <= tx1, own1 = oy0 <= oy1 && ox0 <= ox1;
  // region of level 1 to compute: own tile + what the level-2 tile gathers from
  int ry0 = oy0, ry1 = oy1, rx0 = ox0, rx1 = ox1;
  if (has2) {
    int a, b, c, d;
    adj_range(ty0, 1.0f / sy2, h1, a, b); adj_range(ty1, 1.0f / sy2, h1, c, d);
    const int ny0 = a, ny1 = d;
    adj_range(tx0, 1.0f / sx2, w1, a, b); adj_range(tx1, 1.0f / sx2, w1, c, d);
    const int nx0 = a, nx1 = d;
    if (own1) { ry0 = min(ry0, ny0); ry1 = max(ry1, ny1); rx0 = min(rx0, nx0); rx1 = max(rx1, nx1); }
    else { ry0 = ny0; ry1 = ny1; rx0 = nx0; rx1 = nx1; }
  } else if (!own1) {
    return;
  }
  const int rh = ry1 - ry0 + 1, rw = rx1 - rx0 + 1;            // <= ADJ2_REGION (checked on the host)
  const AdjGlobal3 src0{g0, w0};
  for (int e = t; e < rh * rw; e += 256) {
    const int ry = e / rw, rx = e - ry * rw;
    const int iy = ry0 + ry, ix = rx0 + rx;
    float acc[3];
    adjoint_pixel3(src0, h0, w0, h1, w1, sy1, sx1, iy, ix, acc);
    mid[e * 3 + 0] = acc[0]; mid[e * 3 + 1] = acc[1]; mid[e * 3 + 2] = acc[2];
    if (own1 && iy >= oy0 && iy <= oy1 && ix >= ox0 && ix <= ox1) {
      float* o = g1 + ((size_t)iy * w1 + ix) * 3;
      o[0] = acc[0]; o[1] = acc[1]; o[2] = acc[2];
    }
  }
  __syncthreads();
  if (!has2) return;
  const AdjRegion3 src1{mid, ry0, rx0, rw};
  const int th = ty1 - ty0 + 1, tw = tx1 - tx0 + 1;
  for (int e = t; e < th * tw; e += 256) {
    const int iy = ty0 + e / tw, ix = tx0 + e % tw;
    float acc[3];
    adjoint_pixel3(src1, h1, w1, h2, w2, sy2, sx2, iy, ix, acc);
    float* o = g2 + ((size_t)iy * w2 + ix) * 3;
    o[0] = acc[0]; o[1] = acc[1]; o[2] = acc[2];
  }
}

template <int C>
__global__ __launch_bounds__(256) void resize_adjoint_kernel(const float* __restrict__ gout, int oh, int ow,
                                                             int c_rt, float* __restrict__ gin, int ih, int iw,
                                                             float sy, float sx) {
  const int c = C ? C : c_rt;
  const float isy = 1.0f / sy, isx = 1.0f / sx;
  for (int iy = blockIdx.y; iy < ih; iy += gridDim.y) {
    // candidates: src(o) in (i-1, i+1)  <=>  o in ((i-0.5)/s - 0.5, (i+1.5)/s - 0.5); +-1 safety margin
    const int oy0 = max(0, (int)floorf(((float)iy - 0.5f) * isy - 0.5f) - 1);
    const int oy1 = min(oh - 1, (int)ceilf(((float)iy + 1.5f) * isy - 0.5f) + 1);
    for (int ix = blockIdx.x * 256 + threadIdx.x; ix < iw; ix += gridDim.x * 256) {
      const int ox0 = max(0, (int)floorf(((float)ix - 0.5f) * isx - 0.5f) - 1);
      const int ox1 = min(ow - 1, (int)ceilf(((float)ix + 1.5f) * isx - 0.5f) + 1);
      float acc[C ? C : 4];
#pragma unroll
      for (int ch = 0; ch < (C ? C : 4); ++ch) acc[ch] = 0.f;
      if (C && ox1 - ox0 < ADJ_MAX) {
        float wx[ADJ_MAX];
#pragma unroll
        for (int q = 0; q < ADJ_MAX; ++q) wx[q] = (ox0 + q <= ox1) ? adj_weight(ox0 + q, sx, iw, ix) : 0.f;
        for (int oy = oy0; oy <= oy1; ++oy) {
          const float wy = adj_weight(oy, sy, ih, iy);
          if (wy == 0.f) continue;
          const float* g = gout + ((size_t)oy * ow + ox0) * c;
          float row[C ? C : 1];
#pragma unroll
          for (int ch = 0; ch < C; ++ch) row[ch] = 0.f;
#pragma unroll
          for (int q = 0; q < ADJ_MAX; ++q)
            if (wx[q] != 0.f) {
#pragma unroll
              for (int ch = 0; ch < C; ++ch) row[ch] += wx[q] * g[q * c + ch];
            }
#pragma unroll
          for (int ch = 0; ch < C; ++ch) acc[ch] += wy * row[ch];
        }
#pragma unroll
        for (int ch = 0; ch < C; ++ch) gin[((size_t)iy * iw + ix) * c + ch] = acc[ch];
      } else {
        for (int ch0 = 0; ch0 < c; ch0 += 4) {            // generic: any channel count, any range
          const int nc = min(4, c - ch0);
          float a4[4] = {0.f, 0.f, 0.f, 0.f};
          for (int oy = oy0; oy <= oy1; ++oy) {
            const float wy = adj_weight(oy, sy, ih, iy);
            if (wy == 0.f) continue;
            float row[4] = {0.f, 0.f, 0.f, 0.f};
            for (int ox = ox0; ox <= ox1; ++ox) {
              const float w = adj_weight(ox, sx, iw, ix);
              if (w != 0.f)
                for (int ch = 0; ch < nc; ++ch) row[ch] += w * gout[((size_t)oy * ow + ox) * c + ch0 + ch];
            }
            for (int ch = 0; ch < nc; ++ch) a4[ch] += wy * row[ch];
          }
          for (int ch = 0; ch < nc; ++ch) gin[((size_t)iy * iw + ix) * c + ch0 + ch] = a4[ch];
        }
      }
    }
  }
}

// ---------------------------------------------------------------- hypercolumns
struct SampleTap { int ia, ib, ic, id; float wa, wb, wc, wd; };   // (wa, wb: row x0; wc, wd: row x1)

// strotss_utils.py:31-37 (cumulative float32 `indices /= y`) + 43-64 (floor / clip / 4 taps)
__device__ __forceinline__ SampleTap sample_tap(const strotss_maps_t& m, int k, float gx, float gy,
                                                int bilinear) {
  for (int q = 0; q < m.n_div[k]; ++q) {
    gx = __fdiv_rn(gx, m.div[q]);
    gy = __fdiv_rn(gy, m.div[q]);
  }
  const int h = m.h[k], w = m.w[k];
  SampleTap t;
  if (bilinear) {
    const float gxf = floorf(gx), gyf = floorf(gy);
    const float dx = gx - gxf, dy = gy - gyf;
    t.wa = (1.f - dx) * (1.f - dy);
    t.wb = (1.f - dx) * dy;
    t.wc = dx * (1.f - dy);
    t.wd = dx * dy;
    const int x0 = (int)fminf(fmaxf(gxf, 0.f), (float)(h - 1));
    const int y0 = (int)fminf(fmaxf(gyf, 0.f), (float)(w - 1));
    int x1 = min(x0 + 1, h - 1);
    const int y1 = min(y0 + 1, w - 1);
    int xa = x0;
    if (m.rows[k] > 0) {                            // window of a sharded map (clamped: never out of the buffer)
      const int ra = x0 - m.row0[k], rb = x1 - m.row0[k];
      xa = min(max(ra, 0), m.rows[k] - 1);
      x1 = min(max(rb, 0), m.rows[k] - 1);
      // adjoint of a halo-exchange strip: rows outside take nothing (integer flags: no lane mask kept across the map loop)
      if (m.window_drop & (int)(ra != xa)) { t.wa = 0.f; t.wb = 0.f; }
      if (m.window_drop & (int)(rb != x1)) { t.wc = 0.f; t.wd = 0.f; }
    }
    t.ia = xa * w + y0; t.ib = xa * w + y1; t.ic = x1 * w + y0; t.id = x1 * w + y1;
  } else {
    int xi = (int)fminf(fmaxf(gx, 0.f), (float)(h - 1));   // clip, then truncating cast
    const int yi = (int)fminf(fmaxf(gy, 0.f), (float)(w - 1));
    if (m.rows[k] > 0) xi = min(max(xi - m.row0[k], 0), m.rows[k] - 1);
    t.ia = t.ib = t.ic = t.id = xi * w + yi;
    t.wa = 1.f; t.wb = t.wc = t.wd = 0.f;
  }
  return t;
}

// one block per sample row.  The row's 64-channel chunks (35 of them for the ten maps of a step) are dealt round-robin to the
// block's four waves: every map is a random pixel of a large tensor (an HBM round trip), and ten of them walked one after the
// other by the whole block made the kernel a chain of ten latencies (18 us for 2048 blocks, all resident at once).  Each wave
// now walks its own quarter of the chunks -- the same loads, weights and arithmetic per element, bit for bit.
__device__ __forceinline__ void gather_row(const strotss_maps_t& m, const float* __restrict__ idx, int s, int bilinear,
                                           float* __restrict__ out, int ld, int dtotal) {
  if (m.sample_range && (s < m.sample_range[0] || s >= m.sample_range[1])) return;      // not this rank's sample
  const float gx = idx[2 * s], gy = idx[2 * s + 1];
  float* o = out + (size_t)s * ld;
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), lane = threadIdx.x & 63;
  int off = 0, chunk = 0;
  for (int k = 0; k < m.n_maps; ++k) {
    const int c = m.c[k], nchunk = (c + 63) >> 6;
    // first chunk of this map that belongs to this wave: chunk ids chunk .. chunk + nchunk - 1, wave takes id % 4 == wave
    int q0 = (wave - chunk) & 3;
    if (q0 < nchunk) {
      const SampleTap t = sample_tap(m, k, gx, gy, bilinear);
      const float* src = m.map[k];
      if (bilinear) {
        const float* pa = src + (size_t)t.ia * c; const float* pb = src + (size_t)t.ib * c;
        const float* pc = src + (size_t)t.ic * c; const float* pd = src + (size_t)t.id * c;
        for (int q = q0; q < nchunk; q += 4) {
          const int ch = q * 64 + lane;
          if (ch < c) o[off + ch] = pa[ch] * t.wa + pb[ch] * t.wb + pc[ch] * t.wc + pd[ch] * t.wd;
        }
      } else {
        const float* pa = src + (size_t)t.ia * c;
        for (int q = q0; q < nchunk; q += 4) {
          const int ch = q * 64 + lane;
          if (ch < c) o[off + ch] = pa[ch];
        }
      }
    }
    off += c;
    chunk += nchunk;
  }
  for (int ch = dtotal + threadIdx.x; ch < ld; ch += 256) o[ch] = 0.f;
}
__global__ __launch_bounds__(256) void hypercol_gather_kernel(strotss_maps_t m, const float* __restrict__ idx,
                                                              int bilinear, float* __restrict__ out, int ld,
                                                              int dtotal) {
  gather_row(m, idx, blockIdx.x, bilinear, out, ld, dtotal);
}
// two gathers at the same sample positions in one launch (blocks [0, n): maps a, [n, 2n): maps b) + an optional zero fill of
// `zero_rows` rows of `zero` (block s of the first half clears row s, the last one also the rows from n on)
__global__ __launch_bounds__(256) void hypercol_gather2_kernel(strotss_maps_t ma, strotss_maps_t mb,
                                                               const float* __restrict__ idx, int n, int bilinear,
                                                               float* __restrict__ out_a, float* __restrict__ out_b, int ld,
                                                               int dtotal_a, int dtotal_b, float* __restrict__ zero,
                                                               int zero_rows) {
  const int b = blockIdx.x;
  if (b < n) {
    gather_row(ma, idx, b, bilinear, out_a, ld, dtotal_a);
    if (zero) {
      const int r1 = b == n - 1 ? zero_rows : min(b + 1, zero_rows);
      for (int r = b; r < r1; ++r)
        for (int ch = threadIdx.x; ch < ld; ch += 256) zero[(size_t)r * ld + ch] = 0.f;
    }
  } else {
    gather_row(mb, idx, b - n, bilinear, out_b, ld, dtotal_b);
  }
}
// Tiny maps (at most SCATTER_DENSE_MAX_PIX pixels: the 4 x 4 and 8 x 8 maps of the 64 / 128-px scales): 4096 taps land on 16-64
// pixels, i.e. 64-256 atomic adds per address, and the memory side serialises them (22 us for the 4 x 4 x 512 map against
// 11 us for the same adds spread over a 16 x 16 one).  There ONE workgroup owns a (pixel, 64-channel chunk): it lists the
// samples with a tap on its pixel (weights of coinciding taps summed), in sample order, its four waves add the listed rows of
// gfeat with four loads in flight, and the four partial sums are combined in a fixed order and added with a plain
// read-modify-write: no atomics, bitwise reproducible for these maps.  b: block index within map k, off: first column of map k.
#define SCATTER_DENSE_MAX_PIX 64
__device__ __forceinline__ void scatter_dense_block(const strotss_maps_t& m, const float* __restrict__ idx, int n,
                                                    const float* __restrict__ gfeat, int ld, bool masked, int k, int off, int b) {
  __shared__ int li[1024];
  __shared__ float lw[1024];
  __shared__ int cnts[4];
  __shared__ float red[4][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = m.c[k], nchunk = (c + 63) >> 6;
  const int p = b / nchunk, q = b - p * nchunk;
  const int s0 = m.sample_range ? m.sample_range[0] : 0, s1 = m.sample_range ? min(n, m.sample_range[1]) : n;
  const int ch = q * 64 + lane;
  const bool in = ch < c;
  const float* g = gfeat + off + (in ? ch : c - 1);
  float acc = 0.f;
  for (int base = s0; base < s1; base += 1024) {
    float w4[4];
    int cnt = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int s = base + 4 * tid + u;
      float w = 0.f;
      if (s < s1) {
        const SampleTap t = sample_tap(m, k, idx[2 * s], idx[2 * s + 1], 1);
        w = (t.ia == p ? t.wa : 0.f) + (t.ib == p ? t.wb : 0.f) + (t.ic == p ? t.wc : 0.f) + (t.id == p ? t.wd : 0.f);
      }
      w4[u] = w;
      cnt += w != 0.f;
    }
    int incl = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int up = __shfl_up(incl, o, 64);
      if (lane >= o) incl += up;
    }
    __syncthreads();                                  // (the previous chunk's list has been read)
    if (lane == 63) cnts[wave] = incl;
    __syncthreads();
    int at = incl - cnt, total = 0;
#pragma unroll
    for (int wv = 0; wv < 4; ++wv) {
      at += wv < wave ? cnts[wv] : 0;
      total += cnts[wv];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (w4[u] != 0.f) { li[at] = base + 4 * tid + u; lw[at] = w4[u]; ++at; }
    __syncthreads();
    int e = wave;
    for (; e + 12 < total; e += 16) {                 // this wave's entries e, e + 4, e + 8, e + 12: four rows in flight
      float v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = g[(size_t)li[e + 4 * u] * ld];
#pragma unroll
      for (int u = 0; u < 4; ++u) acc += lw[e + 4 * u] * v[u];
    }
    for (; e < total; e += 4) acc += lw[e] * g[(size_t)li[e] * ld];
  }
  red[wave][lane] = acc;
  __syncthreads();
  if (wave == 0 && in) {
    const float tot = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
    const size_t o = (size_t)p * c + ch;
    if (tot != 0.f && (!masked || m.map[k][o] > 0.f)) m.gmap[k][o] += tot;
  }
}
__global__ __launch_bounds__(256) void hypercol_scatter_kernel(strotss_maps_t m, const float* __restrict__ idx,
                                                               const float* __restrict__ gfeat, int ld,
                                                               int relu_mask_from, int map_begin,
                                                               int map_end, int n, unsigned dense_mask, int dense_blocks) {
  if ((int)blockIdx.x < dense_blocks) {               // the tiny maps' (pixel, channel chunk) blocks come first
    int b = (int)blockIdx.x, off = 0;
    for (int k = 0; k < map_end; ++k) {
      if ((dense_mask >> k) & 1u) {
        const int nb = (m.rows[k] > 0 ? m.rows[k] : m.h[k]) * m.w[k] * ((m.c[k] + 63) >> 6);
        if (b < nb) { scatter_dense_block(m, idx, n, gfeat, ld, k >= relu_mask_from, k, off, b); return; }
        b -= nb;
      }
      off += m.c[k];
    }
    return;
  }
  const int s = (int)blockIdx.x - dense_blocks;
  if (m.sample_range && (s < m.sample_range[0] || s >= m.sample_range[1])) return;
  const float gx = idx[2 * s], gy = idx[2 * s + 1];
  const float* g = gfeat + (size_t)s * ld;
  // 64-channel chunks dealt round-robin to the four waves, as in the gather: a wave's chain of dependent latencies (tap
  // arithmetic, gradient load, ReLU-mask load, atomic) covers a quarter of the maps instead of all of them
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), lane = threadIdx.x & 63;
  int off = 0, chunk = 0;
  for (int k = 0; k < map_end; ++k) {
    const int c = m.c[k], nchunk = (c + 63) >> 6;
    const int q0 = (wave - chunk) & 3;
    chunk += nchunk;
    if (k < map_begin || q0 >= nchunk || ((dense_mask >> k) & 1u)) { off += c; continue; }
    const SampleTap t = sample_tap(m, k, gx, gy, 1);
    const float* act = m.map[k];
    float* dst = m.gmap[k];
    const bool masked = k >= relu_mask_from;
    const int ii[4] = {t.ia, t.ib, t.ic, t.id};
    const float ww[4] = {t.wa, t.wb, t.wc, t.wd};
    // Wave-uniform trip count and a per-lane `in` test made afresh in every trip: no exec mask or tap predicate is kept in
    // an SGPR pair across the channel loop (every saved mask below lives for the few instructions around one atomic).
    for (int q = q0; q < nchunk; q += 4) {
      const int ch = q * 64 + lane;
      const bool in = ch < c;
      const int chc = in ? ch : c - 1;                                     // clamped: the load below is always in range
      const float gv = in ? g[off + chc] : 0.f;
#pragma unroll
      for (int tq = 0; tq < 4; ++tq) {
        // The skip is decided per lane and per trip from the PRODUCT (a vector compare on a value made in this
        // iteration), never from a wave-uniform `ww[q] == 0` test: the compiler kept that one as four lane masks in
        // s[2:9] across the whole channel loop, and the build that lost whole (wave, tap) contributions next to another
        // process was the one with tap 0's mask in s[2:3].  A zero product adds nothing.
        const float v = ww[tq] * gv;
        const size_t o = (size_t)ii[tq] * c + chc;
        if (v != 0.f && (!masked || act[o] > 0.f)) atomicAdd(&dst[o], v);
      }
    }
    off += c;
  }
}

// ---------------------------------------------------------------- deterministic adjoint (sorted scatter)
// The atomic scatter above adds the taps of different samples into a shared pixel in whatever order the hardware
// serves them: the gradient differs in its last bits from run to run.  The PLAN below orders every map's (sample, tap)
// entries by destination pixel ONCE per index set (bitonic sort of <= 4096 keys in LDS, one workgroup per map); the
// sorted scatter then gives each destination pixel to ONE workgroup, which sums its entries in plan order and adds the
// sum with a plain read-modify-write: no atomics, bitwise reproducible.
#define PLAN_E 4096                                  // entries per map: 4 taps x <= 1024 samples
struct PlanView {
  int* nseg; int* seg_start; int* pix; int* smp; float* w;
  __host__ __device__ static size_t ints_per_map() { return 2 + (PLAN_E + 1) + 3 * (size_t)PLAN_E + 1; }
  __host__ __device__ PlanView(void* base, int k) {
    int* b = reinterpret_cast<int*>(base) + (size_t)k * ints_per_map();
    nseg = b; seg_start = b + 2; pix = seg_start + PLAN_E + 1; smp = pix + PLAN_E; w = reinterpret_cast<float*>(smp + PLAN_E);
  }
};
__global__ __launch_bounds__(1024) void scatter_plan_kernel(strotss_maps_t m, const float* __restrict__ idx, int n,
                                                            void* __restrict__ plan) {
  __shared__ unsigned long long key[PLAN_E];
  __shared__ int wsum[16];
  const int k = blockIdx.x, t = threadIdx.x;
  const unsigned long long INVALID = ~0ull;
  const int s_begin = m.sample_range ? m.sample_range[0] : 0, s_end = m.sample_range ? min(n, m.sample_range[1]) : n;
  for (int e = t; e < PLAN_E; e += 1024) {
    const int s = e >> 2, q = e & 3;
    unsigned long long kk = INVALID;
    if (s >= s_begin && s < s_end) {
      const SampleTap tp = sample_tap(m, k, idx[2 * s], idx[2 * s + 1], 1);
      const int ii = q == 0 ? tp.ia : q == 1 ? tp.ib : q == 2 ? tp.ic : tp.id;
      const float ww = q == 0 ? tp.wa : q == 1 ? tp.wb : q == 2 ? tp.wc : tp.wd;
      if (ww != 0.f) kk = ((unsigned long long)(unsigned)ii << 12) | (unsigned)e;
    }
    key[e] = kk;
  }
  __syncthreads();
  for (int size = 2; size <= PLAN_E; size <<= 1)
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int p = t; p < PLAN_E / 2; p += 1024) {
        const int lo = 2 * p - (p & (stride - 1));           // index of the pair's lower element
        const int hi = lo + stride;
        const bool up = (lo & size) == 0;
        const unsigned long long a = key[lo], b = key[hi];
        if ((a > b) == up) { key[lo] = b; key[hi] = a; }
      }
      __syncthreads();
    }
  // heads of the runs of equal pixels -> segment starts (ordered: thread t owns positions 4t .. 4t+3)
  PlanView pv(plan, k);
  int head[4], cnt = 0, valid = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int p = 4 * t + i;
    const unsigned long long a = key[p];
    const bool ok = a != INVALID;
    head[i] = ok && (p == 0 || (key[p - 1] >> 12) != (a >> 12));
    cnt += head[i]; valid += ok;
  }
  int incl = cnt, vincl = valid;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int u = __shfl_up(incl, o, 64), v = __shfl_up(vincl, o, 64);
    if ((t & 63) >= o) { incl += u; vincl += v; }
  }
  if ((t & 63) == 63) { wsum[t >> 6] = incl; }
  __syncthreads();
  int base = incl - cnt;
  for (int wv = 0; wv < (t >> 6); ++wv) base += wsum[wv];
  int total = 0;
#pragma unroll
  for (int wv = 0; wv < 16; ++wv) total += wsum[wv];
  __syncthreads();
  if ((t & 63) == 63) wsum[t >> 6] = vincl;
  __syncthreads();
  int nvalid = 0;
#pragma unroll
  for (int wv = 0; wv < 16; ++wv) nvalid += wsum[wv];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int p = 4 * t + i;
    const unsigned long long a = key[p];
    if (a == INVALID) continue;
    const int e = (int)(a & 4095u), s = e >> 2, q = e & 3;
    const SampleTap tp = sample_tap(m, k, idx[2 * s], idx[2 * s + 1], 1);
    pv.pix[p] = (int)(a >> 12);
    pv.smp[p] = s;
    pv.w[p] = q == 0 ? tp.wa : q == 1 ? tp.wb : q == 2 ? tp.wc : tp.wd;
    if (head[i]) pv.seg_start[base++] = p;
  }
  if (t == 0) { pv.nseg[0] = total; pv.seg_start[total] = nvalid; }
}
// one workgroup per destination pixel of map k (segment of the plan); off = first column of map k in gfeat
__global__ __launch_bounds__(256) void scatter_sorted_kernel(strotss_maps_t m, int k, int off, const void* __restrict__ plan,
                                                             const float* __restrict__ gfeat, int ld, int masked) {
  PlanView pv(const_cast<void*>(plan), k);
  const int sg = blockIdx.x;
  if (sg >= pv.nseg[0]) return;
  const int p0 = pv.seg_start[sg], p1 = pv.seg_start[sg + 1];
  const int c = m.c[k];
  const size_t pixel = (size_t)pv.pix[p0] * c;
  const float* act = m.map[k];
  float* dst = m.gmap[k];
  for (int ch = threadIdx.x; ch < c; ch += 256) {
    if (masked && !(act[pixel + ch] > 0.f)) continue;
    float acc = 0.f;
    for (int p = p0; p < p1; ++p) acc += pv.w[p] * gfeat[(size_t)pv.smp[p] * ld + off + ch];
    dst[pixel + ch] += acc;
  }
}

// ---------------------------------------------------------------- optimiser
__global__ __launch_bounds__(256) void rmsprop_kernel(strotss_tensors_t t, float lr, float rho, float eps) {
  const int k = blockIdx.y;
  float* var = t.var[k];
  float* rms = t.rms[k];
  const float* g = t.grad[k];
  const int64_t n = t.numel[k];
  const float omr = 1.0f - rho;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
    const float gv = g[e];
    const float r = rho * rms[e] + omr * (gv * gv);
    rms[e] = r;
    var[e] = var[e] - lr * gv / (sqrtf(r) + eps);
  }
}

// ---------------------------------------------------------------- postprocess
__global__ __launch_bounds__(256) void minmax_partial_kernel(const float* __restrict__ x, int64_t n,
                                                             float* __restrict__ ws) {
  __shared__ float red[4];
  float mn = INFINITY, mx = -INFINITY;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
    const float v = fminf(fmaxf(x[e], 0.f), 1.f);
    mn = fminf(mn, v);
    mx = fmaxf(mx, v);
  }
  mn = block_min_256(mn, red);
  mx = block_max_256(mx, red);
  if (threadIdx.x == 0) { ws[blockIdx.x] = mn; ws[1024 + blockIdx.x] = mx; }
}
__global__ __launch_bounds__(256) void postprocess_kernel(const float* __restrict__ x, int64_t n,
                                                          const float* __restrict__ ws, int nparts,
                                                          uint8_t* __restrict__ out) {
  __shared__ float red[4];
  float mn = INFINITY, mx = -INFINITY;
  for (int i = threadIdx.x; i < nparts; i += 256) { mn = fminf(mn, ws[i]); mx = fmaxf(mx, ws[1024 + i]); }
  mn = block_min_256(mn, red);
  mx = block_max_256(mx, red);
  const float range = mx - mn;   // == max(x - min) : float subtraction is monotone
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
    float v = fminf(fmaxf(x[e], 0.f), 1.f);
    v = (v - mn) / range;
    out[e] = (uint8_t)(v * 255.0f);   // truncating cast, as tf.cast(float -> uint8)
  }
}

int maps_ok(const strotss_maps_t* m) {
  if (!m || m->n_maps <= 0 || m->n_maps > STROTSS_MAX_MAPS) return 0;
  for (int k = 0; k < m->n_maps; ++k) {
    if (m->h[k] <= 0 || m->w[k] <= 0 || m->c[k] <= 0 || !m->map[k]) return 0;
    if (m->n_div[k] < 0 || m->n_div[k] > STROTSS_MAX_DIVS) return 0;
    if (m->rows[k] < 0 || m->row0[k] < 0 || m->row0[k] + m->rows[k] > m->h[k]) return 0;
  }
  return 1;
}

}  // namespace

extern "C" {

int strotss_abi_version(void) { return 8; }
const char* strotss_build_info(void) { return "libstrotss_hip gfx950 fp32-mfma " __DATE__ " " __TIME__; }

int strotss_resize_bilinear(const float* in, int ih, int iw, int c, float* out, int oh, int ow, float alpha,
                            const float* add, void* stream) {
  ST_CHECK_ARG(in && out && ih > 0 && iw > 0 && oh > 0 && ow > 0 && c > 0, STROTSS_EINVAL);
  const float sy = (float)ih / (float)oh, sx = (float)iw / (float)ow;   // CalculateResizeScale
  const dim3 grid((unsigned)cdiv(ow, 256), (unsigned)min(oh, 32768));
  if (c == 3)
    hipLaunchKernelGGL(resize_bilinear_kernel<3>, grid, dim3(256), 0, (hipStream_t)stream, in, ih, iw, c, out, oh, ow, sy, sx,
                       alpha, add);
  else if (c == 1)
    hipLaunchKernelGGL(resize_bilinear_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, in, ih, iw, c, out, oh, ow, sy, sx,
                       alpha, add);
  else
    hipLaunchKernelGGL(resize_bilinear_kernel<0>, grid, dim3(256), 0, (hipStream_t)stream, in, ih, iw, c, out, oh, ow, sy, sx,
                       alpha, add);
  ST_LAUNCH_RET();
}

int strotss_fold_pyramid(const strotss_pyramid_t* pyr, float* img, void* stream) {
  ST_CHECK_ARG(pyr && img && pyr->n_levels >= 2 && pyr->n_levels <= STROTSS_MAX_TENSORS, STROTSS_EINVAL);
  FoldLevels p;
  p.n = pyr->n_levels;
  for (int k = 0; k < p.n; ++k) {
    ST_CHECK_ARG(pyr->var[k] && pyr->h[k] > 0 && pyr->w[k] > 0, STROTSS_EINVAL);
    p.h[k] = pyr->h[k]; p.w[k] = pyr->w[k]; p.v[k] = pyr->var[k];
  }
  // footprint sides of a 32-pixel tile, level by level, against the LDS buffers: worst case over tile positions is
  // ceil(side * scale) + 2 per axis; shrinking pyramids only (every level at most as large as the one above it)
  int sy = FOLD_TILE, sx = FOLD_TILE;
  for (int k = 1; k < p.n; ++k) {
    ST_CHECK_ARG(p.h[k] <= p.h[k - 1] && p.w[k] <= p.w[k - 1], STROTSS_ERANGE);
    sy = min(p.h[k], (int)((long long)sy * p.h[k] / p.h[k - 1]) + 3);
    sx = min(p.w[k], (int)((long long)sx * p.w[k] / p.w[k - 1]) + 3);
    ST_CHECK_ARG(sy <= FOLD_REGION && sx <= FOLD_REGION, STROTSS_ERANGE);
  }
  hipLaunchKernelGGL(fold_pyramid_kernel, dim3((unsigned)cdiv(p.w[0], FOLD_TILE), (unsigned)cdiv(p.h[0], FOLD_TILE)), dim3(256),
                     0, (hipStream_t)stream, p, img);
  ST_LAUNCH_RET();
}

// whether two adjoint levels fit one launch: each level halves the one above it to within a pixel (then an 8 x 8 tile's
// footprint and the 16 x 16 tile above it span at most 28 rows / columns of the middle level)
static bool adjoint_pair_ok(int a, int b) { return b >= 1 && a >= 2 * b - 1 && a <= 2 * b + 1; }

int strotss_fold_pyramid_adjoint(const strotss_pyramid_t* g, void* stream) {
  ST_CHECK_ARG(g && g->n_levels >= 2 && g->n_levels <= STROTSS_MAX_TENSORS, STROTSS_EINVAL);
  for (int k = 0; k < g->n_levels; ++k) ST_CHECK_ARG(g->var[k] && g->h[k] > 0 && g->w[k] > 0, STROTSS_EINVAL);
  int k = 0;
  while (k + 1 < g->n_levels) {
    const bool pair = k + 2 < g->n_levels && adjoint_pair_ok(g->h[k], g->h[k + 1]) && adjoint_pair_ok(g->w[k], g->w[k + 1]) &&
                      adjoint_pair_ok(g->h[k + 1], g->h[k + 2]) && adjoint_pair_ok(g->w[k + 1], g->w[k + 2]);
    if (pair) {
      const unsigned gx = (unsigned)max(cdiv(g->w[k + 2], ADJ2_T2), cdiv(g->w[k + 1], 2 * ADJ2_T2));
      const unsigned gy = (unsigned)max(cdiv(g->h[k + 2], ADJ2_T2), cdiv(g->h[k + 1], 2 * ADJ2_T2));
      hipLaunchKernelGGL(resize_adjoint_pair_kernel, dim3(gx, gy), dim3(256), 0, (hipStream_t)stream,
                         (const float*)g->var[k], g->h[k], g->w[k], const_cast<float*>(g->var[k + 1]), g->h[k + 1], g->w[k + 1],
                         const_cast<float*>(g->var[k + 2]), g->h[k + 2], g->w[k + 2]);
      k += 2;
    } else {
      const int rc = strotss_resize_bilinear_adjoint(g->var[k], g->h[k], g->w[k], 3, const_cast<float*>(g->var[k + 1]),
                                                     g->h[k + 1], g->w[k + 1], stream);
      if (rc != 0) return rc;
      k += 1;
    }
  }
  ST_LAUNCH_RET();
}

int strotss_resize_bilinear_adjoint(const float* gout, int oh, int ow, int c, float* gin, int ih, int iw,
                                    void* stream) {
  ST_CHECK_ARG(gout && gin && ih > 0 && iw > 0 && oh > 0 && ow > 0 && c > 0, STROTSS_EINVAL);
  const float sy = (float)ih / (float)oh, sx = (float)iw / (float)ow;
  const dim3 grid((unsigned)cdiv(iw, 256), (unsigned)min(ih, 32768));
  if (c == 3)
    hipLaunchKernelGGL(resize_adjoint_kernel<3>, grid, dim3(256), 0, (hipStream_t)stream, gout, oh, ow, c, gin, ih, iw, sy, sx);
  else if (c == 1)
    hipLaunchKernelGGL(resize_adjoint_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, gout, oh, ow, c, gin, ih, iw, sy, sx);
  else
    hipLaunchKernelGGL(resize_adjoint_kernel<0>, grid, dim3(256), 0, (hipStream_t)stream, gout, oh, ow, c, gin, ih, iw, sy, sx);
  ST_LAUNCH_RET();
}

int strotss_hypercol_gather(const strotss_maps_t* maps, const float* idx, int n, int bilinear, float* out,
                            int ld, void* stream) {
  ST_CHECK_ARG(maps_ok(maps) && idx && out && n > 0, STROTSS_EINVAL);
  int d = 0;
  for (int k = 0; k < maps->n_maps; ++k) d += maps->c[k];
  ST_CHECK_ARG(ld >= d, STROTSS_EINVAL);
  strotss_maps_t m = *maps;
  m.window_drop = 0;          // the gather ALWAYS clamps into the window; dropping is the scatter's (adjoint's) business
  hipLaunchKernelGGL(hypercol_gather_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, m, idx, bilinear,
                     out, ld, d);
  ST_LAUNCH_RET();
}

int strotss_hypercol_gather2(const strotss_maps_t* maps_a, const strotss_maps_t* maps_b, const float* idx, int n,
                             int bilinear, float* out_a, float* out_b, int ld, float* zero, int zero_rows, void* stream) {
  ST_CHECK_ARG(maps_ok(maps_a) && maps_ok(maps_b) && idx && out_a && out_b && n > 0, STROTSS_EINVAL);
  ST_CHECK_ARG(!zero || zero_rows > 0, STROTSS_EINVAL);
  int da = 0, db = 0;
  for (int k = 0; k < maps_a->n_maps; ++k) da += maps_a->c[k];
  for (int k = 0; k < maps_b->n_maps; ++k) db += maps_b->c[k];
  ST_CHECK_ARG(ld >= da && ld >= db, STROTSS_EINVAL);
  strotss_maps_t a = *maps_a, b = *maps_b;
  a.window_drop = b.window_drop = 0;        // the gather always clamps into the window (see strotss_hypercol_gather)
  hipLaunchKernelGGL(hypercol_gather2_kernel, dim3(2 * n), dim3(256), 0, (hipStream_t)stream, a, b, idx, n, bilinear, out_a,
                     out_b, ld, da, db, zero, zero_rows);
  ST_LAUNCH_RET();
}

int strotss_hypercol_scatter(const strotss_maps_t* maps, const float* idx, int n, const float* gfeat, int ld,
                             int relu_mask_from, int map_begin, int map_end, void* stream) {
  ST_CHECK_ARG(maps_ok(maps) && idx && gfeat && n > 0, STROTSS_EINVAL);
  ST_CHECK_ARG(map_begin >= 0 && map_begin < map_end && map_end <= maps->n_maps, STROTSS_ERANGE);
  int d = 0;
  for (int k = 0; k < maps->n_maps; ++k) {
    if (k >= map_begin && k < map_end) ST_CHECK_ARG(maps->gmap[k] != nullptr, STROTSS_EINVAL);
    d += maps->c[k];
  }
  ST_CHECK_ARG(ld >= d, STROTSS_EINVAL);
  // maps of at most SCATTER_DENSE_MAX_PIX pixels: one workgroup per (pixel, channel chunk) instead of contended atomics
  // (STROTSS_SCATTER_DENSE: 0 = never, 1 or unset = up to SCATTER_DENSE_MAX_PIX pixels, n > 1 = up to n pixels)
  static const int dense_pix = [] {
    const char* e = getenv("STROTSS_SCATTER_DENSE");
    const int v = e ? atoi(e) : 1;
    return v == 1 ? SCATTER_DENSE_MAX_PIX : v;
  }();
  unsigned dense_mask = 0;
  int dense_blocks = 0;
  for (int k = map_begin; k < map_end; ++k) {
    const int pix = (maps->rows[k] > 0 ? maps->rows[k] : maps->h[k]) * maps->w[k];
    if (pix <= dense_pix) { dense_mask |= 1u << k; dense_blocks += pix * ((maps->c[k] + 63) / 64); }
  }
  hipLaunchKernelGGL(hypercol_scatter_kernel, dim3(n + dense_blocks), dim3(256), 0, (hipStream_t)stream, *maps, idx, gfeat,
                     ld, relu_mask_from, map_begin, map_end, n, dense_mask, dense_blocks);
  ST_LAUNCH_RET();
}

size_t strotss_hypercol_scatter_plan_bytes(int n_maps) {
  return n_maps > 0 ? (size_t)n_maps * PlanView::ints_per_map() * sizeof(int) : 0;
}

int strotss_hypercol_scatter_plan(const strotss_maps_t* maps, const float* idx, int n, void* plan, size_t plan_bytes,
                                  void* stream) {
  ST_CHECK_ARG(maps_ok(maps) && idx && plan && n > 0, STROTSS_EINVAL);
  ST_CHECK_ARG(4 * n <= PLAN_E, STROTSS_ERANGE);
  ST_CHECK_ARG(plan_bytes >= strotss_hypercol_scatter_plan_bytes(maps->n_maps), STROTSS_EINVAL);
  for (int k = 0; k < maps->n_maps; ++k) {
    const long long px = (long long)(maps->rows[k] > 0 ? maps->rows[k] : maps->h[k]) * maps->w[k];
    ST_CHECK_ARG(px < (1ll << 31), STROTSS_ERANGE);
  }
  hipLaunchKernelGGL(scatter_plan_kernel, dim3(maps->n_maps), dim3(1024), 0, (hipStream_t)stream, *maps, idx, n, plan);
  ST_LAUNCH_RET();
}

int strotss_hypercol_scatter_sorted(const strotss_maps_t* maps, const void* plan, int n, const float* gfeat, int ld,
                                    int relu_mask_from, int map_begin, int map_end, void* stream) {
  ST_CHECK_ARG(maps_ok(maps) && plan && gfeat && n > 0 && 4 * n <= PLAN_E, STROTSS_EINVAL);
  ST_CHECK_ARG(map_begin >= 0 && map_begin < map_end && map_end <= maps->n_maps, STROTSS_ERANGE);
  int off = 0;
  for (int k = 0; k < map_end; ++k) {
    if (k >= map_begin) {
      ST_CHECK_ARG(maps->gmap[k] != nullptr && off + maps->c[k] <= ld, STROTSS_EINVAL);
      hipLaunchKernelGGL(scatter_sorted_kernel, dim3(4 * n), dim3(256), 0, (hipStream_t)stream, *maps, k, off, plan, gfeat,
                         ld, k >= relu_mask_from ? 1 : 0);
    }
    off += maps->c[k];
  }
  ST_LAUNCH_RET();
}

int strotss_rmsprop_step(const strotss_tensors_t* t, float lr, float rho, float eps, void* stream) {
  ST_CHECK_ARG(t && t->n_tensors > 0 && t->n_tensors <= STROTSS_MAX_TENSORS, STROTSS_ERANGE);
  int64_t mx = 0;
  for (int k = 0; k < t->n_tensors; ++k) {
    ST_CHECK_ARG(t->var[k] && t->rms[k] && t->grad[k] && t->numel[k] > 0, STROTSS_EINVAL);
    mx = t->numel[k] > mx ? t->numel[k] : mx;
  }
  dim3 grid((unsigned)((mx + 255) / 256 > 2048 ? 2048 : (mx + 255) / 256), t->n_tensors);
  hipLaunchKernelGGL(rmsprop_kernel, grid, dim3(256), 0, (hipStream_t)stream, *t, lr, rho, eps);
  ST_LAUNCH_RET();
}

int strotss_postprocess(const float* img, int64_t numel, uint8_t* out, float* workspace, void* stream) {
  ST_CHECK_ARG(img && out && workspace && numel > 0, STROTSS_EINVAL);
  const int nparts = (int)((numel + 255) / 256 > 1024 ? 1024 : (numel + 255) / 256);
  hipLaunchKernelGGL(minmax_partial_kernel, dim3(nparts), dim3(256), 0, (hipStream_t)stream, img, numel,
                     workspace);
  hipLaunchKernelGGL(postprocess_kernel, dim3(nparts), dim3(256), 0, (hipStream_t)stream, img, numel,
                     workspace, nparts, out);
  ST_LAUNCH_RET();
}

}  // extern "C"
