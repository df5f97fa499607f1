// Winograd F(2x2, 3x3) form of the VGG 3x3 convolutions (nn/model.py:44-48) for the deep layers,
// where the f32 MFMA -- not HBM -- is the limit: 16 Winograd-domain GEMMs of (tiles x Cin)x(Cin x Cout)
// do 4*HW*Cin*Cout MACs instead of the direct form's 9*HW*Cin*Cout (2.25x fewer).
//
//   V[p] = (B^T d B)[p]     input transform   (one 4x4 input patch per 2x2 output tile, p = 0..15)
//   M[p] = V[p] U[p]^T      batched GEMM on the pipelined f32-MFMA kernel (gemm.hip)
//   Y    = A^T M A          output transform + bias + ReLU (forward) or ReLU mask (data-gradient)
//
// U[p] = (G g G^T)[p] is pre-computed once per model on the host (frozen weights).  The two
// transforms are streaming kernels (V and M are 4x the activation size; at the sizes used they
// live in the 256 MiB Infinity Cache between producer and consumer).
#include <stdlib.h>

#include "internal.h"
#include "mfma_x3.h"

namespace {

// V: (16, T, C) with T = ceil(H/2)*ceil(W/2); tile (ty,tx) reads input rows 2ty-1..2ty+2, cols 2tx-1..2tx+2
__global__ __launch_bounds__(256) void winograd_in_kernel(const float* __restrict__ in, int H, int W, int C4,
                                                          int TH, int TW, size_t tile0, size_t T,
                                                          float* __restrict__ V) {
  // this launch covers tiles [tile0, tile0 + T) of the image; V holds (16, T, C) for them
  const size_t total = T * C4;
  const f32x4* src = reinterpret_cast<const f32x4*>(in);
  f32x4* dst = reinterpret_cast<f32x4*>(V);
  auto put = [&](int pos, size_t tile, int c, const f32x4 v) { dst[((size_t)pos * T + tile) * C4 + c] = v; };
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const int c = (int)(e % C4);
    const size_t tile = e / C4;
    const size_t gt = tile0 + tile;
    const int tx = (int)(gt % TW), ty = (int)(gt / TW);
    f32x4 d[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int y = 2 * ty - 1 + r;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int x = 2 * tx - 1 + q;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (y >= 0 && y < H && x >= 0 && x < W) v = src[((size_t)y * W + x) * C4 + c];
        d[r][q] = v;
      }
    }
    // t = B^T d : rows (d0-d2, d1+d2, d2-d1, d1-d3)
    f32x4 tt[4][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      tt[0][q] = d[0][q] - d[2][q];
      tt[1][q] = d[1][q] + d[2][q];
      tt[2][q] = d[2][q] - d[1][q];
      tt[3][q] = d[1][q] - d[3][q];
    }
    // V = t B : columns (t0-t2, t1+t2, t2-t1, t1-t3)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const f32x4 v0 = tt[r][0] - tt[r][2], v1 = tt[r][1] + tt[r][2], v2 = tt[r][2] - tt[r][1],
                  v3 = tt[r][1] - tt[r][3];
      put(r * 4 + 0, tile, c, v0);
      put(r * 4 + 1, tile, c, v1);
      put(r * 4 + 2, tile, c, v2);
      put(r * 4 + 3, tile, c, v3);
    }
  }
}

// Y = A^T M A, A^T = [[1,1,1,0],[0,1,-1,-1]];  out = relu(Y + bias)  or  (mask > 0 ? Y : 0)
__global__ __launch_bounds__(256) void winograd_out_kernel(const float* __restrict__ Mw, int H, int W, int C4,
                                                           int TH, int TW, size_t tile0, size_t T,
                                                           const float* __restrict__ bias,
                                                           const float* __restrict__ mask, int relu,
                                                           float* __restrict__ out) {
  const size_t total = T * C4;
  const f32x4* src = reinterpret_cast<const f32x4*>(Mw);
  const f32x4* msk = reinterpret_cast<const f32x4*>(mask);
  f32x4* dst = reinterpret_cast<f32x4*>(out);
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const int c = (int)(e % C4);
    const size_t tile = e / C4;
    const size_t gt = tile0 + tile;
    const int tx = (int)(gt % TW), ty = (int)(gt / TW);
    f32x4 m[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int q = 0; q < 4; ++q) m[r][q] = src[((size_t)(r * 4 + q) * T + tile) * C4 + c];
    f32x4 s[2][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      s[0][q] = m[0][q] + m[1][q] + m[2][q];
      s[1][q] = m[1][q] - m[2][q] - m[3][q];
    }
    f32x4 b = {0.f, 0.f, 0.f, 0.f};
    if (bias) b = reinterpret_cast<const f32x4*>(bias)[c];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int y = 2 * ty + r;
      if (y >= H) continue;
      const f32x4 y0 = s[r][0] + s[r][1] + s[r][2] + b;
      const f32x4 y1 = s[r][1] - s[r][2] - s[r][3] + b;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int x = 2 * tx + q;
        if (x >= W) continue;
        f32x4 v = q ? y1 : y0;
        const size_t o = ((size_t)y * W + x) * C4 + c;
        if (relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
        if (mask) {
          const f32x4 k = msk[o];
          v[0] = k[0] > 0.f ? v[0] : 0.f; v[1] = k[1] > 0.f ? v[1] : 0.f;
          v[2] = k[2] > 0.f ? v[2] : 0.f; v[3] = k[3] > 0.f ? v[3] : 0.f;
        }
        dst[o] = v;
      }
    }
  }
}

// ---------------------------------------------------------------- F(4x4, 3x3)
// 36 transform-domain GEMMs per layer, 2.25 MACs per output (F(2x2,3x3): 4, direct: 9), V / M are 2.25x the
// activation size (F(2x2,3x3): 4x).  Matrices of Lavin & Gray with points (0, +-1, +-2, inf); in f32 the
// conv error grows to ~1e-5 of the output range (F(2x2,3x3): ~5e-7, direct: ~2e-7), see DESIGN.md.
__device__ __forceinline__ void bt6(f32x4 (&d)[6]) {       // in-place B^T d
  const f32x4 d0 = d[0], d1 = d[1], d2 = d[2], d3 = d[3], d4 = d[4], d5 = d[5];
  d[0] = 4.f * d0 - 5.f * d2 + d4;
  d[1] = -4.f * d1 - 4.f * d2 + d3 + d4;
  d[2] = 4.f * d1 - 4.f * d2 - d3 + d4;
  d[3] = -2.f * d1 - d2 + 2.f * d3 + d4;
  d[4] = 2.f * d1 - d2 - 2.f * d3 + d4;
  d[5] = 4.f * d1 - 5.f * d3 + d5;
}
// Element index -> (tile, channel quad) in 32-bit arithmetic (a shift when C4 is a power of two, as in VGG): the 64-bit
// `%` / `/` these transforms used to do four times per thread cost more instructions than the transform itself.
__device__ __forceinline__ void split_index(unsigned e, int C4, int c4_shift, unsigned& tile, int& c) {
  if (c4_shift >= 0) { tile = e >> c4_shift; c = (int)(e & (unsigned)(C4 - 1)); }
  else { tile = e / (unsigned)C4; c = (int)(e - tile * (unsigned)C4); }
}
static int log2_or_minus1(int v) { return (v > 0 && (v & (v - 1)) == 0) ? __builtin_ctz((unsigned)v) : -1; }

// V: (36, T, C), T = ceil(H/4)*ceil(W/4); tile (ty,tx) reads input rows 4ty-1..4ty+4, cols 4tx-1..4tx+4
__global__ __launch_bounds__(256) void winograd43_in_kernel(const float* __restrict__ in, int H, int W, int C4,
                                                            int TH, int TW, float* __restrict__ V, int c4_shift) {
  const size_t T = (size_t)TH * TW;
  const unsigned total = (unsigned)(T * C4);
  const f32x4* src = reinterpret_cast<const f32x4*>(in);
  f32x4* dst = reinterpret_cast<f32x4*>(V);
  for (unsigned e = blockIdx.x * 256u + threadIdx.x; e < total; e += gridDim.x * 256u) {
    unsigned tile; int c;
    split_index(e, C4, c4_shift, tile, c);
    const int ty = (int)(tile / (unsigned)TW), tx = (int)(tile - (unsigned)ty * (unsigned)TW);
    f32x4 d[6][6];
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      const int y = 4 * ty - 1 + r;
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        const int x = 4 * tx - 1 + q;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (y >= 0 && y < H && x >= 0 && x < W) v = src[((size_t)y * W + x) * C4 + c];
        d[r][q] = v;
      }
    }
#pragma unroll
    for (int q = 0; q < 6; ++q) {        // columns: t = B^T d
      f32x4 col[6] = {d[0][q], d[1][q], d[2][q], d[3][q], d[4][q], d[5][q]};
      bt6(col);
#pragma unroll
      for (int r = 0; r < 6; ++r) d[r][q] = col[r];
    }
#pragma unroll
    for (int r = 0; r < 6; ++r) {        // rows: V = t B
      bt6(d[r]);
#pragma unroll
      for (int q = 0; q < 6; ++q) dst[((size_t)(r * 6 + q) * T + tile) * C4 + c] = d[r][q];
    }
  }
}

// The same transform writing V as x3 panels (mfma_x3.h) for the bf16x3 GEMM core: per position p a panel of
// T rows x C columns.  Lane order: 8 lanes = the 32 channels of one K-block of one tile (128 B read per pixel),
// 8 consecutive tiles per wave: 512 contiguous bytes per plane, position and wave store.
__global__ __launch_bounds__(256) void winograd43_in_x3_kernel(const float* __restrict__ in, int H, int W, int C4,
                                                               int TH, int TW, __bf16* __restrict__ V, int kb_shift) {
  const size_t T = (size_t)TH * TW;
  const int KB = C4 >> 3;
  const unsigned total = (unsigned)(((T + 7) >> 3) * 64 * KB);
  const f32x4* src = reinterpret_cast<const f32x4*>(in);
  const size_t panel = 3 * T * (size_t)C4 * 4;
  for (unsigned e = blockIdx.x * 256u + threadIdx.x; e < total; e += gridDim.x * 256u) {
    const int c8 = (int)(e & 7), tlo = (int)((e >> 3) & 7);
    unsigned r8; int kb;
    split_index(e >> 6, KB, kb_shift, r8, kb);
    const unsigned tile = r8 * 8 + tlo;
    if (tile >= T) continue;
    const int c = kb * 8 + c8;
    const int ty = (int)(tile / (unsigned)TW), tx = (int)(tile - (unsigned)ty * (unsigned)TW);
    f32x4 d[6][6];
#pragma unroll
    for (int rr = 0; rr < 6; ++rr) {
      const int y = 4 * ty - 1 + rr;
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        const int x = 4 * tx - 1 + q;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (y >= 0 && y < H && x >= 0 && x < W) v = src[((size_t)y * W + x) * C4 + c];
        d[rr][q] = v;
      }
    }
#pragma unroll
    for (int q = 0; q < 6; ++q) {        // columns: t = B^T d
      f32x4 col[6] = {d[0][q], d[1][q], d[2][q], d[3][q], d[4][q], d[5][q]};
      bt6(col);
#pragma unroll
      for (int rr = 0; rr < 6; ++rr) d[rr][q] = col[rr];
    }
#pragma unroll
    for (int rr = 0; rr < 6; ++rr) {     // rows: V = t B
      bt6(d[rr]);
#pragma unroll
      for (int q = 0; q < 6; ++q) x3_store4(V + (size_t)(rr * 6 + q) * panel, T, tile, 4 * c, d[rr][q]);
    }
  }
}

// Y = A^T M A (4x4 outputs per tile);  out = relu(Y + bias)  or  (mask > 0 ? Y : 0)
__global__ __launch_bounds__(256) void winograd43_out_kernel(const float* __restrict__ Mw, int H, int W, int C4,
                                                             int TH, int TW, const float* __restrict__ bias,
                                                             const float* __restrict__ mask, int relu,
                                                             float* __restrict__ out, int c4_shift, size_t Tstride,
                                                             const u32x4* __restrict__ bits_in, u32x4* __restrict__ bits_out,
                                                             f32x4* __restrict__ pool, unsigned* __restrict__ pool_code,
                                                             int accumulate) {
  // Tstride: tiles per position plane of Mw (>= TH * TW)
  // accumulate (data-gradient): out += instead of out = (the layer's taps were scattered into `out` before the backward pass)
  // bits_in / bits_out: sign words of the tile grid (include/strotss_hip.h: relu_bits), one per (tile, channel); with
  // bits_in the mask comes from them (16 bytes per thread instead of 16 x 16 bytes of activations)
  // pool (forward): also the 2x2/2 max-pool of the result and (pool_code) its argmax codes, exactly as maxpool2_fwd_kernel
  // would produce them from `out` -- the tile's 4 x 4 outputs hold four whole windows
  const size_t T = (size_t)TH * TW;
  const unsigned total = (unsigned)(T * C4);
  const f32x4* src = reinterpret_cast<const f32x4*>(Mw);
  const f32x4* msk = reinterpret_cast<const f32x4*>(mask);
  f32x4* dst = reinterpret_cast<f32x4*>(out);
  for (unsigned e = blockIdx.x * 256u + threadIdx.x; e < total; e += gridDim.x * 256u) {
    unsigned tile; int c;
    split_index(e, C4, c4_shift, tile, c);
    const int ty = (int)(tile / (unsigned)TW), tx = (int)(tile - (unsigned)ty * (unsigned)TW);
    f32x4 s[4][6];                       // s = A^T M   (rows of A^T applied down the columns of M)
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      f32x4 m[6];
#pragma unroll
      for (int r = 0; r < 6; ++r) m[r] = src[((size_t)(r * 6 + q) * Tstride + tile) * C4 + c];
      s[0][q] = m[0] + m[1] + m[2] + m[3] + m[4];
      s[1][q] = m[1] - m[2] + 2.f * m[3] - 2.f * m[4];
      s[2][q] = m[1] + m[2] + 4.f * m[3] + 4.f * m[4];
      s[3][q] = m[1] - m[2] + 8.f * m[3] - 8.f * m[4] + m[5];
    }
    f32x4 b = {0.f, 0.f, 0.f, 0.f};
    if (bias) b = reinterpret_cast<const f32x4*>(bias)[c];
    u32x4 kw = {0u, 0u, 0u, 0u}, ow = {0u, 0u, 0u, 0u};
    if (bits_in) kw = bits_in[e];                      // e = tile * C4 + c
    f32x4 prev[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int y = 4 * ty + r;
      if (y >= H) continue;
      f32x4 yv[4];
      yv[0] = s[r][0] + s[r][1] + s[r][2] + s[r][3] + s[r][4] + b;
      yv[1] = s[r][1] - s[r][2] + 2.f * s[r][3] - 2.f * s[r][4] + b;
      yv[2] = s[r][1] + s[r][2] + 4.f * s[r][3] + 4.f * s[r][4] + b;
      yv[3] = s[r][1] - s[r][2] + 8.f * s[r][3] - 8.f * s[r][4] + s[r][5] + b;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int x = 4 * tx + q;
        if (x >= W) continue;
        f32x4 v = yv[q];
        const size_t o = ((size_t)y * W + x) * C4 + c;
        if (bits_out) {
#pragma unroll
          for (int j = 0; j < 4; ++j) ow[j] |= (v[j] > 0.f ? 1u : 0u) << (8 * r + q);
        }
        if (relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
        if (bits_in) {
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = ((kw[j] >> (8 * r + q)) & 1u) ? v[j] : 0.f;
        } else if (mask) {
          const f32x4 k = msk[o];
          v[0] = k[0] > 0.f ? v[0] : 0.f; v[1] = k[1] > 0.f ? v[1] : 0.f;
          v[2] = k[2] > 0.f ? v[2] : 0.f; v[3] = k[3] > 0.f ? v[3] : 0.f;
        }
        dst[o] = accumulate ? dst[o] + v : v;
      }
      if (pool) {
        if ((r & 1) == 0) {
#pragma unroll
          for (int q = 0; q < 4; ++q) prev[q] = yv[q];
        } else {
          const int PH = H >> 1, PW = W >> 1, py = y >> 1;
#pragma unroll
          for (int pc = 0; pc < 2; ++pc) {
            const int px = 2 * tx + pc;
            if (py >= PH || px >= PW) continue;
            f32x4 pv;
            unsigned packed = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float v4[4] = {prev[2 * pc][j], prev[2 * pc + 1][j], yv[2 * pc][j], yv[2 * pc + 1][j]};
              int best = 0;
              float bv = v4[0];
#pragma unroll
              for (int k = 1; k < 4; ++k)
                if (v4[k] > bv) { bv = v4[k]; best = k; }
              pv[j] = fmaxf(bv, 0.f);
              packed |= (unsigned)(bv > 0.f ? best : 4) << (8 * j);
            }
            const size_t po = ((size_t)py * PW + px) * C4 + c;
            pool[po] = pv;
            if (pool_code) pool_code[po] = packed;
          }
        }
      }
    }
    if (bits_out) bits_out[e] = ow;
  }
}

// The 36 GEMMs on the bf16x3 core (mfma_x3.h) where the weights' x3 panels are given and the launch has at least
// STROTSS_X3_MIN_TILES 128 x 128 output tiles (default 1024 = four rounds of the 256 CUs; with fewer -- idle CUs,
// half-empty 128-row tiles -- the f32 core's smaller tiles win: 512-px scale 2.28 vs 2.41 ms/step).  The GEMMs run
// 1.55x faster (512->512 @128x128 px: 172 -> 112 us, the layer 209 -> 153 us).  Default ON since round 2
// (STROTSS_X3_CONV=0 switches it off): five alternating A/B runs of the 1024-px step on one box gave 5.294 +- 0.004 ms
// (off) against 5.117 +- 0.010 ms (on), -3.4 % (profiles/r02_ab_x3conv.txt); round 1's four single runs on four boxes
// (+2.8 %, +-0, -1.5 %, -1.9 %) were inside their own noise.  The cost matrices of the loss section use the same core.
static long x3_min_tiles() {
  static long min_tiles = -1;
  if (min_tiles < 0) { const char* m = getenv("STROTSS_X3_MIN_TILES"); min_tiles = m ? atol(m) : 1024; }
  return min_tiles;
}
// with fewer than STROTSS_X3_MIN_TILES 128 x 128 tiles the GEMMs run on 64 x 64 tiles (two workgroups per CU) as long
// as there are that many of THOSE (block5 at 1024 px, block4 at 512 px: 36 x (256 x 512 x 512) = 1152 small tiles)
static bool x3_enabled(size_t T, int cout) {
  static int on = -1;
  if (on < 0) {
    const char* c = getenv("STROTSS_X3_CONV"); on = c ? atoi(c) : 1;
    const char* e = getenv("STROTSS_X3"); if (e && atoi(e) == 0) on = 0;
  }
  return on != 0 && (long)((T + 63) / 64) * ((cout + 63) / 64) * 36 >= x3_min_tiles();
}

static int x3_min_cout() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("STROTSS_X3_MIN_COUT"); v = e ? atoi(e) : 256; }
  return v;
}
// layers the fused kernel could take but the three-kernel form with bf16x3 GEMMs runs faster: 256 output channels with
// enough 128 x 128 GEMM tiles (block3 at 1024 px: step 5.06 -> 4.99 ms; at 512 px the fused kernel wins, 1.99 vs 2.03 ms)
static bool winograd43_prefers_x3(int h, int w, int cout) {
  const size_t T = (size_t)((h + 3) / 4) * ((w + 3) / 4);
  return cout >= x3_min_cout() && x3_enabled(T, cout) && (long)((T + 127) / 128) * ((cout + 127) / 128) * 36 >= x3_min_tiles();
}

// Measurement hook (strotss_debug_winograd_stages): which stages of the F(4x4,3x3) three-kernel form are launched.
// bit 0: input transform, bit 1: the 36 GEMMs, bit 2: output transform.  Process-wide, not thread-safe, default all.
static int g_wino_stages = 7;

static int winograd43_run(const float* in, int h, int w, int cin, const float* U, const float* Upacked,
                          const void* Ux3, const float* bias, int cout, const float* mask, int relu, float* out,
                          float* pool_out, unsigned char* pool_code, void* workspace, size_t workspace_bytes,
                          hipStream_t st, const unsigned* bits_in = nullptr, unsigned* bits_out = nullptr, int accumulate = 0) {
  // 256 output channels and enough tiles for the bf16x3 GEMMs: the three-kernel form wins (1024-px step 5.102 -> 5.039 ms,
  // three alternating runs each); STROTSS_X3_MIN_COUT (default 256) moves the border
  const bool prefer_x3 = Ux3 && cin % 32 == 0 && winograd43_prefers_x3(h, w, cout);
  if (!prefer_x3 && Upacked && cin % 32 == 0 && st_winograd43_fused_enabled(h, w, cout))      // everything on chip
    return st_winograd43_fused(in, h, w, cin, Upacked, bias, cout, mask, relu, out, pool_out, pool_code, bits_in, bits_out, st,
                               accumulate);
  const int TH = (h + 3) / 4, TW = (w + 3) / 4;
  const size_t T = (size_t)TH * TW;
  const bool x3 = Ux3 && cin % 32 == 0 && x3_enabled(T, cout);
  if (T * (size_t)(max(cin, cout) / 4 + 8) >= (1ull << 32)) return STROTSS_ERANGE;     // 32-bit element indices in the transforms
  Workspace ws(workspace, workspace_bytes);
  float* V = ws.take<float>(36 * T * cin * 3 / 2);           // f32 V, or its x3 panels (3 bf16 per value)
  float* Mw = ws.take<float>(36 * T * cout);
  if (!ws.ok()) return STROTSS_EINVAL;
  const size_t tout = T * (cout / 4);
  const size_t Tstride = T;
  int rc;
  const int stages = g_wino_stages;
  if (x3) {
    const size_t tin = ((T + 7) / 8) * 8 * (cin / 4);
    if (stages & 1) hipLaunchKernelGGL(winograd43_in_x3_kernel, dim3((unsigned)min((size_t)16384, (tin + 255) / 256)), dim3(256), 0, st,
                       in, h, w, cin / 4, TH, TW, reinterpret_cast<__bf16*>(V), log2_or_minus1(cin / 32));
    if (!(stages & 2)) rc = 0;
    else rc = st_gemm_x3_batched(V, Ux3, Mw, cout, (long long)T * cout, (int)T, cout, cin, 36, st, x3_min_tiles());
  } else {
    const size_t tin = T * (cin / 4);
    if (stages & 1) hipLaunchKernelGGL(winograd43_in_kernel, dim3((unsigned)min((size_t)16384, (tin + 255) / 256)), dim3(256), 0, st, in,
                       h, w, cin / 4, TH, TW, V, log2_or_minus1(cin / 4));
    rc = (stages & 2) ? st_gemm_nt_batched(V, cin, (long long)T * cin, U, cin, (long long)cout * cin, Mw, cout,
                                           (long long)T * cout, (int)T, cout, cin, 36, st) : 0;
  }
  if (rc != 0) return rc;
  // STROTSS_WINO_OUT_POOL=0: the pooled copy by a pooling launch instead of the output transform's epilogue (A/B)
  static const bool out_pools = [] { const char* v = getenv("STROTSS_WINO_OUT_POOL"); return !v || atoi(v) != 0; }();
  if (stages & 4) hipLaunchKernelGGL(winograd43_out_kernel, dim3((unsigned)min((size_t)16384, (tout + 255) / 256)), dim3(256), 0, st,
                     Mw, h, w, cout / 4, TH, TW, bias, mask, relu, out, log2_or_minus1(cout / 4), Tstride,
                     reinterpret_cast<const u32x4*>(bits_in), reinterpret_cast<u32x4*>(bits_out),
                     reinterpret_cast<f32x4*>(out_pools ? pool_out : nullptr),
                     reinterpret_cast<unsigned*>(out_pools ? pool_code : nullptr), accumulate);
  if (pool_out && !out_pools) return st_maxpool2_fwd(out, h, w, cout, pool_out, pool_code, st);
  ST_LAUNCH_RET();
}

// Optional tiling of the tile range into passes whose V and M (16*Tc*(cin+cout) floats) would stay in the
// 256 MiB Infinity Cache (STROTSS_WINO_CHUNK_MB > 0).  Measured on MI355X: every chunk size tried (48-160 MB)
// is SLOWER than one pass (more, smaller launches; tails), so the default is 0 = one pass.
static size_t wino_chunk_tiles(size_t T, int cin, int cout) {
  static long mb = -1;
  if (mb < 0) { const char* e = getenv("STROTSS_WINO_CHUNK_MB"); mb = e ? atol(e) : 0; }
  if (mb == 0) return T;
  size_t tc = ((size_t)mb << 20) / ((size_t)16 * (cin + cout) * sizeof(float));
  tc = tc / 128 * 128;
  if (tc < 1024) tc = 1024;
  return tc < T ? tc : T;
}

int winograd_run(const float* in, int h, int w, int cin, const float* U, const float* bias, int cout,
                 const float* mask, int relu, float* out, void* workspace, size_t workspace_bytes,
                 hipStream_t st) {
  const int TH = (h + 1) / 2, TW = (w + 1) / 2;
  const size_t T = (size_t)TH * TW;
  const size_t Tc = wino_chunk_tiles(T, cin, cout);
  Workspace ws(workspace, workspace_bytes);
  float* V = ws.take<float>(16 * Tc * cin);
  float* Mw = ws.take<float>(16 * Tc * cout);
  if (!ws.ok()) return STROTSS_EINVAL;
  for (size_t t0 = 0; t0 < T; t0 += Tc) {
    const size_t tc = (T - t0 < Tc) ? T - t0 : Tc;
    const size_t tin = tc * (cin / 4), tout = tc * (cout / 4);
    const dim3 gin((unsigned)min((size_t)16384, (tin + 255) / 256)), gout((unsigned)min((size_t)16384, (tout + 255) / 256));
    hipLaunchKernelGGL(winograd_in_kernel, gin, dim3(256), 0, st, in, h, w, cin / 4, TH, TW, t0, tc, V);
    const int rc = st_gemm_nt_batched(V, cin, (long long)tc * cin, U, cin, (long long)cout * cin, Mw, cout,
                                      (long long)tc * cout, (int)tc, cout, cin, 16, st);
    if (rc != 0) return rc;
    hipLaunchKernelGGL(winograd_out_kernel, gout, dim3(256), 0, st, Mw, h, w, cout / 4, TH, TW, t0, tc, bias, mask, relu,
                       out);
  }
  ST_LAUNCH_RET();
}

}  // namespace

// U[a * (m + 2) + b][n][k] = (G g[n][k] G^T)[a][b] in float64, rounded once to float32 (one thread per (n, k) pair)
__global__ __launch_bounds__(256) void winograd_weights_kernel(const float* __restrict__ g, long long nk, int tile_m,
                                                              float* __restrict__ U) {
  // Lavin & Gray: F(2x2,3x3) and F(4x4,3x3) (points 0, +-1, +-2, inf)
  const double G2[4][3] = {{1.0, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1.0}};
  const double G4[6][3] = {{1.0 / 4, 0, 0}, {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                           {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0, 0, 1.0}};
  const int P = tile_m + 2;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < nk; e += (long long)gridDim.x * 256) {
    double w[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int q = 0; q < 3; ++q) w[r][q] = (double)g[e * 9 + r * 3 + q];
    for (int a = 0; a < P; ++a) {
      double t[3];                                    // (G g)[a][q]
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        t[q] = 0.0;
#pragma unroll
        for (int r = 0; r < 3; ++r) t[q] += (tile_m == 2 ? G2[a][r] : G4[a][r]) * w[r][q];
      }
      for (int b = 0; b < P; ++b) {
        double u = 0.0;
#pragma unroll
        for (int q = 0; q < 3; ++q) u += t[q] * (tile_m == 2 ? G2[b][q] : G4[b][q]);
        U[(size_t)(a * P + b) * nk + e] = (float)u;
      }
    }
  }
}

extern "C" {

int strotss_conv3x3_winograd_weights(const float* g_nk33, int n, int k, int tile_m, float* u_pnk, void* stream) {
  ST_CHECK_ARG(g_nk33 && u_pnk && n > 0 && k > 0 && (tile_m == 2 || tile_m == 4), STROTSS_EINVAL);
  const long long nk = (long long)n * k;
  hipLaunchKernelGGL(winograd_weights_kernel, dim3((unsigned)min((long long)4096, (nk + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, g_nk33, nk, tile_m, u_pnk);
  ST_LAUNCH_RET();
}

size_t strotss_relu_bits_bytes(int h, int w, int c) {
  return (size_t)((h + 3) / 4) * ((w + 3) / 4) * (size_t)c * sizeof(unsigned int);
}

size_t strotss_conv3x3_winograd_workspace_bytes(int h, int w, int cin, int cout, int tile_m) {
  if (tile_m == 4) {
    const size_t T4 = (size_t)((h + 3) / 4) * ((w + 3) / 4);
    return ws_slice(36 * T4 * cin * 3 / 2, sizeof(float)) + ws_slice(36 * ((T4 + 127) / 128 * 128) * cout, sizeof(float));
  }
  const size_t T = (size_t)((h + 1) / 2) * ((w + 1) / 2);
  return ws_slice(16 * T * cin, sizeof(float)) + ws_slice(16 * T * cout, sizeof(float));
}

int strotss_conv3x3_winograd_fwd(const float* in, int h, int w, int cin, const float* u_pok, const float* u_packed,
                                 const void* u_x3, const float* bias, int cout, int tile_m, float* out, float* pool_out,
                                 unsigned char* pool_code, unsigned int* relu_bits_out, void* workspace,
                                 size_t workspace_bytes, void* stream) {
  ST_CHECK_ARG(in && u_pok && bias && out && workspace && h > 0 && w > 0, STROTSS_EINVAL);
  ST_CHECK_ARG(!relu_bits_out || tile_m == 4, STROTSS_EINVAL);
  ST_CHECK_ARG(cin > 0 && cin % 32 == 0 && cout > 0 && cout % 64 == 0, STROTSS_EALIGN);
  ST_CHECK_ARG(tile_m == 2 || tile_m == 4, STROTSS_EINVAL);
  ST_CHECK_ARG(!pool_out || (h >= 2 && w >= 2), STROTSS_EINVAL);
  ST_CHECK_ARG(!pool_code || pool_out, STROTSS_EINVAL);
  int rc;
  if (tile_m == 4)
    rc = winograd43_run(in, h, w, cin, u_pok, u_packed, u_x3, bias, cout, nullptr, 1, out, pool_out, pool_code, workspace,
                        workspace_bytes, (hipStream_t)stream, nullptr, relu_bits_out);
  else
    rc = winograd_run(in, h, w, cin, u_pok, bias, cout, nullptr, 1, out, workspace, workspace_bytes,
                      (hipStream_t)stream);
  if (rc != 0 || !pool_out || tile_m == 4) return rc;     // the F(4x4) kernels pool in their epilogues
  return st_maxpool2_fwd(out, h, w, cout, pool_out, pool_code, (hipStream_t)stream);
}

int strotss_conv3x3_winograd_dgrad(const float* gout, int h, int w, int cout, const float* u_pik,
                                   const float* u_packed, const void* u_x3, int cin, int tile_m,
                                   const float* act_in, const unsigned int* relu_bits, float* gin, int accumulate,
                                   void* workspace, size_t workspace_bytes, void* stream) {
  ST_CHECK_ARG(gout && u_pik && gin && workspace && h > 0 && w > 0, STROTSS_EINVAL);
  ST_CHECK_ARG(!relu_bits || tile_m == 4, STROTSS_EINVAL);
  ST_CHECK_ARG(!accumulate || (tile_m == 4 && (act_in || relu_bits)), STROTSS_EINVAL);
  ST_CHECK_ARG(cout > 0 && cout % 32 == 0 && cin > 0 && cin % 64 == 0, STROTSS_EALIGN);
  ST_CHECK_ARG(tile_m == 2 || tile_m == 4, STROTSS_EINVAL);
  if (tile_m == 4)
    return winograd43_run(gout, h, w, cout, u_pik, u_packed, u_x3, nullptr, cin, act_in, 0, gin, nullptr, nullptr, workspace,
                          workspace_bytes, (hipStream_t)stream, relu_bits, nullptr, accumulate);
  return winograd_run(gout, h, w, cout, u_pik, nullptr, cin, act_in, 0, gin, workspace, workspace_bytes,
                      (hipStream_t)stream);
}

int strotss_debug_winograd_stages(int mask) {
  const int old = g_wino_stages;
  g_wino_stages = mask & 7;
  return old;
}

int strotss_conv3x3_winograd_route(int h, int w, int cin, int cout, int tile_m, int has_packed, int has_x3) {
  if (h <= 0 || w <= 0 || cin <= 0 || cout <= 0 || (tile_m != 2 && tile_m != 4)) return STROTSS_EINVAL;
  if (tile_m == 2) return STROTSS_ROUTE_F2_GEMM_F32;
  const size_t T = (size_t)((h + 3) / 4) * ((w + 3) / 4);
  const bool prefer_x3 = has_x3 && cin % 32 == 0 && winograd43_prefers_x3(h, w, cout);
  if (!prefer_x3 && has_packed && cin % 32 == 0 && st_winograd43_fused_enabled(h, w, cout)) return STROTSS_ROUTE_F4_FUSED_F32;
  if (!(has_x3 && cin % 32 == 0 && x3_enabled(T, cout))) return STROTSS_ROUTE_F4_GEMM_F32;
  return (long)((T + 127) / 128) * ((cout + 127) / 128) * 36 >= x3_min_tiles() ? STROTSS_ROUTE_F4_X3_GEMM_128
                                                                             : STROTSS_ROUTE_F4_X3_GEMM_64;
}

size_t strotss_conv3x3_winograd_x3_bytes(int rows, int k) { return (size_t)36 * 3 * rows * k * sizeof(unsigned short); }

int strotss_conv3x3_winograd_x3pack(const float* u_prk, int rows, int k, void* u_x3, void* stream) {
  ST_CHECK_ARG(u_prk && u_x3 && rows > 0 && k > 0, STROTSS_EINVAL);
  ST_CHECK_ARG(k % 32 == 0, STROTSS_EALIGN);
  return st_x3_split_rows(u_prk, rows, k, k, (long long)rows * k, u_x3, 36, (hipStream_t)stream);
}

int strotss_conv3x3_winograd_pack(const float* u_prk, int rows, int k, float* u_packed, void* stream) {
  ST_CHECK_ARG(u_prk && u_packed && rows > 0 && k > 0, STROTSS_EINVAL);
  ST_CHECK_ARG(rows % 32 == 0 && k % 8 == 0, STROTSS_EALIGN);
  return st_winograd43_pack(u_prk, rows, k, u_packed, (hipStream_t)stream);
}

}  // extern "C"
