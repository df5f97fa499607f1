// Winograd F(4x4, 3x3) in ONE kernel for the 64-output-channel layers (block1_conv2 forward and data-gradient,
// data-gradient of block2_conv1; nn/model.py:44-48 of the reference).
//
// The three-kernel form (winograd.hip) moves V = B^T d B and M = V U^T -- each 2.25x the activation -- through
// HBM: 2.4 GB per launch at 1024^2 x 64 for 0.54 GB of algorithmic traffic, and with Cin = 64 the 36 GEMMs have
// only 16 flop per byte.  Here a workgroup owns a 4x8 block of Winograd tiles (16x32 output pixels) and 32 of
// the 64 output channels, and keeps everything on chip:
//
//   for every chunk of 16 input channels:
//     U    : this wave's 3 positions x (32 couts x 16 ch), straight from L2 into registers (issued first, so the
//            in-order vmcnt never makes an MFMA wait on the HBM loads behind them)
//     V    : 36 x (32 tiles x 16 ch) = B^T d B of the 18x34 input patch           LDS -> LDS   (49 KB -> 92 KB)
//     raw  : the next chunk's patch, global -> registers during the MFMA phase -> LDS
//     M   += V[p] U[p]^T : 12 waves x 3 positions x 8 v_mfma_f32_32x32x2_f32 (32 tiles x 32 couts per position)
//   Y = A^T M A : accumulators -> LDS (147 KB) -> bias/ReLU or ReLU mask -> global
//
// HBM traffic = input (x1.2 halo; the second cout half hits L2) + output.  One workgroup of 768 threads per CU
// (LDS), 48 accumulator registers per lane, 3 waves per SIMD.
#include <stdlib.h>

#include "internal.h"
#include "mfma_pipe.h"

namespace {

constexpr int F_TR = 4, F_TC = 8;                  // Winograd tiles per workgroup: rows x cols
constexpr int F_TILES = F_TR * F_TC;               // 32 = M of the MFMA tile
constexpr int F_PH = 4 * F_TR + 2, F_PW = 4 * F_TC + 2;   // input patch 18 x 34
constexpr int F_KC = 16;                           // input channels per chunk
constexpr int F_PS = 20;                           // LDS floats per patch pixel / per V row (80 B: conflict-free b128)
constexpr int F_RAW = F_PH * F_PW * F_PS;          // 12240 floats
constexpr int F_V = 36 * F_TILES * F_PS;           // 23040 floats
constexpr int F_MX = 36 * F_TILES * 32;            // 36864 floats (epilogue exchange, one 32-cout half)
constexpr int F_LDS_FLOATS = (F_RAW + F_V + 4 > F_MX) ? F_RAW + F_V + 4 : F_MX;   // + dummy slot of store_raw
constexpr int F_NT = 768;                          // threads: 12 waves, 3 positions each
constexpr int F_NLOAD = (F_PH * F_PW * 4 + F_NT - 1) / F_NT;   // float4 loads per thread per chunk (4)

__device__ __forceinline__ void bt6s(float (&d)[6]) {      // in-place B^T d (Lavin & Gray, points 0, +-1, +-2, inf)
  const float d0 = d[0], d1 = d[1], d2 = d[2], d3 = d[3], d4 = d[4], d5 = d[5];
  d[0] = 4.f * d0 - 5.f * d2 + d4;
  d[1] = -4.f * d1 - 4.f * d2 + d3 + d4;
  d[2] = 4.f * d1 - 4.f * d2 - d3 + d4;
  d[3] = -2.f * d1 - d2 + 2.f * d3 + d4;
  d[4] = 2.f * d1 - d2 - 2.f * d3 + d4;
  d[5] = 4.f * d1 - 5.f * d3 + d5;
}

// in: (H, W, K) NHWC; U: (36, Cout, K); out / mask: (H, W, Cout), Cout % 32 == 0.
// !MASK: out = relu ? max(Y + bias, 0) : Y + bias;  MASK: out = mask > 0 ? Y : 0.
// Work item = (region r of 16x32 output pixels, row-major) x (group g of 32 couts): item = r * NG + g.  The grid is
// persistent (one workgroup per CU); workgroup b walks the items of its XCD's contiguous range, so the NG groups of
// a region (same input patch) run on neighbouring CUs of one XCD at the same time and share its L2.
template <bool MASK>
__global__ __launch_bounds__(F_NT) void winograd43_fused_kernel(const float* __restrict__ in, int H, int W, int K,
                                                                const float* __restrict__ U, int Cout,
                                                                const float* __restrict__ bias,
                                                                const float* __restrict__ mask, int relu,
                                                                float* __restrict__ out, int RW, int NG, int nitems) {
  __shared__ __attribute__((aligned(16))) float lds[F_LDS_FLOATS];
  float* raw = lds;
  float* Vs = lds + F_RAW;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int l31 = lane & 31, hh = lane >> 5;

  // items of this workgroup: XCD x = blockIdx % 8 owns [x * per, (x + 1) * per), its workgroups interleave them
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslot = gridDim.x >> 3;
  const int per = (nitems + 7) >> 3;
  const int item_end = min(nitems, (xcd + 1) * per);
  int item = xcd * per + slot;
  if (item >= item_end) return;                    // whole workgroup

  // ---- raw patch loader of one item: element e = t + 768 i  ->  pixel e >> 2, channel quad e & 3
  int goff[F_NLOAD];
  unsigned okm = 0;
  auto setup = [&](int it, int (&go)[F_NLOAD], unsigned& ok_mask) {
    const int region = it / NG;
    const int ry = region / RW, rx = region - ry * RW;
    ok_mask = 0;
#pragma unroll
    for (int i = 0; i < F_NLOAD; ++i) {
      const int e = t + F_NT * i;
      const int px = e >> 2, c4 = e & 3;
      const int py = px / F_PW, pxx = px - py * F_PW;
      const int gy = ry * (4 * F_TR) - 1 + py, gx = rx * (4 * F_TC) - 1 + pxx;
      const bool ok = (e < F_PH * F_PW * 4) && gy >= 0 && gy < H && gx >= 0 && gx < W;
      ok_mask |= (unsigned)ok << i;
      go[i] = ok ? (gy * W + gx) * K + c4 * 4 : 0;     // out-of-image: load pixel 0, zeroed at the LDS store
    }
  };
  f32x4 stage[F_NLOAD];
  auto load_raw = [&](const int (&go)[F_NLOAD], int kc) {
#pragma unroll
    for (int i = 0; i < F_NLOAD; ++i) stage[i] = *reinterpret_cast<const f32x4*>(in + go[i] + kc * F_KC);
  };
  auto store_raw = [&](unsigned ok_mask) {         // branch-free: surplus threads write a dummy slot behind V
#pragma unroll
    for (int i = 0; i < F_NLOAD; ++i) {
      const int e = t + F_NT * i;
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      const int o = (e < F_PH * F_PW * 4) ? (e >> 2) * F_PS + (e & 3) * 4 : F_RAW + F_V;
      *reinterpret_cast<f32x4*>(raw + o) = ((ok_mask >> i) & 1u) ? stage[i] : z;
    }
  };
  // ---- input transform: thread (< 512) = (tile t >> 4, channel t & 15)
  auto transform = [&]() {
    const int ttile = t >> 4, tch = t & 15;
    const int tty = ttile >> 3, ttx = ttile & 7;
    float d[6][6];
    const float* src = raw + ((4 * tty) * F_PW + 4 * ttx) * F_PS + tch;
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
      for (int q = 0; q < 6; ++q) d[r][q] = src[(r * F_PW + q) * F_PS];
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      float col[6] = {d[0][q], d[1][q], d[2][q], d[3][q], d[4][q], d[5][q]};
      bt6s(col);
#pragma unroll
      for (int r = 0; r < 6; ++r) d[r][q] = col[r];
    }
    float* dst = Vs + ttile * F_PS + tch;
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      bt6s(d[r]);
#pragma unroll
      for (int q = 0; q < 6; ++q) dst[(r * 6 + q) * (F_TILES * F_PS)] = d[r][q];
    }
  };
  // U fragments of (item, chunk): positions wave, wave + 12, wave + 24; B = U[p][32 g + l31][16 kc + 8 hh ..]
  const size_t b_step = (size_t)12 * Cout * K;
  auto load_u = [&](int it, int kc, f32x4 (&b)[3][2]) {
    const int g = it - (it / NG) * NG;
    const float* bp = U + ((size_t)wave * Cout + g * 32 + l31) * K + 8 * hh + kc * F_KC;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      b[j][0] = *reinterpret_cast<const f32x4*>(bp + j * b_step);
      b[j][1] = *reinterpret_cast<const f32x4*>(bp + j * b_step + 4);
    }
  };
  const float* a_base = Vs + (wave * F_TILES + l31) * F_PS + 8 * hh;      // A = V[p][tile l31][8 hh ..]
  constexpr int a_step = 12 * F_TILES * F_PS;
  const int nchunk = K / F_KC;

  f32x4 b[3][2];
  setup(item, goff, okm);
  load_raw(goff, 0);
  store_raw(okm);
  __syncthreads();

  for (;;) {
    f32x16 acc[3];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const int next = item + nslot;
    const bool more = next < item_end;              // uniform

    for (int kc = 0; kc < nchunk; ++kc) {
      // This chunk's U fragments (they land during the transform), then -- after the transform, whose registers
      // they would otherwise compete with -- the input patch of the NEXT chunk (of the next item after the last
      // chunk; the very last chunk of the workgroup re-loads itself).  vmcnt retires in order, so the MFMAs wait
      // for the older U loads only and the patch has the MFMA phase to land.  Loads and LDS stores are
      // unconditional: behind a branch the compiler sinks them below the MFMAs.
      const bool last = kc + 1 == nchunk;
      load_u(item, kc, b);
      __builtin_amdgcn_sched_barrier(0);
      if (t < 512) transform();
      if (last && more) setup(next, goff, okm);     // this item's patches are all loaded
      load_raw(goff, last ? (more ? 0 : kc) : kc + 1);
      __syncthreads();                              // V complete; raw consumed
      __builtin_amdgcn_sched_barrier(0);
      f32x4 a[3][2];
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        a[j][0] = *reinterpret_cast<const f32x4*>(a_base + j * a_step);
        a[j][1] = *reinterpret_cast<const f32x4*>(a_base + j * a_step + 4);
      }
#pragma unroll
      for (int j = 0; j < 3; ++j) {
#pragma unroll
        for (int s = 0; s < 4; ++s) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j][0][s], b[j][0][s], acc[j], 0, 0, 0);
#pragma unroll
        for (int s = 0; s < 4; ++s) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j][1][s], b[j][1][s], acc[j], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);            // keep the patch's zero-selects (which wait for it) behind the MFMAs
      if (!last) store_raw(okm);                    // after the last chunk the patch buffer becomes the exchange area
      __syncthreads();                              // all waves done with V; next raw complete
    }

    // ---- output transform through LDS: Mx[p][tile][32 couts]
    const int region = item / NG, g = item - region * NG;
    const int ry = region / RW, rx = region - ry * RW;
    const int y0 = ry * (4 * F_TR), x0 = rx * (4 * F_TC);
    float* Mx = lds;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      float* dst = Mx + ((wave + 12 * j) * F_TILES + 4 * hh) * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) dst[((r & 3) + 8 * (r >> 2)) * 32] = acc[j][r];
    }
    __syncthreads();
#pragma unroll 1
    for (int e = t; e < F_TILES * 32; e += F_NT) {
      const int cl = e & 31, tile = e >> 5;
      const int ty = tile >> 3, tx = tile & 7;
      const int co = g * 32 + cl;
      const float* src = Mx + tile * 32 + cl;
      float s[4][6];
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        float m[6];
#pragma unroll
        for (int r = 0; r < 6; ++r) m[r] = src[(r * 6 + q) * (F_TILES * 32)];
        s[0][q] = m[0] + m[1] + m[2] + m[3] + m[4];
        s[1][q] = m[1] - m[2] + 2.f * m[3] - 2.f * m[4];
        s[2][q] = m[1] + m[2] + 4.f * m[3] + 4.f * m[4];
        s[3][q] = m[1] - m[2] + 8.f * m[3] - 8.f * m[4] + m[5];
      }
      float bv = 0.f;
      if constexpr (!MASK) { if (bias) bv = bias[co]; }   // the plain data-gradient has neither bias nor mask
      const int yb = y0 + 4 * ty, xb = x0 + 4 * tx;
      // branch-free: rows / columns beyond the image are clamped for the (batched) mask loads and predicated
      // at the stores, so no load or store waits on another
      const bool tile_in = yb < H && xb < W;
      const size_t ob = tile_in ? ((size_t)yb * W + xb) * Cout + co : (size_t)co;
      int off[4][4];
      bool ok[4][4];
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          ok[r][q] = tile_in && yb + r < H && xb + q < W;
          off[r][q] = ok[r][q] ? (r * W + q) * Cout : 0;
        }
      float* op = out + ob;
      float mk[4][4];
      if constexpr (MASK) {
        const float* mp = mask + ob;
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int q = 0; q < 4; ++q) mk[r][q] = mp[off[r][q]];
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float yv[4];
        yv[0] = s[r][0] + s[r][1] + s[r][2] + s[r][3] + s[r][4] + bv;
        yv[1] = s[r][1] - s[r][2] + 2.f * s[r][3] - 2.f * s[r][4] + bv;
        yv[2] = s[r][1] + s[r][2] + 4.f * s[r][3] + 4.f * s[r][4] + bv;
        yv[3] = s[r][1] - s[r][2] + 8.f * s[r][3] - 8.f * s[r][4] + s[r][5] + bv;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float v = yv[q];
          if constexpr (MASK) v = mk[r][q] > 0.f ? v : 0.f;
          else if (relu) v = fmaxf(v, 0.f);
          if (ok[r][q]) op[off[r][q]] = v;
        }
      }
    }
    if (!more) break;
    __syncthreads();                                // exchange area read; it becomes the patch buffer again
    item = next;
    store_raw(okm);                                 // the next item's first patch, loaded during the last chunk
    __syncthreads();
  }
}

}  // namespace

// Policy (measured per layer, tools/conv_bench.py): the fused kernel wins where the three-kernel form is bound by
// its transform traffic -- up to 128 output channels -- and there are at least two work items per CU.
// STROTSS_WINO_FUSED: 0 = never, 1 (default) = that policy, 2 = every layer it supports.
bool st_winograd43_fused_enabled(int h, int w, int cout) {
  static int on = -1;
  if (on < 0) { const char* e = getenv("STROTSS_WINO_FUSED"); on = e ? atoi(e) : 1; }
  if (on == 0 || cout % 32 != 0) return false;
  if (on >= 2) return true;
  const int TH = (h + 3) / 4, TW = (w + 3) / 4;
  const long items = (long)((TH + F_TR - 1) / F_TR) * ((TW + F_TC - 1) / F_TC) * (cout / 32);
  return cout <= 128 && items >= 512;
}

int st_winograd43_fused(const float* in, int h, int w, int cin, const float* U, const float* bias, int cout,
                        const float* mask, int relu, float* out, hipStream_t st) {
  if (cin % F_KC != 0 || cout % 32 != 0) return STROTSS_EALIGN;
  if ((size_t)h * w * cin >= ((size_t)1 << 31) || (size_t)h * w * cout >= ((size_t)1 << 31)) return STROTSS_EALIGN;
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
  if (mask)
    hipLaunchKernelGGL(winograd43_fused_kernel<true>, dim3((unsigned)grid), dim3(F_NT), 0, st, in, h, w, cin, U, cout,
                       bias, mask, relu, out, RW, NG, nitems);
  else
    hipLaunchKernelGGL(winograd43_fused_kernel<false>, dim3((unsigned)grid), dim3(F_NT), 0, st, in, h, w, cin, U, cout,
                       bias, mask, relu, out, RW, NG, nitems);
  ST_LAUNCH_RET();
}
