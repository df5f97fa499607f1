// VGG16 trunk kernels (nn/model.py:44-55): 3x3 SAME conv + bias + ReLU as an implicit GEMM on the
// fp32 MFMA tile engine (forward and data-gradient; the net is frozen so there is no weight
// gradient), the 3-channel first layer and its pixel gradient, and 2x2 max-pool fwd/bwd.
// Layout NHWC: a pixel's channels are contiguous, so the GEMM's A operand (pixels x Cin-chunk)
// is gathered as whole 128-byte channel segments and K = (tap, ci) needs no im2col buffer.
#include <stdlib.h>

#include "internal.h"
#include "mfma_tile.h"

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

// v2: double-buffered LDS, ONE barrier per K-step.  Tile s+1 (global loads issued a whole K-step
// earlier) is written into the other LDS buffer before tile s's MFMAs; the loads of tile s+2 are
// issued next and fly under those MFMAs.  (tap, ci) advance incrementally: no division per step.
template <int BM, int BN>
__global__ __launch_bounds__(256) void conv3x3_mfma_db_kernel(const float* __restrict__ in, int H, int W,
                                                              int Cin, const float* __restrict__ wt,
                                                              const float* __restrict__ bias, int Cout,
                                                              const float* __restrict__ mask,
                                                              float* __restrict__ out, int relu) {
  constexpr int A_FL = OperandLds<BM>::kc_floats, B_FL = OperandLds<BN>::kc_floats;
  __shared__ __attribute__((aligned(16))) float lds[2 * (A_FL + B_FL)];
  const int HW = H * W;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int t = threadIdx.x, c4 = t & 7, r0 = t >> 3;

  int py[BM / 32], px[BM / 32];
#pragma unroll
  for (int i = 0; i < BM / 32; ++i) {
    const int p = m0 + r0 + 32 * i;
    if (p < HW) { py[i] = p / W; px[i] = p - py[i] * W; }
    else { py[i] = -4; px[i] = -4; }
  }
  const int steps = 9 * (Cin >> 5);
  int tap = 0, ci0 = 0;      // state of the NEXT load
  f32x4 ra[BM / 32], rb[BN / 32];
  auto load = [&]() {
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
      const int co = n0 + r0 + 32 * i;
      rb[i] = *reinterpret_cast<const f32x4*>(&wt[((size_t)tap * Cout + co) * Cin + ci0 + c4 * 4]);
    }
    ci0 += 32;
    if (ci0 == Cin) { ci0 = 0; ++tap; }
  };

  f32x16 acc[BM / 64][BN / 64];
  acc_zero<BM, BN>(acc);
  load();
  lds_store_kc<BM>(lds, ra);
  lds_store_kc<BN>(lds + A_FL, rb);
  if (steps > 1) load();
  __syncthreads();
  for (int s = 0; s < steps; ++s) {
    float* cur = lds + (s & 1) * (A_FL + B_FL);
    float* nxt = lds + ((s + 1) & 1) * (A_FL + B_FL);
    if (s + 1 < steps) {
      lds_store_kc<BM>(nxt, ra);
      lds_store_kc<BN>(nxt + A_FL, rb);
    }
    if (s + 2 < steps) load();
    mma_kstep<BM, BN, true, true>(cur, cur + A_FL, acc);
    __syncthreads();
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

// v3: v2 + (a) branch-free staging loads (clamped address now, select-to-zero deferred to the LDS
// store one K-step later, so no load is waited for at issue; no end-of-loop branches: the last two
// iterations reload / rewrite harmlessly), (b) register double-buffered MFMA fragments, (c) the
// staging work of a K-step (LDS writes of tile s+1, global loads of tile s+2) is cut into 4 pieces
// pinned between the MFMA groups with sched_barrier, so it executes in the shadow of the 64-cycle
// f32 MFMAs instead of in front of them; (d) the epilogue batches its mask loads.
template <int BM, int BN>
__global__ __launch_bounds__(256) void conv3x3_mfma_v3_kernel(const float* __restrict__ in, int H, int W,
                                                              int Cin, const float* __restrict__ wt,
                                                              const float* __restrict__ bias, int Cout,
                                                              const float* __restrict__ mask,
                                                              float* __restrict__ out, int relu) {
  constexpr int A_FL = OperandLds<BM>::kc_floats, B_FL = OperandLds<BN>::kc_floats;
  constexpr int TM = BM / 64, TN = BN / 64, NA = BM / 32, NB = BN / 32;
  __shared__ __attribute__((aligned(16))) float lds[2 * (A_FL + B_FL)];
  const int HW = H * W;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int t = threadIdx.x, c4 = t & 7, r0 = t >> 3;
  const int lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, hh = lane >> 5;

  // per staged row: element offset of its own pixel (clamped) and the taps that are in bounds
  int abase[NA];
  unsigned amask[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int p = m0 + r0 + 32 * i;
    unsigned mk = 0;
    int pc = 0;
    if (p < HW) {
      pc = p;
      const int y = p / W, x = p - y * W;
#pragma unroll
      for (int tp = 0; tp < 9; ++tp) {
        const int yy = y + tp / 3 - 1, xx = x + tp % 3 - 1;
        if (yy >= 0 && yy < H && xx >= 0 && xx < W) mk |= 1u << tp;
      }
    }
    abase[i] = pc * Cin + c4 * 4;
    amask[i] = mk;
  }
  int bbase[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) bbase[i] = (n0 + r0 + 32 * i) * Cin + c4 * 4;

  const int steps = 9 * (Cin >> 5);
  int tap = 0, ci0 = 0;            // (tap, ci) of the NEXT tile to load
  int aoff = (-W - 1) * Cin;       // uniform element offset of that tile's tap: (dy*W + dx)*Cin + ci0
  int boff = 0;                    // tap*Cout*Cin + ci0
  f32x4 ra[NA], rb[NB];
  unsigned okbits = 0;             // bit i: ra[i] (the tile held in registers) is an in-bounds tap

  auto load_a = [&](int i) {
    const bool ok = (amask[i] >> tap) & 1u;
    ra[i] = *reinterpret_cast<const f32x4*>(in + (ok ? abase[i] + aoff : abase[i]));
    okbits = (okbits & ~(1u << i)) | ((unsigned)ok << i);
  };
  auto load_b = [&](int i) { rb[i] = *reinterpret_cast<const f32x4*>(wt + bbase[i] + boff); };
  auto advance = [&]() {
    ci0 += 32; aoff += 32; boff += 32;
    if (ci0 == Cin) {
      ci0 = 0; ++tap;
      if (tap == 9) tap = 0;     // past the end: wrap, so the (unused) extra loads stay in bounds
      const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
      aoff = (dy * W + dx) * Cin;
      boff = tap * Cout * Cin;
    }
  };
  auto store_a = [&](float* buf, int i) {
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    *reinterpret_cast<f32x4*>(&buf[(r0 + 32 * i) * KC_LD + c4 * 4]) = ((okbits >> i) & 1u) ? ra[i] : z;
  };
  auto store_b = [&](float* buf, int i) {
    *reinterpret_cast<f32x4*>(&buf[A_FL + (r0 + 32 * i) * KC_LD + c4 * 4]) = rb[i];
  };

  f32x16 acc[TM][TN];
  acc_zero<BM, BN>(acc);
  // prologue: tile 0 -> buffer 0, tile 1 -> registers
#pragma unroll
  for (int i = 0; i < NA; ++i) load_a(i);
#pragma unroll
  for (int i = 0; i < NB; ++i) load_b(i);
  advance();
#pragma unroll
  for (int i = 0; i < NA; ++i) store_a(lds, i);
#pragma unroll
  for (int i = 0; i < NB; ++i) store_b(lds, i);
#pragma unroll
  for (int i = 0; i < NA; ++i) load_a(i);
#pragma unroll
  for (int i = 0; i < NB; ++i) load_b(i);
  advance();
  __syncthreads();

  const int arow = (wm * (BM / 2) + l31) * KC_LD + 4 * hh;
  const int brow = A_FL + (wn * (BN / 2) + l31) * KC_LD + 4 * hh;
  // fragments of k-group 0 of the first tile; afterwards they are fetched right after the
  // (early) barrier of the previous K-step, under its last MFMAs
  f32x4 fa[2][TM], fb[2][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) fa[0][i] = *reinterpret_cast<const f32x4*>(&lds[arow + i * 32 * KC_LD]);
#pragma unroll
  for (int i = 0; i < TN; ++i) fb[0][i] = *reinterpret_cast<const f32x4*>(&lds[brow + i * 32 * KC_LD]);
  for (int s = 0; s < steps; ++s) {
    const float* cur = lds + (s & 1) * (A_FL + B_FL);
    float* nxt = lds + ((s + 1) & 1) * (A_FL + B_FL);
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
          // staging piece g: rows {g, g+4, ...} of A -> LDS (tile s+1), then reload them (tile s+2)
#pragma unroll
          for (int i = g; i < NA; i += 4) { store_a(nxt, i); load_a(i); }
          __builtin_amdgcn_sched_barrier(0);
        }
        if (j == 1) {
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int i = g; i < NB; i += 4) { store_b(nxt, i); load_b(i); }
          __builtin_amdgcn_sched_barrier(0);
          if (g == 3) {
            // Every wave has issued all its reads of `cur` (k-group 3's fragments were read during
            // group 2) and all its writes of `nxt`: one barrier here orders both, and the first
            // fragments of the next tile are fetched under the 8 MFMAs that remain.
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
    advance();
  }

  AccMap<BM, BN> map;
#pragma unroll
  for (int in_ = 0; in_ < TN; ++in_) {
    const int co = n0 + map.colof(in_);
    const float b = bias ? bias[co] : 0.f;
#pragma unroll
    for (int im = 0; im < TM; ++im) {
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
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int p = m0 + map.row(im, reg);
        if (p < HW) {
          float v = acc[im][in_][reg] + b;
          if (relu) v = fmaxf(v, 0.f);
          v = (mv[reg] > 0.f) ? v : 0.f;
          out[(size_t)p * Cout + co] = v;
        }
      }
    }
  }
}

static int conv_variant() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("STROTSS_CONV_VARIANT");
    v = e ? atoi(e) : 3;
  }
  return v;
}

template <int BM, int BN>
int launch_conv(const float* in, int H, int W, int Cin, const float* wt, const float* bias, int Cout,
                const float* mask, float* out, int relu, hipStream_t s) {
  dim3 grid(Cout / BN, cdiv((int64_t)H * W, BM));
  if (conv_variant() == 1)
    hipLaunchKernelGGL((conv3x3_mfma_kernel<BM, BN>), grid, dim3(256), 0, s, in, H, W, Cin, wt, bias, Cout,
                       mask, out, relu);
  else if (conv_variant() == 2)
    hipLaunchKernelGGL((conv3x3_mfma_db_kernel<BM, BN>), grid, dim3(256), 0, s, in, H, W, Cin, wt, bias, Cout,
                       mask, out, relu);
  else
    hipLaunchKernelGGL((conv3x3_mfma_v3_kernel<BM, BN>), grid, dim3(256), 0, s, in, H, W, Cin, wt, bias, Cout,
                       mask, out, relu);
  ST_LAUNCH_RET();
}

int conv_dispatch(const float* in, int H, int W, int Cin, const float* wt, const float* bias, int Cout,
                  const float* mask, float* out, int relu, hipStream_t s) {
  const int64_t M = (int64_t)H * W;
  if (Cout % 128 == 0 && cdiv(M, 128) * (Cout / 128) >= 512)
    return launch_conv<128, 128>(in, H, W, Cin, wt, bias, Cout, mask, out, relu, s);
  if (cdiv(M, 128) * (Cout / 64) >= 512)
    return launch_conv<128, 64>(in, H, W, Cin, wt, bias, Cout, mask, out, relu, s);
  return launch_conv<64, 64>(in, H, W, Cin, wt, bias, Cout, mask, out, relu, s);
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

// ---------------------------------------------------------------- 2x2/2 VALID max-pool
__global__ __launch_bounds__(256) void maxpool2_fwd_kernel(const float* __restrict__ in, int H, int W, int C4,
                                                           float* __restrict__ out) {
  const int Ho = H >> 1, Wo = W >> 1;
  const size_t total = (size_t)Ho * Wo * C4;
  const f32x4* src = reinterpret_cast<const f32x4*>(in);
  f32x4* dst = reinterpret_cast<f32x4*>(out);
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const int c = (int)(e % C4);
    const size_t pix = e / C4;
    const int ox = (int)(pix % Wo), oy = (int)(pix / Wo);
    const size_t b = ((size_t)(2 * oy) * W + 2 * ox) * C4 + c;
    const f32x4 v00 = src[b], v01 = src[b + C4], v10 = src[b + (size_t)W * C4], v11 = src[b + (size_t)W * C4 + C4];
    f32x4 m;
#pragma unroll
    for (int k = 0; k < 4; ++k) m[k] = fmaxf(fmaxf(v00[k], v01[k]), fmaxf(v10[k], v11[k]));
    dst[e] = m;
  }
}
// gin[y,x,c] = (this pixel is the FIRST max of its window, scan order (0,0),(0,1),(1,0),(1,1))
//              ? gout[y/2,x/2,c] : 0, times (act > 0); pixels outside the pooled area get 0.
__global__ __launch_bounds__(256) void maxpool2_bwd_kernel(const float* __restrict__ act, int H, int W, int C4,
                                                           const float* __restrict__ gout,
                                                           float* __restrict__ gin) {
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
    dst[e] = r;
  }
}

}  // namespace

extern "C" {

int strotss_conv3x3_c3_fwd(const float* img, int h, int w, const float* w_kio, const float* bias, int cout,
                           const float* mean3, const float* std3, float* out, void* stream) {
  ST_CHECK_ARG(img && w_kio && bias && mean3 && std3 && out && h > 0 && w > 0, STROTSS_EINVAL);
  ST_CHECK_ARG(cout >= 16 && cout <= 256 && (cout & (cout - 1)) == 0, STROTSS_EALIGN);
  const f32x4 m = {mean3[0], mean3[1], mean3[2], 0.f};
  const f32x4 is = {1.0f / std3[0], 1.0f / std3[1], 1.0f / std3[2], 0.f};
  const int ppb = 256 / (cout >> 4);
  hipLaunchKernelGGL(conv3x3_c3_fwd_kernel, dim3(cdiv((int64_t)h * w, ppb)), dim3(256),
                     27 * cout * sizeof(float), (hipStream_t)stream, img, h, w, w_kio, bias, cout, m, is, out);
  ST_LAUNCH_RET();
}

int strotss_conv3x3_relu_fwd(const float* in, int h, int w, int cin, const float* w_tok, const float* bias,
                             int cout, float* out, void* stream) {
  ST_CHECK_ARG(in && w_tok && bias && out && h > 0 && w > 0, STROTSS_EINVAL);
  ST_CHECK_ARG(cin > 0 && cin % 32 == 0 && cout > 0 && cout % 64 == 0, STROTSS_EALIGN);
  return conv_dispatch(in, h, w, cin, w_tok, bias, cout, nullptr, out, 1, (hipStream_t)stream);
}

int strotss_conv3x3_dgrad(const float* gout, int h, int w, int cout, const float* w_tik, int cin,
                          const float* act_in, float* gin, void* stream) {
  ST_CHECK_ARG(gout && w_tik && gin && h > 0 && w > 0, STROTSS_EINVAL);
  ST_CHECK_ARG(cout > 0 && cout % 32 == 0 && cin > 0 && cin % 64 == 0, STROTSS_EALIGN);
  // the transposed convolution is a convolution with K = cout and N = cin
  return conv_dispatch(gout, h, w, cout, w_tik, nullptr, cin, act_in, gin, 0, (hipStream_t)stream);
}

int strotss_conv3x3_c3_dgrad(const float* gout, int h, int w, int cout, const float* w_tic,
                             const float* std3, float* gimg, int accumulate, void* stream) {
  ST_CHECK_ARG(gout && w_tic && std3 && gimg && h > 0 && w > 0, STROTSS_EINVAL);
  ST_CHECK_ARG(cout >= 16 && cout <= 256 && (cout & (cout - 1)) == 0, STROTSS_EALIGN);
  const f32x4 is = {1.0f / std3[0], 1.0f / std3[1], 1.0f / std3[2], 0.f};
  const int ppb = 256 / (cout >> 4);
  hipLaunchKernelGGL(conv3x3_c3_dgrad_kernel, dim3(cdiv((int64_t)h * w, ppb)), dim3(256),
                     27 * cout * sizeof(float), (hipStream_t)stream, gout, h, w, cout, w_tic, is, gimg,
                     accumulate);
  ST_LAUNCH_RET();
}

int strotss_maxpool2_fwd(const float* in, int h, int w, int c, float* out, void* stream) {
  ST_CHECK_ARG(in && out && h >= 2 && w >= 2 && c > 0, STROTSS_EINVAL);
  ST_CHECK_ARG(c % 4 == 0, STROTSS_EALIGN);
  const size_t total = (size_t)(h / 2) * (w / 2) * (c / 4);
  hipLaunchKernelGGL(maxpool2_fwd_kernel, dim3(min((size_t)8192, (total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, in, h, w, c / 4, out);
  ST_LAUNCH_RET();
}

int strotss_maxpool2_bwd(const float* act, int h, int w, int c, const float* gout, float* gin, void* stream) {
  ST_CHECK_ARG(act && gout && gin && h >= 2 && w >= 2 && c > 0, STROTSS_EINVAL);
  ST_CHECK_ARG(c % 4 == 0, STROTSS_EALIGN);
  const size_t total = (size_t)h * w * (c / 4);
  hipLaunchKernelGGL(maxpool2_bwd_kernel, dim3(min((size_t)8192, (total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, act, h, w, c / 4, gout, gin);
  ST_LAUNCH_RET();
}

}  // extern "C"
