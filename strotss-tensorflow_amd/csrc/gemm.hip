// fp32-MFMA GEMMs of the loss path (cost matrices, covariance, their backward products).
// All are C[m][n] = sum_k A(m,k) B(n,k) on the tile engine of mfma_tile.h with fused epilogues.
#include <stdlib.h>

#include "internal.h"
#include "mfma_pipe.h"
#include "mfma_x3.h"

namespace {

template <int BM, int BN, bool AKC, bool BKC>
__device__ __forceinline__ void gemm_mainloop(const float* __restrict__ A, int lda, int M,
                                              const float* __restrict__ B, int ldb, int N, int K,
                                              int m0, int n0, float* ldsA, float* ldsB,
                                              f32x16 (&acc)[BM / 64][BN / 64]) {
  f32x4 ra[BM / 32], rb[BN / 32];
  auto load = [&](int k0) {
    if constexpr (AKC) gload_kc<BM>(ra, A, lda, m0, M, k0); else gload_rc<BM>(ra, A, lda, m0, M, k0);
    if constexpr (BKC) gload_kc<BN>(rb, B, ldb, n0, N, k0); else gload_rc<BN>(rb, B, ldb, n0, N, k0);
  };
  load(0);
  for (int k0 = 0; k0 < K; k0 += 32) {
    __syncthreads();  // everyone is done reading the previous tile
    if constexpr (AKC) lds_store_kc<BM>(ldsA, ra); else lds_store_rc<BM>(ldsA, ra);
    if constexpr (BKC) lds_store_kc<BN>(ldsB, rb); else lds_store_rc<BN>(ldsB, rb);
    __syncthreads();
    if (k0 + 32 < K) load(k0 + 32);  // in flight under the MFMAs below
    mma_kstep<BM, BN, AKC, BKC>(ldsA, ldsB, acc);
  }
}

// ---------------------------------------------------------------- epilogues
struct EpiCosDist {  // C = 1 - acc * (ra[i]*rb[j])      (losses.py:12-15)
  const float* ra; const float* rb; float* C; int ldc; int M, N;
  int symm;            // x == y: only tiles on or above the diagonal are computed, the rest is mirrored (bitwise equal)
  static constexpr bool SYMM = true;
  __device__ __forceinline__ void set_batch(int) {}
  __device__ __forceinline__ float value(int r, int c, float v) const { return 1.0f - v * (ra[r] * rb[c]); }
  __device__ __forceinline__ float apply(int r, int c, float v) const {
    if (r < M && c < N) C[(size_t)r * ldc + c] = value(r, c, v);
    return 0.f;
  }
  __device__ __forceinline__ void mirror(int r, int c, float val) const { C[(size_t)r * ldc + c] = val; }
  __device__ __forceinline__ void finish(float*, float) const {}
};
// The same on the bf16x3 core.  There the partial products of (i,j) and (j,i) are accumulated in different orders, so
// symmetry is made exact by construction: only entries on or above the diagonal are stored by their owner, every
// entry below is the mirrored copy of its transpose (diagonal tiles mirror their own strict upper triangle).
struct EpiCosDistX3 : EpiCosDist, X3NoPrefetch<EpiCosDistX3> {
  using X3NoPrefetch<EpiCosDistX3>::apply;
  long long rstride, cstride;               // batch z: norms at ra/rb + z * rstride, matrix at C + z * cstride
  __device__ __forceinline__ void set_batch(int z) { ra += z * rstride; rb += z * rstride; C += z * cstride; }
  __device__ __forceinline__ float value(int r, int c, float v) const {     // unconditional loads (clamped addresses)
    return 1.0f - v * (ra[min(r, M - 1)] * rb[min(c, N - 1)]);
  }
  __device__ __forceinline__ float apply(int r, int c, float v) const {
    const float d = value(r, c, v);
    if (r < M && c < N && !(symm && r > c)) C[(size_t)r * ldc + c] = d;
    return 0.f;
  }
};
struct EpiL2Dist {  // C = sqrt(max(xs[i] + ys[j] - 2 acc, 1e-6) / D)      (losses.py:18-24)
  static constexpr bool SYMM = false;
  const float* xs; const float* ys; float* C; int ldc; int M, N; float d;
  __device__ __forceinline__ void set_batch(int) {}
  __device__ __forceinline__ float apply(int r, int c, float v) const {
    if (r < M && c < N) C[(size_t)r * ldc + c] = sqrtf(fmaxf(xs[r] + ys[c] - 2.0f * v, 1e-06f) / d);
    return 0.f;
  }
  __device__ __forceinline__ void finish(float*, float) const {}
};
// relaxed_emd cost for dist_metrics 'l2' / 'both' (losses.py:18-28) at any width, from ONE product: the l2 distance
// (metric 1) or cosine + l2 (metric 2) in C, and the l2 part alone in S with its sign telling whether tf.maximum(m, 1e-6)
// passes gradient to m (m >= 1e-6: +l2) or not (-l2) -- the backward kernel needs the l2 term of its selected entries.
struct EpiRemdCost {
  static constexpr bool SYMM = false;
  const float* ra; const float* rb; const float* sa; const float* sb; float* C; float* S; int ldc; int M, N; float d; int metric;
  __device__ __forceinline__ void set_batch(int) {}
  __device__ __forceinline__ float apply(int r, int c, float v) const {
    if (r < M && c < N) {
      const float m = sa[r] + sb[c] - 2.0f * v;
      const float l2 = sqrtf(fmaxf(m, 1e-06f) / d);
      const size_t o = (size_t)r * ldc + c;
      C[o] = metric == 1 ? l2 : (1.0f - v * (ra[r] * rb[c])) + l2;
      S[o] = m >= 1e-06f ? l2 : -l2;
    }
    return 0.f;
  }
  __device__ __forceinline__ void finish(float*, float) const {}
};
struct EpiScaleStore {  // C = alpha * acc   (batched: C += z * strideC)
  static constexpr bool SYMM = false;
  float* C; int ldc; int M, N; float alpha; long long strideC;
  __device__ __forceinline__ void set_batch(int z) { C += (long long)z * strideC; }
  __device__ __forceinline__ float apply(int r, int c, float v) const {
    if (r < M && c < N) C[(size_t)r * ldc + c] = alpha * v;
    return 0.f;
  }
  __device__ __forceinline__ void finish(float*, float) const {}
};
struct EpiAxpbyBias {  // C = alpha*acc + C + bias[c]
  static constexpr bool SYMM = false;
  float* C; int ldc; int M, N; float alpha; const float* bias; float bias_scale;
  __device__ __forceinline__ void set_batch(int) {}
  __device__ __forceinline__ float apply(int r, int c, float v) const {
    if (r < M && c < N) {
      float* p = &C[(size_t)r * ldc + c];
      *p = *p + alpha * v + bias_scale * bias[c];
    }
    return 0.f;
  }
  __device__ __forceinline__ void finish(float*, float) const {}
};
// moment_matching forward (losses.py:49-52): diff = acc/n - Sx; T = sign(diff); sum |diff|.
// Both covariances are bitwise symmetric (same k order for (i,j) and (j,i)): tiles below the diagonal are skipped,
// their T entries mirrored and their |diff| counted by doubling the tile above.
struct EpiMomentFwd {
  static constexpr bool SYMM = true;
  static constexpr int symm = 1;
  const float* Sx; float* T; int ld; int M, N; float inv_n; float* partial;
  __device__ __forceinline__ float value(int r, int c, float v) const { return signf(v * inv_n - Sx[(size_t)r * ld + c]); }
  __device__ __forceinline__ float apply(int r, int c, float v) const {
    if (r < M && c < N) {
      const size_t o = (size_t)r * ld + c;
      const float diff = v * inv_n - Sx[o];
      T[o] = signf(diff);
      return fabsf(diff);
    }
    return 0.f;
  }
  __device__ __forceinline__ void mirror(int r, int c, float val) const { T[(size_t)r * ld + c] = val; }
  __device__ __forceinline__ void finish(float* red, float local) const {
    const float s = block_sum_256(blockIdx.x > blockIdx.y ? 2.f * local : local, red);
    if (threadIdx.x == 0) partial[blockIdx.y * gridDim.x + blockIdx.x] = s;   // (two-barrier kernel: 2-D grid)
  }
};
// self_similarity backward: dX[i,d] += g * r_i * (acc - xhat[i,d] * q_i), xhat = x * r_i
struct EpiSelfsimBwd {
  static constexpr bool SYMM = false;
  const float* x; const float* r; const float* q; float* dx; int ld; int M, N; float g;
  __device__ __forceinline__ void set_batch(int) {}
  __device__ __forceinline__ float apply(int row, int c, float v) const {
    if (row < M && c < N) {
      const size_t o = (size_t)row * ld + c;
      const float ri = r[row];
      dx[o] += g * ri * (v - x[o] * ri * q[row]);
    }
    return 0.f;
  }
  __device__ __forceinline__ void finish(float*, float) const {}
};

// Writes the transpose of this workgroup's BM x BN tile (values val(r, c)) to (c, r) through LDS so that the mirrored
// stores are row-contiguous.  `tile` holds >= BM * (BN + 1) floats.
// STRICT: only entries strictly above the diagonal are mirrored (a diagonal tile fills its own lower triangle).
template <int BM, int BN, int NT, bool STRICT = false, class Epi, class Val>
__device__ __forceinline__ void mirror_tile(const Epi& epi, float* tile, int m0, int n0, int M, int N, Val val) {
  __syncthreads();                                  // operands in LDS are dead
  val([&](int r, int c, float v) { tile[(r - m0) * (BN + 1) + (c - n0)] = v; });
  __syncthreads();
  for (int i = threadIdx.x; i < BM * BN; i += NT) {
    const int c = i / BM, r = i - c * BM;           // consecutive lanes: consecutive r = contiguous in the mirrored row
    if (m0 + r < M && n0 + c < N && (!STRICT || n0 + c > m0 + r)) epi.mirror(n0 + c, m0 + r, tile[r * (BN + 1) + c]);
  }
}

template <int BM, int BN, bool AKC, bool BKC, class Epi>
__global__ __launch_bounds__(256) void gemm_kernel(const float* __restrict__ A, int lda, int M,
                                                   const float* __restrict__ B, int ldb, int N, int K,
                                                   Epi epi) {
  __shared__ __attribute__((aligned(16))) float lds[OperandLds<BM>::floats + OperandLds<BN>::floats];
  float* ldsA = lds;
  float* ldsB = lds + OperandLds<BM>::floats;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  if constexpr (Epi::SYMM) {
    if (epi.symm && n0 < m0) { epi.finish(lds, 0.f); return; }      // mirrored by the tile above the diagonal
  }
  f32x16 acc[BM / 64][BN / 64];
  acc_zero<BM, BN>(acc);
  gemm_mainloop<BM, BN, AKC, BKC>(A, lda, M, B, ldb, N, K, m0, n0, ldsA, ldsB, acc);
  AccMap<BM, BN> map;
  float local = 0.f;
#pragma unroll
  for (int im = 0; im < BM / 64; ++im)
#pragma unroll
    for (int in = 0; in < BN / 64; ++in)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg)
        local += epi.apply(m0 + map.row(im, reg), n0 + map.colof(in), acc[im][in][reg]);
  if constexpr (Epi::SYMM) {
    static_assert(!Epi::SYMM || OperandLds<BM>::floats + OperandLds<BN>::floats >= BM * (BN + 1), "mirror tile");
    if (epi.symm && n0 > m0)
      mirror_tile<BM, BN, 256>(epi, lds, m0, n0, M, N, [&](auto put) {
#pragma unroll
        for (int im = 0; im < BM / 64; ++im)
#pragma unroll
          for (int in = 0; in < BN / 64; ++in)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
              const int r = m0 + map.row(im, reg), c = n0 + map.colof(in);
              put(r, c, (r < M && c < N) ? epi.value(r, c, acc[im][in][reg]) : 0.f);
            }
      });
  }
  epi.finish(lds, local);
}

// Pipelined K-contiguous x K-contiguous GEMM on the shared main loop of mfma_pipe.h.
// Batched over blockIdx.z (strideA/strideB elements, epi.set_batch(z)).
template <class Cfg, class Epi>
__global__ __launch_bounds__(Cfg::NT) void gemm_kc_pipe_kernel(const float* __restrict__ A, int lda, int M,
                                                               long long strideA, const float* __restrict__ B,
                                                               int ldb, int N, long long strideB, int K, Epi epi) {
  __shared__ __attribute__((aligned(16))) float lds[Cfg::LDS_FLOATS];
  // 1-D launch, XCD-aware tile order: N-tile fastest, then M-tile, then batch
  const unsigned gx = (N + Cfg::BN - 1) / Cfg::BN, gy = (M + Cfg::BM - 1) / Cfg::BM;
  const unsigned tile = xcd_swizzle(blockIdx.x, gridDim.x);
  const unsigned bz = tile / (gx * gy), rem = tile - bz * (gx * gy);
  A += (long long)bz * strideA;
  B += (long long)bz * strideB;
  epi.set_batch(bz);
  const int m0 = (rem / gx) * Cfg::BM, n0 = (rem % gx) * Cfg::BN;
  if constexpr (Epi::SYMM) {
    if (epi.symm && n0 < m0) return;                // mirrored by the tile above the diagonal
  }
  RowMajorLoader<Cfg, Cfg::NA> la(A, lda, m0, M, K);
  RowMajorLoader<Cfg, Cfg::NB> lb(B, ldb, n0, N, K);
  f32x16 acc[Cfg::TM][Cfg::TN];
#pragma unroll
  for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
    for (int j = 0; j < Cfg::TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  pipe_mainloop<Cfg>(lds, K >> 5, la, lb, acc);
  PipeAccMap<Cfg> map;
  float local = 0.f;
#pragma unroll
  for (int im = 0; im < Cfg::TM; ++im)
#pragma unroll
    for (int in = 0; in < Cfg::TN; ++in)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg)
        local += epi.apply(m0 + map.row(im, reg), n0 + map.colof(in), acc[im][in][reg]);
  if constexpr (Epi::SYMM) {
    static_assert(!Epi::SYMM || Cfg::LDS_FLOATS >= Cfg::BM * (Cfg::BN + 1), "mirror tile");
    if (epi.symm && n0 > m0)
      mirror_tile<Cfg::BM, Cfg::BN, Cfg::NT>(epi, lds, m0, n0, M, N, [&](auto put) {
#pragma unroll
        for (int im = 0; im < Cfg::TM; ++im)
#pragma unroll
          for (int in = 0; in < Cfg::TN; ++in)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
              const int r = m0 + map.row(im, reg), c = n0 + map.colof(in);
              put(r, c, (r < M && c < N) ? epi.value(r, c, acc[im][in][reg]) : 0.f);
            }
      });
  }
  __syncthreads();
  epi.finish(lds, local);
}

static int gemm_waves() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("STROTSS_GEMM_WAVES");
    v = e ? atoi(e) : 4;
  }
  return v;
}

template <int BM, int BN, class Epi>
int launch_pipe(const float* A, int lda, int M, long long strideA, const float* B, int ldb, int N,
                long long strideB, int K, int batch, Epi epi, hipStream_t s) {
  dim3 grid((unsigned)cdiv(N, BN) * cdiv(M, BM) * batch);
  if constexpr (BM == 128) {
    if (gemm_waves() == 8) {
      using Cfg = PipeCfg<BM, BN, (BN == 128 ? 2 : 4), (BN == 128 ? 4 : 2)>;
      hipLaunchKernelGGL((gemm_kc_pipe_kernel<Cfg, Epi>), grid, dim3(Cfg::NT), 0, s, A, lda, M, strideA, B, ldb, N,
                         strideB, K, epi);
      ST_LAUNCH_RET();
    }
  }
  {
    using Cfg = PipeCfg<BM, BN, 2, 2>;
    hipLaunchKernelGGL((gemm_kc_pipe_kernel<Cfg, Epi>), grid, dim3(Cfg::NT), 0, s, A, lda, M, strideA, B, ldb, N,
                       strideB, K, epi);
  }
  ST_LAUNCH_RET();
}

template <int BM, int BN, bool AKC, bool BKC, class Epi>
int launch(const float* A, int lda, int M, const float* B, int ldb, int N, int K, Epi epi,
           hipStream_t s) {
  dim3 grid(cdiv(N, BN), cdiv(M, BM));
  hipLaunchKernelGGL((gemm_kernel<BM, BN, AKC, BKC, Epi>), grid, dim3(256), 0, s, A, lda, M, B, ldb, N,
                     K, epi);
  ST_LAUNCH_RET();
}

}  // namespace

// C[i,j] = 1 - <x_i, y_j> rx[i] ry[j]
int st_cosine_distance(const float* x, const float* rx, int nx, const float* y, const float* ry, int ny,
                       int ld, float* C, int ldc, hipStream_t s) {
  EpiCosDist e{rx, ry, C, ldc, nx, ny, (x == y && rx == ry && nx == ny) ? 1 : 0};
  return launch_pipe<64, 64>(x, ld, nx, 0, y, ld, ny, 0, ld, 1, e, s);
}

int st_l2_distance(const float* x, const float* xs, int nx, const float* y, const float* ys, int ny, int ld, int d,
                   float* C, int ldc, hipStream_t s) {
  EpiL2Dist e{xs, ys, C, ldc, nx, ny, (float)d};
  return launch_pipe<64, 64>(x, ld, nx, 0, y, ld, ny, 0, ld, 1, e, s);
}

int st_remd_cost(const float* x, const float* rx, const float* sx, int nx, const float* y, const float* ry, const float* sy,
                 int ny, int ld, int d, int metric, float* C, float* S, int ldc, hipStream_t s) {
  EpiRemdCost e{rx, ry, sx, sy, C, S, ldc, nx, ny, (float)d, metric};
  return launch_pipe<64, 64>(x, ld, nx, 0, y, ld, ny, 0, ld, 1, e, s);
}

// C(M x N) = alpha * A^T A where A is (K x ld) row-major, rows [krows..) zero: covariance.
int st_gram_tn(const float* A, int krows, int ld, float alpha, float* C, hipStream_t s) {
  EpiScaleStore e{C, ld, ld, ld, alpha, 0};
  return launch<64, 64, false, false>(A, ld, ld, A, ld, ld, krows, e, s);
}

int st_moment_fwd_gemm(const float* cy, int krows, int ld, const float* Sx, float* T, float inv_n,
                       float* partial, int* n_partial, hipStream_t s) {
  EpiMomentFwd e{Sx, T, ld, ld, ld, inv_n, partial};
  *n_partial = cdiv(ld, 64) * cdiv(ld, 64);
  return launch<64, 64, false, false>(cy, ld, ld, cy, ld, ld, krows, e, s);
}

// dY(n x ld) += alpha * cy(n x ld) @ T(ld x ld, symmetric) + bias_scale * bias[c]
int st_moment_bwd_gemm(const float* cy, int n, int ld, const float* T, float alpha, const float* bias,
                       float bias_scale, float* dY, hipStream_t s) {
  EpiAxpbyBias e{dY, ld, n, ld, alpha, bias, bias_scale};
  return launch_pipe<64, 64>(cy, ld, n, 0, T, ld, ld, 0, ld, 1, e, s);
}

// dX(n x ld) += g * r_i (Mq(n x kpad) @ B(kpad x ld) - xhat q); B = x for self_similarity, the other side's rows for
// a cross cost matrix (sinkhorn)
int st_selfsim_bwd_gemm(const float* Mq, int ldm, int kpad, const float* bmat, const float* x, const float* r,
                        const float* q, int n, int ld, float g, float* dx, hipStream_t s) {
  EpiSelfsimBwd e{x, r, q, dx, ld, n, ld, g};
  return launch<64, 64, true, false>(Mq, ldm, n, bmat, ld, ld, kpad, e, s);
}

// Batched C[z] (M x N, ldc) = A[z] (M x K, lda) * B[z]^T (N x K, ldb): the 16 / 36 Winograd-domain GEMMs.
int st_gemm_nt_batched(const float* A, int lda, long long strideA, const float* B, int ldb, long long strideB,
                       float* C, int ldc, long long strideC, int M, int N, int K, int batch, hipStream_t s) {
  EpiScaleStore e{C, ldc, M, N, 1.0f, strideC};
  const long long t128 = (long long)cdiv(M, 128) * cdiv(N, 128);
  if (t128 * batch >= 512 && N % 128 == 0) {
    // 128x128 tiles run two workgroups per CU: 512 slots.  When the tile count ends in a fraction of a round (512-channel
    // layers at 1024 px: 36 x 32 = 1152 = 2.25 rounds), the whole batches that fill complete rounds go first and the
    // remaining batches run as 128x64 tiles, whose shorter workgroups make the tail half as long.
    static int split = -1;
    if (split < 0) { const char* env = getenv("STROTSS_GEMM_TAIL_SPLIT"); split = env ? atoi(env) : 1; }
    const long long slots = 512, total = t128 * batch;
    const long long rem = total % slots;
    if (split && total > slots && rem != 0 && rem * 2 <= slots && slots % t128 == 0) {
      const int tail = (int)(rem / t128), head = batch - tail;
      int rc = launch_pipe<128, 128>(A, lda, M, strideA, B, ldb, N, strideB, K, head, e, s);
      if (rc != 0) return rc;
      EpiScaleStore e2{C + (long long)head * strideC, ldc, M, N, 1.0f, strideC};
      return launch_pipe<128, 64>(A + (long long)head * strideA, lda, M, strideA, B + (long long)head * strideB, ldb, N,
                                  strideB, K, tail, e2, s);
    }
    return launch_pipe<128, 128>(A, lda, M, strideA, B, ldb, N, strideB, K, batch, e, s);
  }
  const long long t64 = (long long)cdiv(M, 128) * cdiv(N, 64);
  if (t64 * batch >= 512) {
    // same idea one size down (512-channel layers at 64x64 px: 36 x 16 = 576 = 1.125 rounds of two 128x64 workgroups per CU)
    static int split = -1;
    if (split < 0) { const char* env = getenv("STROTSS_GEMM_TAIL_SPLIT"); split = env ? atoi(env) : 1; }
    const long long slots = 512, total = t64 * batch, rem = total % slots;
    if (split && total > slots && rem != 0 && rem * 2 <= slots && slots % t64 == 0) {
      const int tail = (int)(rem / t64), head = batch - tail;
      int rc = launch_pipe<128, 64>(A, lda, M, strideA, B, ldb, N, strideB, K, head, e, s);
      if (rc != 0) return rc;
      EpiScaleStore e2{C + (long long)head * strideC, ldc, M, N, 1.0f, strideC};
      return launch_pipe<64, 64>(A + (long long)head * strideA, lda, M, strideA, B + (long long)head * strideB, ldb, N,
                                 strideB, K, tail, e2, s);
    }
    return launch_pipe<128, 64>(A, lda, M, strideA, B, ldb, N, strideB, K, batch, e, s);
  }
  return launch_pipe<64, 64>(A, lda, M, strideA, B, ldb, N, strideB, K, batch, e, s);
}

// ---- f32 GEMM cores on the bf16 MFMA by exact 3-way operand splitting (mfma_x3.h) -----------------
// x (batch, rows, ld) f32 row-major, first K columns -> x3 panels (batch stride 3 * rows * K bf16).
int st_x3_split_rows(const float* x, int rows, int ld, int K, long long stride_in, void* panels, int batch,
                     hipStream_t s) {
  const size_t total = (size_t)rows * (K >> 2);
  hipLaunchKernelGGL(x3_split_rows_kernel, dim3((unsigned)min((size_t)2048, (total + 255) / 256), batch), dim3(256), 0, s,
                     x, rows, ld, K, stride_in, (__bf16*)panels, (long long)3 * rows * K);
  ST_LAUNCH_RET();
}

struct EpiScaleStoreX3 : EpiScaleStore, X3NoPrefetch<EpiScaleStoreX3> {
  using EpiScaleStore::apply;
  using X3NoPrefetch<EpiScaleStoreX3>::apply;
};
struct X3NoMirror {
  template <class Epi, class Pre, class Acc, class Map>
  __device__ __forceinline__ void operator()(const Epi&, Pre&, float*, int, int, int, int, Acc&, const Map&) const {}
};
template <class Cfg>
struct X3Mirror {       // writes the transpose of the workgroup's tile through LDS (mirror_tile above)
  template <class Epi, class Pre, class Acc, class Map>
  __device__ __forceinline__ void operator()(const Epi& epi, Pre&, float* tile, int m0, int n0, int M, int N, Acc& acc,
                                             const Map& map) const {
    mirror_tile<Cfg::BM, Cfg::BN, Cfg::NT, true>(epi, tile, m0, n0, M, N, [&](auto put) {
#pragma unroll
      for (int im = 0; im < Cfg::T; ++im)
#pragma unroll
        for (int in = 0; in < Cfg::T; ++in)
#pragma unroll
          for (int reg = 0; reg < 16; ++reg) {
            const int r = m0 + map.row(im, reg), c = n0 + map.colof(in);
            put(r, c, (r < M && c < N) ? epi.value(r, c, acc[im][in][reg]) : 0.f);
          }
    });
  }
};

// C[z] (M x N, ldc) = A[z] B[z]^T, both operands as x3 panels of K columns (K % 32 == 0).
// 128 x 128 tiles (one workgroup per CU) when there are at least `min_tiles128` of them, else 64 x 64 (two per CU): few
// big tiles leave CUs idle (36 x (256 x 512 x 512): 288 big tiles on 256 CUs = 2 rounds of 1.1, 1152 small ones fill it).
int st_gemm_x3_batched(const void* A, const void* B, float* C, int ldc, long long strideC, int M, int N, int K,
                       int batch, hipStream_t s, long min_tiles128) {
  EpiScaleStoreX3 e{{C, ldc, M, N, 1.0f, strideC}, {}};
  static const bool k16 = [] { const char* v = getenv("STROTSS_X3_K16"); return !v || atoi(v) != 0; }();
  if ((long)cdiv(N, 128) * cdiv(M, 128) * batch >= min_tiles128 && k16) {
    // two co-resident workgroups per CU (K-step 16, 72 KiB ring): one computes while the other fills, stores or waits
    using Cfg = X3CfgK16<3>;
    dim3 grid((unsigned)cdiv(N, 128) * cdiv(M, 128) * batch);
    hipLaunchKernelGGL((gemm_x3_kernel<Cfg, EpiScaleStoreX3, X3NoMirror>), grid, dim3(Cfg::NT), 0, s, (const __bf16*)A, M,
                       (long long)3 * M * K, (const __bf16*)B, N, (long long)3 * N * K, K, e, X3NoMirror{});
  } else if ((long)cdiv(N, 128) * cdiv(M, 128) * batch >= min_tiles128) {
    using Cfg = X3Cfg<128>;
    dim3 grid((unsigned)cdiv(N, 128) * cdiv(M, 128) * batch);
    hipLaunchKernelGGL((gemm_x3_kernel<Cfg, EpiScaleStoreX3, X3NoMirror>), grid, dim3(Cfg::NT), 0, s, (const __bf16*)A, M,
                       (long long)3 * M * K, (const __bf16*)B, N, (long long)3 * N * K, K, e, X3NoMirror{});
  } else {
    using Cfg = X3Cfg<64>;
    dim3 grid((unsigned)cdiv(N, 64) * cdiv(M, 64) * batch);
    hipLaunchKernelGGL((gemm_x3_kernel<Cfg, EpiScaleStoreX3, X3NoMirror>), grid, dim3(Cfg::NT), 0, s, (const __bf16*)A, M,
                       (long long)3 * M * K, (const __bf16*)B, N, (long long)3 * N * K, K, e, X3NoMirror{});
  }
  ST_LAUNCH_RET();
}

// The two read-modify-write epilogues fetch what they need BEFORE the main loop (gemm_x3_kernel: Pre / prefetch), with
// clamped addresses and no conditions; only the store is predicated.
struct EpiSelfsimBwdX3 : EpiSelfsimBwd {
  template <int T> struct Pre { float dx0[T][T][16], x0[T][T][16], ri[T][16], qi[T][16]; };
  template <class P> __device__ __forceinline__ void prefetch(P& p, int im, int in, int reg, int row, int c) const {
    const int rr = min(row, M - 1);
    const size_t o = (size_t)rr * ld + min(c, N - 1);
    p.dx0[im][in][reg] = dx[o];
    p.x0[im][in][reg] = x[o];
    if (in == 0) { p.ri[im][reg] = r[rr]; p.qi[im][reg] = q[rr]; }
  }
  template <class P> __device__ __forceinline__ float apply(P& p, int im, int in, int reg, int row, int c, float v) const {
    const float ri = p.ri[im][reg];
    const float out = p.dx0[im][in][reg] + g * ri * (v - p.x0[im][in][reg] * ri * p.qi[im][reg]);
    if (row < M && c < N) dx[(size_t)row * ld + c] = out;
    return 0.f;
  }
};
struct EpiAxpbyBiasX3 : EpiAxpbyBias {
  template <int T> struct Pre { float c0[T][T][16], b[T]; };
  template <class P> __device__ __forceinline__ void prefetch(P& p, int im, int in, int reg, int r, int c) const {
    p.c0[im][in][reg] = C[(size_t)min(r, M - 1) * ldc + min(c, N - 1)];
    if (im == 0 && reg == 0) p.b[in] = bias[min(c, N - 1)];
  }
  template <class P> __device__ __forceinline__ float apply(P& p, int im, int in, int reg, int r, int c, float v) const {
    const float out = p.c0[im][in][reg] + alpha * v + bias_scale * p.b[in];
    if (r < M && c < N) C[(size_t)r * ldc + c] = out;
    return 0.f;
  }
};
// moment_matching on the x3 core.  Forward: Pt = x3 panels of the centred prediction rows TRANSPOSED (rows = feature,
// K = sample index, zero-padded to a multiple of 32); the covariance difference's sign matrix T goes out as a
// single-plane bf16 panel (values -1, 0, 1 are exact), which is the B operand of the backward product.
struct EpiMomentFwdX3 {
  static constexpr bool SYMM = true;
  static constexpr int symm = 1;
  const float* Sx; __bf16* Tp; int ld; int M, N; float inv_n; float* partial;
  unsigned first_block = 0;      // this problem's first workgroup in a grouped launch (partial[] is indexed per problem)
  __device__ __forceinline__ void set_batch(int) {}
  // The style covariance entries of the tile are fetched before the main loop (clamped addresses, no conditions): one
  // workgroup per CU has nothing to hide a load behind (inside `if`s in the epilogue they cost 45 us per tile).
  template <int T> struct Pre { float sx[T][T][16]; };
  template <class P> __device__ __forceinline__ void prefetch(P& p, int im, int in, int reg, int r, int c) const {
    p.sx[im][in][reg] = Sx[(size_t)min(r, M - 1) * ld + min(c, N - 1)];
  }
  // the owner of the upper triangle counts |diff| (twice off the diagonal); T is written by X3MomentStore below
  template <class P> __device__ __forceinline__ float apply(P& p, int im, int in, int reg, int r, int c, float v) const {
    const float a = fabsf(v * inv_n - p.sx[im][in][reg]);
    const float w = (r < M && c < N && r <= c) ? (r == c ? 1.f : 2.f) : 0.f;
    return w * a;
  }
  __device__ __forceinline__ void finish(float* red, float local) const {
    const float s = block_sum_256(local, red);
    if (threadIdx.x == 0) partial[blockIdx.x - first_block] = s;
  }
};
// Writes the tile's sign values into the single-plane panel Tp -- as they stand AND transposed (tiles below the
// diagonal are never computed) -- from an LDS copy, 8 bf16 = 16 bytes per store (2-byte global stores straight from
// the accumulators made this kernel 2.4x slower).  Diagonal tiles are symmetrised in LDS first (upper triangle wins).
template <class Cfg>
struct X3MomentStore {
  template <class Pre, class Acc, class Map>
  __device__ __forceinline__ void operator()(const EpiMomentFwdX3& epi, Pre& pre, float* tile, int m0, int n0, int M, int N,
                                             Acc& acc, const Map& map) const {
    constexpr int B = Cfg::BM, LD = B + 1;
    __syncthreads();                                  // operands in LDS are dead
#pragma unroll
    for (int im = 0; im < Cfg::T; ++im)
#pragma unroll
      for (int in = 0; in < Cfg::T; ++in)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const int r = map.row(im, reg), c = map.colof(in);
          tile[r * LD + c] = signf(acc[im][in][reg] * epi.inv_n - pre.sx[im][in][reg]);    // out-of-range: never stored
        }
    __syncthreads();
    const bool diag = n0 == m0;
    if (diag) {
      for (int i = threadIdx.x; i < B * B; i += Cfg::NT) {
        const int r = i / B, c = i - r * B;
        if (r > c) tile[r * LD + c] = tile[c * LD + r];
      }
      __syncthreads();
    }
    auto put8 = [&](int row, int k0, const float (&v)[8]) {       // 8 consecutive k of one panel row
      bf16x8 o;
#pragma unroll
      for (int q = 0; q < 8; ++q) o[q] = (__bf16)v[q];
      *reinterpret_cast<bf16x8*>(epi.Tp + ((size_t)(k0 >> 5) * epi.ld + row) * 32 + (k0 & 31)) = o;
    };
    for (int i = threadIdx.x; i < B * (B / 8); i += Cfg::NT) {   // as it stands: row m0 + r, k = n0 + c8 ..
      const int r = i / (B / 8), c8 = (i - r * (B / 8)) * 8;
      if (m0 + r >= M || n0 + c8 >= N) continue;
      float v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = tile[r * LD + c8 + q];
      put8(m0 + r, n0 + c8, v);
    }
    if (!diag) {
      for (int i = threadIdx.x; i < B * (B / 8); i += Cfg::NT) { // transposed: row n0 + c, k = m0 + r8 ..
        const int c = i % B, r8 = (i / B) * 8;
        if (n0 + c >= N || m0 + r8 >= M) continue;
        float v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = tile[(r8 + q) * LD + c];
        put8(n0 + c, m0 + r8, v);
      }
    }
  }
};

// C(ld x ld) = alpha * c^T c from the transposed x3 panels: the style covariance, by the SAME main loop, tile shape and
// product order as st_moment_fwd_x3, so that moment_matching(x, x) is exactly zero.
struct EpiSymScaleX3 : X3NoPrefetch<EpiSymScaleX3> {
  using X3NoPrefetch<EpiSymScaleX3>::apply;
  static constexpr bool SYMM = true;
  static constexpr int symm = 1;
  float* C; int ld; int M, N; float alpha;
  __device__ __forceinline__ void set_batch(int) {}
  __device__ __forceinline__ float value(int, int, float v) const { return alpha * v; }
  __device__ __forceinline__ float apply(int r, int c, float v) const {
    if (r < M && c < N && r <= c) C[(size_t)r * ld + c] = alpha * v;
    return 0.f;
  }
  __device__ __forceinline__ void mirror(int r, int c, float val) const { C[(size_t)r * ld + c] = val; }
  __device__ __forceinline__ void finish(float*, float) const {}
};
int st_gram_tn_x3(const void* Pt, int npad, int ld, float alpha, float* C, hipStream_t s) {
  using Cfg = X3CfgK16<3>;      // (round 4: the K16 ring, as st_moment_fwd_x3 and the grouped forward launch)
  EpiSymScaleX3 e{{}, C, ld, ld, ld, alpha};
  const int g = cdiv(ld, 128);
  hipLaunchKernelGGL((gemm_x3_kernel<Cfg, EpiSymScaleX3, X3Mirror<Cfg>>), dim3(g * (g + 1) / 2), dim3(Cfg::NT), 0, s,
                     (const __bf16*)Pt, ld, 0LL, (const __bf16*)Pt, ld, 0LL, npad, e, X3Mirror<Cfg>{});
  ST_LAUNCH_RET();
}

int st_moment_fwd_x3(const void* Pt, int npad, int ld, const float* Sx, void* Tp, float inv_n, float* partial,
                     int* n_partial, hipStream_t s) {
  using Cfg = X3CfgK16<3>;      // 72 KB ring, two workgroups per CU: the configuration the grouped forward launch runs it in
  EpiMomentFwdX3 e{Sx, (__bf16*)Tp, ld, ld, ld, inv_n, partial, 0u};
  const int g = cdiv(ld, 128);
  *n_partial = g * (g + 1) / 2;                      // upper-triangular launch: 171 workgroups at ld = 2208, one round
  hipLaunchKernelGGL((gemm_x3_kernel<Cfg, EpiMomentFwdX3, X3MomentStore<Cfg>>), dim3(g * (g + 1) / 2), dim3(Cfg::NT), 0, s,
                     (const __bf16*)Pt, ld, 0LL, (const __bf16*)Pt, ld, 0LL, npad, e, X3MomentStore<Cfg>{});
  ST_LAUNCH_RET();
}

// The three forward products of a train step's loss section in ONE launch (gemm_x3_group3_kernel): the covariance of the
// centred prediction rows against the style covariance (st_moment_fwd_x3: upper-triangular 128 x 128 tiles, here on the
// K16 ring so that two workgroups share a CU with the 64 x 64 tiles), the two symmetric self-similarity cost matrices
// (st_cosine_distance_x3, batch 2) and the prediction x style cost matrix of the relaxed EMD (pred-major).  Each problem
// runs the tile code of its stand-alone launch: bitwise the same matrices.
int st_loss_forward_group_x3(const void* Pt, int npad, int ld, const float* Sx, void* Tp, float inv_n, float* partial,
                             int* n_partial,
                             const void* xp, const float* rp, int n, long long pstride, long long rstride, float* Dx,
                             int ldc, long long dstride,
                             const void* xs, const float* rs, int ns, float* Ct, int ldt, hipStream_t s) {
  using CM = X3CfgK16<3>;
  using CC = X3Cfg<64>;
  const unsigned gm = cdiv(ld, 128), nm = gm * (gm + 1) / 2;
  const unsigned g = cdiv(n, 64), nc = g * (g + 1) / 2 * 2;
  const unsigned gs = cdiv(ns, 64), nr = g * gs;
  *n_partial = (int)nm;
  X3Problem<EpiMomentFwdX3> p0{(const __bf16*)Pt, ld, 0LL, (const __bf16*)Pt, ld, 0LL, npad,
                               EpiMomentFwdX3{Sx, (__bf16*)Tp, ld, ld, ld, inv_n, partial, 0u}, 0, 0, nm};
  X3Problem<EpiCosDistX3> p1{(const __bf16*)xp, n, pstride, (const __bf16*)xp, n, pstride, ld,
                             EpiCosDistX3{{rp, rp, Dx, ldc, n, n, 1}, {}, rstride, dstride}, 0, 0, nc};
  // prediction-major cost matrix Ct[j][i] = 1 - <yhat_j, xhat_i> (remd_cos_core): full grid, 2-D XCD blocking as stand-alone
  int bh = 0, bw = 0;
  if ((gs * g) % 8 == 0) {
    const unsigned per = gs * g / 8;
    unsigned best = 0;
    for (unsigned h = 1; h <= per; ++h) {
      if (per % h || g % h) continue;
      const unsigned w = per / h;
      if (gs % w) continue;
      if (!best || h + w < best + per / best) best = h;
    }
    if (best) { bh = (int)best; bw = (int)(per / best); }
  }
  X3Problem<EpiCosDistX3> p2{(const __bf16*)xp, n, 0LL, (const __bf16*)xs, ns, 0LL, ld,
                             EpiCosDistX3{{rp, rs, Ct, ldt, n, ns, 0}, {}, 0LL, 0LL}, bh, bw, nr};
  const unsigned grid = x3_pad8(nm) + x3_pad8(nc) + nr;
  hipLaunchKernelGGL((gemm_x3_group3_kernel<CM, EpiMomentFwdX3, X3MomentStore<CM>, CC, EpiCosDistX3, X3Mirror<CC>, CC, EpiCosDistX3,
                                            X3Mirror<CC>>), dim3(grid), dim3(256), 0, s, p0, p1, p2);
  ST_LAUNCH_RET();
}

// dY(n x ld) += alpha * c(n x ld) @ T(ld x ld, symmetric) + bias_scale * bias[c]: c as x3 panels, T as the
// single-plane panel written by st_moment_fwd_x3 (three partial products instead of six).
int st_moment_bwd_x3(const void* Pc, int n, int ld, const void* Tp, float alpha, const float* bias, float bias_scale,
                     float* dY, hipStream_t s) {
  using Cfg = X3Cfg<128, 1>;
  EpiAxpbyBiasX3 e{{dY, ld, n, ld, alpha, bias, bias_scale}};
  dim3 grid((unsigned)cdiv(ld, 128) * cdiv(n, 128));
  hipLaunchKernelGGL((gemm_x3_kernel<Cfg, EpiAxpbyBiasX3, X3NoMirror>), grid, dim3(Cfg::NT), 0, s, (const __bf16*)Pc, n, 0LL,
                     (const __bf16*)Tp, ld, 0LL, ld, e, X3NoMirror{});
  ST_LAUNCH_RET();
}

// dX(n x ld) += g * r_i (Mq(n x kpad) @ X(kpad x ld) - xhat q): Mq and X^T as x3 panels (rows n / rows ld, K = kpad)
int st_selfsim_bwd_x3(const void* Mp, int kpad, const void* Xt, const float* x, const float* r, const float* q, int n,
                      int ld, float g, float* dx, hipStream_t s) {
  using Cfg = X3Cfg<128>;
  EpiSelfsimBwdX3 e{{x, r, q, dx, ld, n, ld, g}};
  dim3 grid((unsigned)cdiv(ld, 128) * cdiv(n, 128));
  hipLaunchKernelGGL((gemm_x3_kernel<Cfg, EpiSelfsimBwdX3, X3NoMirror>), grid, dim3(Cfg::NT), 0, s, (const __bf16*)Mp, n, 0LL,
                     (const __bf16*)Xt, ld, 0LL, kpad, e, X3NoMirror{});
  ST_LAUNCH_RET();
}

// st_cosine_distance on x3 panels of x and y (64 x 64 tiles: 256 workgroups at 1024 x 1024, two per CU).  batch > 1:
// further matrices of the same shape at panel / norm / output strides (elements): symmetric pairs share one launch.
int st_cosine_distance_x3(const void* xp, const float* rx, int nx, const void* yp, const float* ry, int ny, int K,
                          int symm, float* C, int ldc, int batch, long long pstride, long long rstride,
                          long long cstride, hipStream_t s) {
  using Cfg = X3Cfg<64>;
  EpiCosDistX3 e{{rx, ry, C, ldc, nx, ny, symm}, {}, rstride, cstride};
  const unsigned g = cdiv(nx, 64), gxn = cdiv(ny, 64);
  dim3 grid((symm ? g * (g + 1) / 2 : gxn * g) * batch);
  // 2-D XCD blocking of the full (non-symmetric) product when the tile grid splits evenly over the 8 XCDs: block of
  // bh x bw tiles per XCD with bh * bw = tiles / 8, as square as the divisibility allows (STROTSS_X3_XCD_BLOCK=0: rows)
  int bh = 0, bw = 0;
  static int blocking = -1;
  if (blocking < 0) { const char* ev = getenv("STROTSS_X3_XCD_BLOCK"); blocking = ev ? atoi(ev) : 1; }
  if (!symm && blocking && (gxn * g) % 8 == 0) {
    const unsigned per = gxn * g / 8;
    unsigned best = 0;
    for (unsigned h = 1; h <= per; ++h) {
      if (per % h || g % h) continue;
      const unsigned w = per / h;
      if (gxn % w) continue;
      if (!best || h + w < best + per / best) best = h;
    }
    if (best) { bh = (int)best; bw = (int)(per / best); }
  }
  hipLaunchKernelGGL((gemm_x3_kernel<Cfg, EpiCosDistX3, X3Mirror<Cfg>>), grid, dim3(Cfg::NT), 0, s, (const __bf16*)xp, nx,
                     pstride, (const __bf16*)yp, ny, pstride, K, e, X3Mirror<Cfg>{}, bh, bw);
  ST_LAUNCH_RET();
}
