// Loss path of the STROTSS step (nn/losses.py:12-80, run_strotss.py:21-40): reductions and
// sparse backward passes around the MFMA cost-matrix GEMMs of gemm.hip.  Every reduction
// uses a fixed tree, so results are bitwise reproducible run to run.
#include <stdlib.h>

#include "internal.h"
#include "mfma_x3.h"

namespace {

// r[i] = 1/sqrt(max(sum x^2, 1e-12))                 tf.nn.l2_normalize (losses.py:13-14)
__global__ __launch_bounds__(256) void row_inv_norm_kernel(const float* __restrict__ x, int n, int ld,
                                                           float* __restrict__ r) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= n) return;
  const float* p = x + (size_t)row * ld;
  float s = 0.f;
  for (int k = lane * 4; k < ld; k += 256) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(p + k);
    s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
  }
  s = wave_sum(s);
  if (lane == 0) r[row] = 1.0f / sqrtf(fmaxf(s, 1e-12f));
}

// s[i] = sum x[i,:]^2      (l2_distance, losses.py:19-20)
__global__ __launch_bounds__(256) void row_sq_norm_kernel(const float* __restrict__ x, int n, int ld, float* __restrict__ out) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= n) return;
  const float* p = x + (size_t)row * ld;
  float s = 0.f;
  for (int k = lane * 4; k < ld; k += 256) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(p + k);
    s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
  }
  s = wave_sum(s);
  if (lane == 0) out[row] = s;
}

// The same pass also writing the row as x3 panels (mfma_x3.h: three bf16 planes per value, K-blocked) for the
// bf16x3 cost-matrix GEMM.  r may be NULL (norms already known).  ld % 32 == 0.
// blockIdx.y = 1 selects a second matrix (x1, n1, r1, panels1): both feature matrices of a loss in one launch.
__device__ __forceinline__ void row_inv_norm_x3_kernel_body(const float* __restrict__ x, int n, int ld,
                                                              float* __restrict__ r, __bf16* __restrict__ panels,
                                                              const float* __restrict__ x1, int n1,
                                                              float* __restrict__ r1, __bf16* __restrict__ panels1, const int bx, const int by) {
  if (by) { x = x1; n = n1; r = r1; panels = panels1; }
  const int row = bx * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= n) return;
  const float* p = x + (size_t)row * ld;
  float s = 0.f;
  for (int k = lane * 4; k < ld; k += 256) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(p + k);
    s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    x3_store4(panels, n, row, k, v);
  }
  if (r) {
    s = wave_sum(s);
    if (lane == 0) r[row] = 1.0f / sqrtf(fmaxf(s, 1e-12f));
  }
}
__global__ __launch_bounds__(256) void row_inv_norm_x3_kernel(const float* __restrict__ x, int n, int ld,
                                                              float* __restrict__ r, __bf16* __restrict__ panels,
                                                              const float* __restrict__ x1, int n1,
                                                              float* __restrict__ r1, __bf16* __restrict__ panels1) {
  row_inv_norm_x3_kernel_body(x, n, ld, r, panels, x1, n1, r1, panels1, (int)blockIdx.x, (int)blockIdx.y);
}

// s[i] = sum_{j<n} D[i,j]
__global__ __launch_bounds__(256) void row_sum_kernel(const float* __restrict__ D, int n, int ldc,
                                                      float* __restrict__ s, const float* __restrict__ D1,
                                                      float* __restrict__ s1) {
  __shared__ float red[4];
  if (blockIdx.y) { D = D1; s = s1; }                  // second matrix of the same shape
  const float* p = D + (size_t)blockIdx.x * ldc;
  float a = 0.f;
  for (int j = threadIdx.x; j < n; j += 256) a += p[j];
  a = block_sum_256(a, red);
  if (threadIdx.x == 0) s[blockIdx.x] = a;
}

// self_similarity (losses.py:55-66) on the TRANSPOSED matrices: the cosine matrices are bitwise
// symmetric (gemm.hip), so column j of D / colsum[j] == row j of D / rowsum[j].
// Row j, ONE pass over its two rows (held in registers up to n = 1024, re-read beyond):
//   sx_j = sum Dx[j,:], sy_j likewise;  A' = Dx[j,:]/sx_j, B' = Dy[j,:]/sy_j, loss_j = sum |A'-B'|,
//   S' = sign(A'-B') * sscale,  t_j = sum S' A' (0 when the clamp is active).
// Out: isx[j] = 1/max(sx_j, 1e-12), isy[j], t[j], lossrow[j] -- all the backward needs of row j:
//   dL/dDx[:,j] = Q[j,:] = (S' - t_j) * isx_j   is recomputed where it is used (selfsim_sym_kernel), never stored.
#define SS_REG 4
__device__ __forceinline__ void selfsim_rowstat_kernel_body(const float* __restrict__ Dx,
                                                              const float* __restrict__ Dy, int n, int ldc,
                                                              float sscale, float* __restrict__ isx_out,
                                                              float* __restrict__ isy_out, float* __restrict__ t_out,
                                                              float* __restrict__ lossrow, const int bx, const int by) {
  __shared__ float red[4];
  const int j = bx;
  const float* px = Dx + (size_t)j * ldc;
  const float* py = Dy + (size_t)j * ldc;
  float vx[SS_REG], vy[SS_REG];
  float sxr = 0.f, syr = 0.f;
#pragma unroll
  for (int k = 0; k < SS_REG; ++k) {
    const int i = threadIdx.x + 256 * k;
    vx[k] = i < n ? px[i] : 0.f;
    vy[k] = i < n ? py[i] : 0.f;
    sxr += vx[k]; syr += vy[k];
  }
  for (int i = threadIdx.x + 256 * SS_REG; i < n; i += 256) { sxr += px[i]; syr += py[i]; }
  sxr = block_sum_256(sxr, red);
  syr = block_sum_256(syr, red);
  const float isx = 1.0f / fmaxf(sxr, 1e-12f), isy = 1.0f / fmaxf(syr, 1e-12f);
  float l = 0.f, t = 0.f;
#pragma unroll
  for (int k = 0; k < SS_REG; ++k) {
    if (threadIdx.x + 256 * k < n) {
      const float a = vx[k] * isx, b = vy[k] * isy;
      const float d = a - b;
      l += fabsf(d);
      t += signf(d) * sscale * a;
    }
  }
  for (int i = threadIdx.x + 256 * SS_REG; i < n; i += 256) {
    const float a = px[i] * isx, b = py[i] * isy;
    const float d = a - b;
    l += fabsf(d);
    t += signf(d) * sscale * a;
  }
  l = block_sum_256(l, red);
  t = block_sum_256(t, red);
  if (!(sxr >= 1e-12f)) t = 0.f;
  if (threadIdx.x == 0) { isx_out[j] = isx; isy_out[j] = isy; t_out[j] = t; lossrow[j] = l; }
}
__global__ __launch_bounds__(256) void selfsim_rowstat_kernel(const float* __restrict__ Dx,
                                                              const float* __restrict__ Dy, int n, int ldc,
                                                              float sscale, float* __restrict__ isx_out,
                                                              float* __restrict__ isy_out, float* __restrict__ t_out,
                                                              float* __restrict__ lossrow) {
  selfsim_rowstat_kernel_body(Dx, Dy, n, ldc, sscale, isx_out, isy_out, t_out, lossrow, (int)blockIdx.x, (int)blockIdx.y);
}

// M[i,j] = -(Q[i,j] + Q[j,i]) with Q[i,j] = (sign(Dx[i,j] isx_i - Dy[i,j] isy_i) sscale - t_i) isx_i and, the cosine
// matrices being bitwise symmetric, Q[j,i] from the SAME two entries with row j's statistics: one coalesced pass over
// rows i of Dx and Dy, no N x N intermediate.  Mq[i,j] = M[i,j] * r[j] (zero for n <= j < kpad);
// qdot[i] = sum_j M[i,j] (1 - Dx[i,j])   (= xhat_i . dL/dxhat_i).  Workgroup 0 also reduces the rows' losses:
// loss_out[0] = loss_scale * sum_j lossrow[j] (fixed order).
// group_sum_256: block_sum_256's arithmetic for a GROUP of 256 threads (tid = index in the group, red = the group's four
// floats of LDS) -- the whole workgroup of the stand-alone kernel, or one quarter of a 1024-thread workgroup of
// select_sym_kernel; every thread of the workgroup reaches its barriers.
__device__ __forceinline__ float group_sum_256(float v, float* red, int tid) {
  v = wave_sum(v);
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}
struct SelfsimSymArgs {
  const float *Dx, *Dy, *isx, *isy, *tt, *r;
  int n, ldc, kpad; float sscale;
  float *Mq, *qdot; __bf16* Mp; const float* lossrow; float loss_scale; float* loss_out;
};
// row i by the 256 threads of one group; i >= n: no loads or stores, barriers only.  with_loss (uniform over the WORKGROUP):
// every group also reduces the rows' losses the same way, group i == 0 writes it.
__device__ __forceinline__ void selfsim_sym_body(const SelfsimSymArgs& a, const int i, const int tid, float* red,
                                                 const bool with_loss) {
  const float* __restrict__ Dx = a.Dx; const float* __restrict__ Dy = a.Dy;
  const float* __restrict__ isx = a.isx; const float* __restrict__ isy = a.isy;
  const float* __restrict__ tt = a.tt; const float* __restrict__ r = a.r;
  float* __restrict__ Mq = a.Mq; __bf16* __restrict__ Mp = a.Mp;
  const int n = a.n, ldc = a.ldc, kpad = a.kpad;
  const float sscale = a.sscale;
  // Mp != NULL: Mq goes out as x3 panels (rows = n, K = kpad; mfma_x3.h) for the bf16x3 backward GEMM instead
  const bool live = i < n;
  const int ic = live ? i : 0;
  const float isx_i = isx[ic], isy_i = isy[ic], t_i = tt[ic];
  float acc = 0.f;
  for (int j = tid; live && j < kpad; j += 256) {
    float out = 0.f;
    if (j < n) {
      const float dx = Dx[(size_t)i * ldc + j], dy = Dy[(size_t)i * ldc + j];
      const float isx_j = isx[j];
      const float q_ij = (signf(dx * isx_i - dy * isy_i) * sscale - t_i) * isx_i;
      const float q_ji = (signf(dx * isx_j - dy * isy[j]) * sscale - tt[j]) * isx_j;
      const float m = -(q_ij + q_ji);
      acc += m * (1.0f - dx);
      out = m * r[j];
    }
    if (Mp) {
      const __bf16 h = (__bf16)out;
      const float r1 = out - (float)h;
      const __bf16 mm = (__bf16)r1;
      const size_t o = ((size_t)(j >> 5) * 3 * n + i) * 32 + (j & 31);
      Mp[o] = h; Mp[o + (size_t)n * 32] = mm; Mp[o + (size_t)2 * n * 32] = (__bf16)(r1 - (float)mm);
    } else {
      Mq[(size_t)i * ldc + j] = out;
    }
  }
  acc = group_sum_256(acc, red, tid);
  if (tid == 0 && live) a.qdot[i] = acc;
  if (with_loss) {
    float l = 0.f;
    for (int j = tid; j < n; j += 256) l += a.lossrow[j];
    l = group_sum_256(l, red, tid);
    if (tid == 0 && i == 0) a.loss_out[0] = l * a.loss_scale;
  }
}
__global__ __launch_bounds__(256) void selfsim_sym_kernel(SelfsimSymArgs a) {
  __shared__ float red[4];
  selfsim_sym_body(a, (int)blockIdx.x, (int)threadIdx.x, red, blockIdx.x == 0);
}

// out[0] = scale * sum(partial[0..count))      (single block, fixed order)
__global__ __launch_bounds__(256) void reduce_sum_kernel(const float* __restrict__ partial, int count,
                                                         float scale, float* __restrict__ out) {
  __shared__ float red[4];
  float a = 0.f;
  for (int i = threadIdx.x; i < count; i += 256) a += partial[i];
  a = block_sum_256(a, red);
  if (threadIdx.x == 0) out[0] = a * scale;
}

// ---------------------------------------------------------------- relaxed EMD
// rmin[i] = min_j C[i,j], rcnt[i] = #{j : C[i,j] == rmin[i]}
__device__ __forceinline__ void row_min_block(const float* __restrict__ C, int row, int n, int ldc,
                                              float* __restrict__ rmin, float* __restrict__ rcnt) {
  __shared__ float red[4];
  const float* p = C + (size_t)row * ldc;
  float m = INFINITY;
  for (int j = threadIdx.x; j < n; j += 256) m = fminf(m, p[j]);
  m = block_min_256(m, red);
  float c = 0.f;
  for (int j = threadIdx.x; j < n; j += 256) c += (p[j] == m) ? 1.f : 0.f;
  c = block_sum_256(c, red);
  if (threadIdx.x == 0) { rmin[row] = m; rcnt[row] = c; }
}
// cmin[j] = min_i C[i,j], ccnt[j] = #{i : C[i,j] == cmin[j]} in two stages so that the column reduction
// uses the whole chip: stage 1 reduces COL_CHUNKS row chunks (grid (n/64, COL_CHUNKS)) to a local
// (min, count-of-min) per column, stage 2 combines them (count = sum over the chunks that attain the
// global min).  Fixed order -> bitwise reproducible.
#define COL_CHUNKS 16
__device__ __forceinline__ void col_min_partial_block(const float* __restrict__ C, int bx, int by, int ns, int n, int ldc,
                                                      float* __restrict__ pmin, float* __restrict__ pcnt) {
  __shared__ float sm[4][64];
  const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int j = bx * 64 + c;
  const int per = (ns + COL_CHUNKS - 1) / COL_CHUNKS;
  const int i0 = by * per, i1 = min(ns, i0 + per);
  float m = INFINITY;
  if (j < n)
    for (int i = i0 + g; i < i1; i += 4) m = fminf(m, C[(size_t)i * ldc + j]);
  sm[g][c] = m;
  __syncthreads();
  m = fminf(fminf(sm[0][c], sm[1][c]), fminf(sm[2][c], sm[3][c]));
  __syncthreads();
  float cnt = 0.f;
  if (j < n)
    for (int i = i0 + g; i < i1; i += 4) cnt += (C[(size_t)i * ldc + j] == m) ? 1.f : 0.f;
  sm[g][c] = cnt;
  __syncthreads();
  if (g == 0 && j < n) {
    pmin[(size_t)by * ldc + j] = m;
    pcnt[(size_t)by * ldc + j] = (sm[0][c] + sm[1][c]) + (sm[2][c] + sm[3][c]);
  }
}
// Row minima (one workgroup per row: rows x n matrix) and stage 1 of the column minima in ONE launch:
// workgroups [0, rows) take the rows, the following cdiv(n, 64) * COL_CHUNKS the column chunks.
__device__ __forceinline__ void row_col_min_kernel_body(const float* __restrict__ C, int rows, int n, int ldc,
                                                        float* __restrict__ rmin, float* __restrict__ rcnt,
                                                        float* __restrict__ pmin, float* __restrict__ pcnt, const int b) {
  if (b < rows) {
    row_min_block(C, b, n, ldc, rmin, rcnt);
  } else {
    const int gx = (n + 63) / 64, bb = b - rows;
    col_min_partial_block(C, bb % gx, bb / gx, rows, n, ldc, pmin, pcnt);
  }
}
__global__ __launch_bounds__(256) void row_col_min_kernel(const float* __restrict__ C, int rows, int n, int ldc,
                                                          float* __restrict__ rmin, float* __restrict__ rcnt,
                                                          float* __restrict__ pmin, float* __restrict__ pcnt) {
  row_col_min_kernel_body(C, rows, n, ldc, rmin, rcnt, pmin, pcnt, (int)blockIdx.x);
}
// Stage 2 of the column minima + the choice of the REMD branch, one workgroup of 1024 threads (a column per thread:
// its 2 x COL_CHUNKS partials are all in flight at once):
//   cmin[j] = min over the chunks, ccnt[j] = count of the minimum;
//   loss = max(R_X, R_Y) with R_X = mean of `xmin` (minima per style row), R_Y = mean of the minima per prediction
//   row; sel[0] = 1 when the R_X branch carries the gradient (tf.maximum: first argument on ties).
// col_is_x: the column minima are the R_X side (pred-major cost matrix of the cosine REMD), else the R_Y side.
// pmin == NULL: cmin / ccnt are final already (palette: computed directly), only the means and the branch are taken.
__device__ __forceinline__ float block_sum_1024(float v, float* red) {     // fixed order
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float a = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k) a += red[k];
  return a;
}
__device__ __forceinline__ void col_min_final_select_kernel_body(const float* __restrict__ pmin,
                                                                  const float* __restrict__ pcnt, int n, int ldc,
                                                                  float* __restrict__ cmin, float* __restrict__ ccnt,
                                                                  const float* __restrict__ rowmin, int rows,
                                                                  int col_is_x, float* __restrict__ loss_out,
                                                                  int* __restrict__ sel, int swapped, float* red) {
  float a = 0.f, b = 0.f;
  for (int j = threadIdx.x; j < n; j += 1024) {
    if (pmin) {
      float pm[COL_CHUNKS], pc[COL_CHUNKS];
#pragma unroll
      for (int k = 0; k < COL_CHUNKS; ++k) { pm[k] = pmin[(size_t)k * ldc + j]; pc[k] = pcnt[(size_t)k * ldc + j]; }
      float m = INFINITY;
#pragma unroll
      for (int k = 0; k < COL_CHUNKS; ++k) m = fminf(m, pm[k]);
      float cnt = 0.f;
#pragma unroll
      for (int k = 0; k < COL_CHUNKS; ++k) cnt += (pm[k] == m) ? pc[k] : 0.f;
      cmin[j] = m;
      ccnt[j] = cnt;
      a += m;
    } else {
      a += cmin[j];
    }
  }
  for (int i = threadIdx.x; i < rows; i += 1024) b += rowmin[i];
  a = block_sum_1024(a, red);
  b = block_sum_1024(b, red);
  if (threadIdx.x == 0) {
    const float mc = a / (float)n, mr = b / (float)rows;
    const float rx = col_is_x ? mc : mr, ry = col_is_x ? mr : mc;
    // tf.maximum(R_X, R_Y) sends a tie to its FIRST argument.  swapped: the caller asked for the gradient w.r.t. the
    // reference's first argument and passed it as `pred` -- the kernel's "style" side is then the reference's R_Y
    const int s = swapped ? (rx > ry) : (rx >= ry);
    loss_out[0] = s ? rx : ry;
    sel[0] = s;
  }
}
__global__ __launch_bounds__(1024) void col_min_final_select_kernel(const float* __restrict__ pmin,
                                                                    const float* __restrict__ pcnt, int n, int ldc,
                                                                    float* __restrict__ cmin, float* __restrict__ ccnt,
                                                                    const float* __restrict__ rowmin, int rows,
                                                                    int col_is_x, float* __restrict__ loss_out,
                                                                    int* __restrict__ sel, int swapped) {
  __shared__ float red[16];
  col_min_final_select_kernel_body(pmin, pcnt, n, ldc, cmin, ccnt, rowmin, rows, col_is_x, loss_out, sel, swapped, red);
}

#define REMD_MAX_LIST 2048
// One block per pred sample j.  Builds the (ordered) list of style rows whose cost entry
// carries gradient for column j, then dY_j += g * ry_j (ghat - yhat (yhat.ghat)), ghat = -sum w xhat_i.
// C is the PRED-major cost matrix Ct[j][i] (row j contiguous: the two scans of "column j" are coalesced).
__global__ __launch_bounds__(256) void remd_cos_bwd_kernel(
    const float* __restrict__ C, int ldc, const float* __restrict__ style, const float* __restrict__ rs,
    int ns, const float* __restrict__ pred, const float* __restrict__ rp, int n, int ld,
    const float* __restrict__ rmin, const float* __restrict__ rcnt, const float* __restrict__ cmin,
    const float* __restrict__ ccnt, const int* __restrict__ sel, float gscale, float* __restrict__ gpred) {
  __shared__ int li[REMD_MAX_LIST];
  __shared__ float lw[REMD_MAX_LIST];
  __shared__ int cnts[256];
  __shared__ float red[4];
  const int j = blockIdx.x, t = threadIdx.x;
  const int row_branch = sel[0];
  const int per = (ns + 255) / 256;
  const int i0 = t * per, i1 = min(ns, i0 + per);
  const float cm = cmin[j], cc = ccnt[j];
  int c = 0;
  for (int i = i0; i < i1; ++i) {
    const float v = C[(size_t)j * ldc + i];
    c += row_branch ? (v == rmin[i]) : (v == cm);
  }
  // exclusive prefix of the per-thread counts: shuffle scan inside the wave + the 4 wave totals through LDS (ordered)
  int incl = c;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int up = __shfl_up(incl, o, 64);
    if ((t & 63) >= o) incl += up;
  }
  if ((t & 63) == 63) cnts[t >> 6] = incl;
  __syncthreads();
  int off = incl - c, total = 0;
#pragma unroll
  for (int wv = 0; wv < 4; ++wv) {
    off += (wv < (t >> 6)) ? cnts[wv] : 0;
    total += cnts[wv];
  }
  float qpart = 0.f;
  for (int i = i0; i < i1; ++i) {
    const float v = C[(size_t)j * ldc + i];
    const bool hit = row_branch ? (v == rmin[i]) : (v == cm);
    if (hit) {
      const float w = row_branch ? 1.0f / ((float)ns * rcnt[i]) : 1.0f / ((float)n * cc);
      if (off < REMD_MAX_LIST) { li[off] = i; lw[off] = w * rs[i]; }
      qpart += w * (1.0f - v);
      ++off;
    }
  }
  const float q = -block_sum_256(qpart, red);  // yhat_j . ghat_j   (also syncs li/lw)
  total = min(total, REMD_MAX_LIST);
  if (total == 0) return;
  const float rj = rp[j];
  const float live = (rj < 1.0f / sqrtf(1e-12f)) ? 1.f : 0.f;
  const float* y = pred + (size_t)j * ld;
  float* gy = gpred + (size_t)j * ld;
  // The list's rows are summed in list order per channel (bitwise reproducible).  A prediction row that many style rows point
  // at (row branch) used to walk its list once per 256-channel slice, eight 4-byte loads in flight: ~9 x L/8 dependent L2
  // round trips.  Now a thread owns up to three float4 columns (k4, k4 + 256, k4 + 512 of the ld / 4 <= 768) and walks the
  // list ONCE with 8 entries x 3 sixteen-byte loads in flight: L/8 round trips, the same additions in the same order.
  const int nv = ld >> 2;
  const f32x4* __restrict__ S4 = (const f32x4*)style;
  const f32x4* __restrict__ y4 = (const f32x4*)y;
  f32x4* __restrict__ gy4 = (f32x4*)gy;
  for (int k0 = t; k0 < nv; k0 += 768) {
    const int k1 = min(k0 + 256, nv - 1), k2 = min(k0 + 512, nv - 1);      // clamped: loaded unconditionally, stored if in range
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0;
    int e = 0;
    for (; e + 8 <= total; e += 8) {
      f32x4 v0[8], v1[8], v2[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const f32x4* row = S4 + (size_t)li[e + u] * nv;
        v0[u] = row[k0]; v1[u] = row[k1]; v2[u] = row[k2];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const float w = lw[e + u];
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) {
          a0[c4] = __builtin_fmaf(w, v0[u][c4], a0[c4]);
          a1[c4] = __builtin_fmaf(w, v1[u][c4], a1[c4]);
          a2[c4] = __builtin_fmaf(w, v2[u][c4], a2[c4]);
        }
      }
    }
    for (; e < total; ++e) {
      const f32x4* row = S4 + (size_t)li[e] * nv;
      const f32x4 u0 = row[k0], u1 = row[k1], u2 = row[k2];
      const float w = lw[e];
#pragma unroll
      for (int c4 = 0; c4 < 4; ++c4) {
        a0[c4] = __builtin_fmaf(w, u0[c4], a0[c4]);
        a1[c4] = __builtin_fmaf(w, u1[c4], a1[c4]);
        a2[c4] = __builtin_fmaf(w, u2[c4], a2[c4]);
      }
    }
    const float s = gscale * rj;
    {
      f32x4 g = gy4[k0]; const f32x4 yy = y4[k0];
#pragma unroll
      for (int c4 = 0; c4 < 4; ++c4) g[c4] += s * (-a0[c4] - yy[c4] * rj * q * live);
      gy4[k0] = g;
    }
    if (k0 + 256 < nv) {
      f32x4 g = gy4[k1]; const f32x4 yy = y4[k1];
#pragma unroll
      for (int c4 = 0; c4 < 4; ++c4) g[c4] += s * (-a1[c4] - yy[c4] * rj * q * live);
      gy4[k1] = g;
    }
    if (k0 + 512 < nv) {
      f32x4 g = gy4[k2]; const f32x4 yy = y4[k2];
#pragma unroll
      for (int c4 = 0; c4 < 4; ++c4) g[c4] += s * (-a2[c4] - yy[c4] * rj * q * live);
      gy4[k2] = g;
    }
  }
}

// The same for dist_metrics 'l2' (metric 1) and 'both' (metric 2) at any width (losses.py:18-28, 69-80).  C: the cost
// matrix the minima were taken of (pred-major), S: its l2 part, negative where tf.maximum(m, 1e-6) blocks the gradient.
//   d l2_ji / d y_j = (y_j - x_i) / (D l2_ji)   [m_ji >= 1e-6],      cosine part as in remd_cos_bwd_kernel with
//   cos_ji = C_ji - l2_ji.
__global__ __launch_bounds__(256) void remd_generic_bwd_kernel(
    const float* __restrict__ C, const float* __restrict__ S, int ldc, const float* __restrict__ style,
    const float* __restrict__ rs, int ns, const float* __restrict__ pred, const float* __restrict__ rp, int n, int ld,
    float dwidth, int metric, const float* __restrict__ rmin, const float* __restrict__ rcnt,
    const float* __restrict__ cmin, const float* __restrict__ ccnt, const int* __restrict__ sel, float gscale,
    float* __restrict__ gpred) {
  __shared__ int li[REMD_MAX_LIST];
  __shared__ float lw[REMD_MAX_LIST];
  __shared__ float la[REMD_MAX_LIST];
  __shared__ int cnts[256];
  __shared__ float red[4];
  const int j = blockIdx.x, t = threadIdx.x;
  const int row_branch = sel[0];
  const int per = (ns + 255) / 256;
  const int i0 = t * per, i1 = min(ns, i0 + per);
  const float cm = cmin[j], cc = ccnt[j];
  int c = 0;
  for (int i = i0; i < i1; ++i) {
    const float v = C[(size_t)j * ldc + i];
    c += row_branch ? (v == rmin[i]) : (v == cm);
  }
  int incl = c;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int up = __shfl_up(incl, o, 64);
    if ((t & 63) >= o) incl += up;
  }
  if ((t & 63) == 63) cnts[t >> 6] = incl;
  __syncthreads();
  int off = incl - c, total = 0;
#pragma unroll
  for (int wv = 0; wv < 4; ++wv) {
    off += (wv < (t >> 6)) ? cnts[wv] : 0;
    total += cnts[wv];
  }
  float qpart = 0.f, apart = 0.f;
  for (int i = i0; i < i1; ++i) {
    const float v = C[(size_t)j * ldc + i];
    const bool hit = row_branch ? (v == rmin[i]) : (v == cm);
    if (hit) {
      const float w = row_branch ? 1.0f / ((float)ns * rcnt[i]) : 1.0f / ((float)n * cc);
      const float sl = S[(size_t)j * ldc + i], l2 = fabsf(sl);
      const float a = sl > 0.f ? w / (dwidth * l2) : 0.f;
      if (off < REMD_MAX_LIST) { li[off] = i; lw[off] = metric == 2 ? w * rs[i] : 0.f; la[off] = a; }
      if (metric == 2) qpart += w * (1.0f - (v - l2));
      apart += a;
      ++off;
    }
  }
  const float q = -block_sum_256(qpart, red);
  const float asum = block_sum_256(apart, red);     // (also syncs li / lw / la)
  total = min(total, REMD_MAX_LIST);
  if (total == 0) return;
  const float rj = metric == 2 ? rp[j] : 0.f;
  const float live = (rj < 1.0f / sqrtf(1e-12f)) ? 1.f : 0.f;
  const float* y = pred + (size_t)j * ld;
  float* gy = gpred + (size_t)j * ld;
  for (int k = t; k < ld; k += 256) {
    float acc = 0.f, accl = 0.f;
    for (int e = 0; e < total; ++e) {
      const float xv = style[(size_t)li[e] * ld + k];
      acc += lw[e] * xv;
      accl += la[e] * xv;
    }
    gy[k] += gscale * (rj * (-acc - y[k] * rj * q * live) + (asum * y[k] - accl));
  }
}

// ---------------------------------------------------------------- palette (D = 3, pure VALU)
// yuv[i] = (Y, U, V, r) with r the inverse norm of the YUV vector  (strotss_utils.py:166-167)
__device__ __forceinline__ void palette_prepare_kernel_body(const float* __restrict__ feat, int n, int ld,
                                                              f32x4* __restrict__ yuv, int convert,
                                                              const float* __restrict__ feat1, int n1,
                                                              f32x4* __restrict__ yuv1, const int bx, const int by) {
  if (by) { feat = feat1; n = n1; yuv = yuv1; }    // second matrix (style and prediction in one launch)
  const int i = bx * 256 + threadIdx.x;
  if (i >= n) return;
  const float* p = feat + (size_t)i * ld;
  const float R = p[0], G = p[1], B = p[2];
  float Y = R, U = G, V = B;
  if (convert) {
    Y = R * 0.299f + G * 0.587f + B * 0.114f;
    U = R * -0.14714119f + G * -0.28886916f + B * 0.43601035f;
    V = R * 0.61497538f + G * -0.51496512f + B * -0.10001026f;
  }
  const float ss = Y * Y + U * U + V * V;
  f32x4 o = {Y, U, V, 1.0f / sqrtf(fmaxf(ss, 1e-12f))};
  yuv[i] = o;
}
__global__ __launch_bounds__(256) void palette_prepare_kernel(const float* __restrict__ feat, int n, int ld,
                                                              f32x4* __restrict__ yuv, int convert,
                                                              const float* __restrict__ feat1, int n1,
                                                              f32x4* __restrict__ yuv1) {
  palette_prepare_kernel_body(feat, n, ld, yuv, convert, feat1, n1, yuv1, (int)blockIdx.x, (int)blockIdx.y);
}
__device__ __forceinline__ void palette_pair(const f32x4 a, const f32x4 b, float& ccos, float& m) {
  const float dot = a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
  ccos = 1.0f - dot * (a[3] * b[3]);
  const float xs = a[0] * a[0] + a[1] * a[1] + a[2] * a[2];
  const float ys = b[0] * b[0] + b[1] * b[1] + b[2] * b[2];
  m = xs + ys - 2.0f * dot;                        // losses.py:22
}
// C[i,j] = cosine + sqrt(max(m,1e-6)/3)           dist_metrics['both'] (losses.py:27-28).  The N_s x N matrix is never
// stored: the minima kernel and the backward kernel both evaluate this ONE function on the same operands in the same
// order, so `cost == minimum` compares bitwise equal values in both.
__device__ __forceinline__ float palette_cost(const f32x4 x, const f32x4 y, float& ccos, float& m) {
  palette_pair(x, y, ccos, m);
  return ccos + sqrtf(fmaxf(m, 1e-6f) / 3.0f);
}
// Minima and their multiplicities, rows and columns in ONE launch: workgroup b < ns takes style row b (minimum over
// the predictions j), workgroup ns + j the prediction j (minimum over the style rows i).
__device__ __forceinline__ void palette_minima_kernel_body(const f32x4* __restrict__ ys, int ns,
                                                           const f32x4* __restrict__ yp, int n,
                                                           float* __restrict__ rmin, float* __restrict__ rcnt,
                                                           float* __restrict__ cmin, float* __restrict__ ccnt, const int b) {
  __shared__ float red[4];
  const bool row = b < ns;
  const f32x4 mine = row ? ys[b] : yp[b - ns];
  const f32x4* other = row ? yp : ys;
  const int cnt_other = row ? n : ns;
  float mn = INFINITY;
  for (int k = threadIdx.x; k < cnt_other; k += 256) {
    float cc, m;
    const f32x4 o = other[k];
    mn = fminf(mn, row ? palette_cost(mine, o, cc, m) : palette_cost(o, mine, cc, m));
  }
  mn = block_min_256(mn, red);
  float c = 0.f;
  for (int k = threadIdx.x; k < cnt_other; k += 256) {
    float cc, m;
    const f32x4 o = other[k];
    c += ((row ? palette_cost(mine, o, cc, m) : palette_cost(o, mine, cc, m)) == mn) ? 1.f : 0.f;
  }
  c = block_sum_256(c, red);
  if (threadIdx.x == 0) {
    if (row) { rmin[b] = mn; rcnt[b] = c; } else { cmin[b - ns] = mn; ccnt[b - ns] = c; }
  }
}
__global__ __launch_bounds__(256) void palette_minima_kernel(const f32x4* __restrict__ ys, int ns,
                                                             const f32x4* __restrict__ yp, int n,
                                                             float* __restrict__ rmin, float* __restrict__ rcnt,
                                                             float* __restrict__ cmin, float* __restrict__ ccnt) {
  palette_minima_kernel_body(ys, ns, yp, n, rmin, rcnt, cmin, ccnt, (int)blockIdx.x);
}
// One wave per pred sample j.
__global__ __launch_bounds__(64) void palette_bwd_kernel(
    const f32x4* __restrict__ ys, int ns, const f32x4* __restrict__ yp,
    int n, const float* __restrict__ rmin, const float* __restrict__ rcnt, const float* __restrict__ cmin,
    const float* __restrict__ ccnt, const int* __restrict__ sel, float gscale, float* __restrict__ gpred,
    int ld, int convert) {
  const int j = blockIdx.x, lane = threadIdx.x;
  const int row_branch = sel[0];
  const f32x4 y = yp[j];
  const float cm = cmin[j], cc = ccnt[j];
  float gh0 = 0.f, gh1 = 0.f, gh2 = 0.f, q = 0.f, sk = 0.f, kx0 = 0.f, kx1 = 0.f, kx2 = 0.f;
  for (int i = lane; i < ns; i += 64) {
    const f32x4 x = ys[i];
    float ccos, m;
    const float v = palette_cost(x, y, ccos, m);
    const bool hit = row_branch ? (v == rmin[i]) : (v == cm);
    if (hit) {
      const float w = row_branch ? 1.0f / ((float)ns * rcnt[i]) : 1.0f / ((float)n * cc);
      const float wr = w * x[3];
      gh0 -= wr * x[0]; gh1 -= wr * x[1]; gh2 -= wr * x[2];
      q -= w * (1.0f - ccos);
      if (m >= 1e-6f) {
        const float k = w / (3.0f * sqrtf(m / 3.0f));
        sk += k; kx0 += k * x[0]; kx1 += k * x[1]; kx2 += k * x[2];
      }
    }
  }
  gh0 = wave_sum(gh0); gh1 = wave_sum(gh1); gh2 = wave_sum(gh2); q = wave_sum(q);
  sk = wave_sum(sk); kx0 = wave_sum(kx0); kx1 = wave_sum(kx1); kx2 = wave_sum(kx2);
  if (lane == 0) {
    const float rj = y[3];
    const float live = (rj < 1.0f / sqrtf(1e-12f)) ? 1.f : 0.f;
    const float d0 = rj * (gh0 - y[0] * rj * q * live) + sk * y[0] - kx0;
    const float d1 = rj * (gh1 - y[1] * rj * q * live) + sk * y[1] - kx1;
    const float d2 = rj * (gh2 - y[2] * rj * q * live) + sk * y[2] - kx2;
    // back through yuv = rgb @ M :  d_rgb = d_yuv @ M^T
    float* g = gpred + (size_t)j * ld;
    if (convert) {
      g[0] += gscale * (d0 * 0.299f + d1 * -0.14714119f + d2 * 0.61497538f);
      g[1] += gscale * (d0 * 0.587f + d1 * -0.28886916f + d2 * -0.51496512f);
      g[2] += gscale * (d0 * 0.114f + d1 * 0.43601035f + d2 * -0.10001026f);
    } else {
      g[0] += gscale * d0; g[1] += gscale * d1; g[2] += gscale * d2;
    }
  }
}

// ---------------------------------------------------------------- moment matching helpers
// mean[c] = (1/n) sum_{i<n} y[i,c] in two stages (grid (ld/64, COL_CHUNKS) partial sums, then a fixed-order
// combine), so the column reduction uses the whole chip.
__device__ __forceinline__ void col_sum_partial_kernel_body(const float* __restrict__ y, int n, int ld,
                                                              float* __restrict__ psum, const int bx, const int by) {
  __shared__ float sm[4][64];
  const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int col = bx * 64 + c;
  const int per = (n + COL_CHUNKS - 1) / COL_CHUNKS;
  const int i0 = by * per, i1 = min(n, i0 + per);
  float a = 0.f;
  if (col < ld)
    for (int i = i0 + g; i < i1; i += 4) a += y[(size_t)i * ld + col];
  sm[g][c] = a;
  __syncthreads();
  if (g == 0 && col < ld) psum[(size_t)by * ld + col] = (sm[0][c] + sm[1][c]) + (sm[2][c] + sm[3][c]);
}
__global__ __launch_bounds__(256) void col_sum_partial_kernel(const float* __restrict__ y, int n, int ld,
                                                              float* __restrict__ psum) {
  col_sum_partial_kernel_body(y, n, ld, psum, (int)blockIdx.x, (int)blockIdx.y);
}
__global__ __launch_bounds__(256) void col_mean_final_kernel(const float* __restrict__ psum, int n, int ld,
                                                             float* __restrict__ mean) {
  const int col = blockIdx.x * 256 + threadIdx.x;
  if (col >= ld) return;
  float a = 0.f;
#pragma unroll
  for (int k = 0; k < COL_CHUNKS; ++k) a += psum[(size_t)k * ld + col];
  mean[col] = a / (float)n;
}
// cy[i,c] = y[i,c] - mean[c] for i < n, 0 for n <= i < rows
__global__ __launch_bounds__(256) void center_kernel(const float* __restrict__ y, int n, int rows, int ld,
                                                     const float* __restrict__ mean, float* __restrict__ cy) {
  const size_t total4 = (size_t)rows * ld / 4;
  const int ld4 = ld / 4;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total4; e += (size_t)gridDim.x * 256) {
    const int i = (int)(e / ld4), c4 = (int)(e % ld4);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (i < n) {
      v = reinterpret_cast<const f32x4*>(y)[e];
      const f32x4 m = reinterpret_cast<const f32x4*>(mean)[c4];
      v = v - m;
    }
    reinterpret_cast<f32x4*>(cy)[e] = v;
  }
}
// The centred rows as x3 panels (mfma_x3.h) for the bf16x3 moment GEMMs, both ways round: Pc (rows = samples i < n,
// K = ld features: the backward product's A) and Pt (rows = ld features, K = npad samples, zeros for i >= n: both
// operands of the covariance).  One 32 x 32 tile per workgroup, transposed through LDS; grid (ld/32, npad/32).
// mean == NULL: no centring; Pc == NULL: transposed panels only (the self-similarity backward's B operand).
__device__ __forceinline__ void center_x3_kernel_body(const float* __restrict__ y, int n, int npad, int ld,
                                                        const float* __restrict__ mean, __bf16* __restrict__ Pc,
                                                        __bf16* __restrict__ Pt, const float* __restrict__ psum,
                                                        float* __restrict__ mean_out, const int bx, const int by) {
  // psum != NULL: the column means come from col_sum_partial_kernel's partial sums here (the arithmetic of
  // col_mean_final_kernel, chunk order and division included -- one launch less), block row 0 also stores them
  __shared__ float tile[32][33];
  __shared__ __attribute__((aligned(16))) float smean[32];
  const int j0 = bx * 32, i0 = by * 32;
  if (psum) {
    if (threadIdx.x < 32) {
      const int col = j0 + threadIdx.x;
      float a = 0.f;
#pragma unroll
      for (int k = 0; k < COL_CHUNKS; ++k) a += psum[(size_t)k * ld + col];
      a = a / (float)n;
      smean[threadIdx.x] = a;
      if (mean_out && by == 0) mean_out[col] = a;
    }
    __syncthreads();
  }
  {
    const int il = threadIdx.x >> 3, j4 = threadIdx.x & 7;
    const int i = i0 + il, j = j0 + 4 * j4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (i < n) {
      v = *reinterpret_cast<const f32x4*>(y + (size_t)i * ld + j);
      if (psum) v = v - *reinterpret_cast<const f32x4*>(smean + 4 * j4);
      else if (mean) v = v - *reinterpret_cast<const f32x4*>(mean + j);
      if (Pc) x3_store4(Pc, n, i, j, v);
    }
    tile[il][4 * j4 + 0] = v[0]; tile[il][4 * j4 + 1] = v[1]; tile[il][4 * j4 + 2] = v[2]; tile[il][4 * j4 + 3] = v[3];
  }
  __syncthreads();
  {
    const int ic = threadIdx.x & 7, jl = threadIdx.x >> 3;
    const f32x4 v = {tile[4 * ic][jl], tile[4 * ic + 1][jl], tile[4 * ic + 2][jl], tile[4 * ic + 3][jl]};
    x3_store4(Pt, ld, j0 + jl, i0 + 4 * ic, v);
  }
}
__global__ __launch_bounds__(256) void center_x3_kernel(const float* __restrict__ y, int n, int npad, int ld,
                                                        const float* __restrict__ mean, __bf16* __restrict__ Pc,
                                                        __bf16* __restrict__ Pt, const float* __restrict__ psum = nullptr,
                                                        float* __restrict__ mean_out = nullptr) {
  center_x3_kernel_body(y, n, npad, ld, mean, Pc, Pt, psum, mean_out, (int)blockIdx.x, (int)blockIdx.y);
}
// loss = sum(partial)/d^2 + sum_c |mx-my|/d ;  sgn[c] = sign(my - mx)
__device__ __forceinline__ void moment_finalize_kernel_body(const float* __restrict__ partial, int count,
                                                              const float* __restrict__ mx,
                                                              const float* __restrict__ my, int d, int ld,
                                                              float* __restrict__ sgn,
                                                              float* __restrict__ loss_out, const int bx, const int by) {
  __shared__ float red[4];
  float a = 0.f, b = 0.f;
  for (int i = threadIdx.x; i < count; i += 256) a += partial[i];
  for (int c = threadIdx.x; c < ld; c += 256) {
    const float df = (c < d) ? (my[c] - mx[c]) : 0.f;
    b += fabsf(df);
    sgn[c] = signf(df);
  }
  a = block_sum_256(a, red);
  b = block_sum_256(b, red);
  if (threadIdx.x == 0) loss_out[0] = a / ((float)d * (float)d) + b / (float)d;
}
__global__ __launch_bounds__(256) void moment_finalize_kernel(const float* __restrict__ partial, int count,
                                                              const float* __restrict__ mx,
                                                              const float* __restrict__ my, int d, int ld,
                                                              float* __restrict__ sgn,
                                                              float* __restrict__ loss_out) {
  moment_finalize_kernel_body(partial, count, mx, my, d, ld, sgn, loss_out, (int)blockIdx.x, (int)blockIdx.y);
}

// ---------------------------------------------------------------- Sinkhorn-Knopp (build-defined, see strotss_hip.h)
// All matrices are stored pred-major: Mt[j][i], j < n (prediction row), i < ns (style row), row stride ldm.
#define SK_EPS 1e-12f
__global__ __launch_bounds__(256) void sk_fill_kernel(float* __restrict__ x, int n, float v) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) x[i] = v;
}
__global__ __launch_bounds__(256) void sk_exp_kernel(const float* __restrict__ Mt, size_t total, float l,
                                                     float* __restrict__ Kt) {
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) Kt[e] = __expf(-l * Mt[e]);
}
// out[i] = sum_j Kt[j][i] * (Mt ? Mt[j][i] * wm[j] + w[j] : w[j])   -- two-stage column reduction, stage 1
__global__ __launch_bounds__(256) void sk_coldot_partial_kernel(const float* __restrict__ Kt, const float* __restrict__ Mt,
                                                                int n, int ns, int ldm, const float* __restrict__ w,
                                                                const float* __restrict__ wm, float* __restrict__ part) {
  __shared__ float sm[4][64];
  const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + c;
  const int per = (n + COL_CHUNKS - 1) / COL_CHUNKS;
  const int j0 = blockIdx.y * per, j1 = min(n, j0 + per);
  float a = 0.f;
  if (col < ns)
    for (int j = j0 + g; j < j1; j += 4) {
      const size_t o = (size_t)j * ldm + col;
      const float wj = w ? w[j] : 0.f;
      a += Kt[o] * (Mt ? Mt[o] * wm[j] + wj : wj);
    }
  sm[g][c] = a;
  __syncthreads();
  if (g == 0 && col < ns) part[(size_t)blockIdx.y * ns + col] = (sm[0][c] + sm[1][c]) + (sm[2][c] + sm[3][c]);
}
// stage 2.  mode 0 (forward): a = sum -> u[i] = px / max(a, eps).
//           mode 1 (backward): gu = sum -> da[i] = a was above eps ? -gu u[i]^2 / px : 0   (u = px / max(a, eps))
__global__ __launch_bounds__(256) void sk_col_final_kernel(const float* __restrict__ part, int ns, int mode, float px,
                                                           float* __restrict__ u, float* __restrict__ da) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= ns) return;
  float a = 0.f;
#pragma unroll
  for (int k = 0; k < COL_CHUNKS; ++k) a += part[(size_t)k * ns + i];
  if (mode == 0) {
    u[i] = px / fmaxf(a, SK_EPS);
  } else {
    const float ui = u[i];
    da[i] = (ui < px / SK_EPS) ? -a * ui * ui / px : 0.f;
  }
}
// b[j] = sum_i Kt[j][i] * (Mt ? Mt[j][i] : 1) * w[i], one workgroup per row j.
// mode 0 (forward): v[j] = py / max(b, eps).
// mode 1 (backward): gv = b -> db[j] = -gv vref[j]^2 / py (0 where vref's b was clamped); with Mt also cost[j] = vref[j] * b.
__global__ __launch_bounds__(256) void sk_rowdot_kernel(const float* __restrict__ Kt, const float* __restrict__ Mt,
                                                        int ns, int ldm, const float* __restrict__ w, int mode,
                                                        float py, float* __restrict__ v,
                                                        const float* __restrict__ vref, float* __restrict__ db,
                                                        float* __restrict__ cost) {
  __shared__ float red[4];
  const int j = blockIdx.x;
  const float* kr = Kt + (size_t)j * ldm;
  const float* mr = Mt ? Mt + (size_t)j * ldm : nullptr;
  float a = 0.f;
  for (int i = threadIdx.x; i < ns; i += 256) a += kr[i] * (mr ? mr[i] : 1.f) * w[i];
  const float b = block_sum_256(a, red);
  if (threadIdx.x == 0) {
    if (mode == 0) {
      v[j] = py / fmaxf(b, SK_EPS);
    } else {
      const float vj = vref[j];
      db[j] = (vj < py / SK_EPS) ? -b * vj * vj / py : 0.f;
      if (cost) cost[j] = vj * b;
    }
  }
}
// Gradient assembly, one workgroup per prediction row j:
//   dM[j][i] = K (u_T[i] v_T[j] (1 - l M) - l sum_t (U_t[i] DB_t[j] + DA_t[i] V_{t-1}[j]))
//   W[j][i] = -dM * rs[i]   (operand of the backward GEMM against the style rows),   q[j] = sum_i -dM (1 - M)
__global__ __launch_bounds__(256) void sk_assemble_kernel(const float* __restrict__ Kt, const float* __restrict__ Mt,
                                                          int ns, int ldm, int T, float l, const float* __restrict__ U,
                                                          const float* __restrict__ DA, const float* __restrict__ V,
                                                          const float* __restrict__ DB, int n,
                                                          const float* __restrict__ rs, float* __restrict__ W,
                                                          float* __restrict__ q) {
  __shared__ float red[4];
  __shared__ float dbj[64], vpj[64];
  const int j = blockIdx.x;
  for (int t = threadIdx.x; t < T; t += 256) { dbj[t] = DB[(size_t)t * n + j]; vpj[t] = V[(size_t)t * n + j]; }   // V[t] = v_{t-1}
  __syncthreads();
  const float vT = V[(size_t)T * n + j];
  float qs = 0.f;
  for (int i = threadIdx.x; i < ldm; i += 256) {
    float w = 0.f;
    if (i < ns) {
      const size_t o = (size_t)j * ldm + i;
      float s = 0.f;
      for (int t = 0; t < T; ++t) s += U[(size_t)t * ns + i] * dbj[t] + DA[(size_t)t * ns + i] * vpj[t];
      const float m = Mt[o];
      const float dm = Kt[o] * (U[(size_t)(T - 1) * ns + i] * vT * (1.f - l * m) - l * s);
      w = -dm;
      qs += w * (1.f - m);
      w *= rs[i];
    }
    W[(size_t)j * ldm + i] = w;
  }
  const float tot = block_sum_256(qs, red);
  if (threadIdx.x == 0) q[j] = tot;
}

// ---- merged launches of strotss_step_losses_fwd_bwd (round 4): kernels that depend on the same inputs and on nothing else
// share ONE launch, each keeping its own blocks and its own arithmetic (the `_body` functions above are the kernels' bodies).
// Prologue: [A] reciprocal norms + x3 panels of the prediction and content rows, [B] column partial sums of the prediction
// rows, [C] YUV rows of style and prediction (palette term), [D] the prediction rows transposed as x3 panels (B operand of
// the self-similarity backward GEMM) -- all functions of the gathered rows alone.
struct StepPrologueArgs {
  const float* pred; const float* content; int n; int ld;
  float* rp; __bf16* xp; float* rc; __bf16* xc;                 // A
  float* psum;                                                  // B
  const float* style; int ns; f32x4* ys; f32x4* yp; int rgb_to_yuv;   // C
  int ldc; __bf16* xt;                                          // D
  int nA, nB, nC;                                               // block counts of the first three ranges
};
__global__ __launch_bounds__(256) void step_losses_prologue_kernel(StepPrologueArgs a) {
  int b = (int)blockIdx.x;
  if (b < a.nA) {
    const int half = a.nA / 2;
    row_inv_norm_x3_kernel_body(a.pred, a.n, a.ld, a.rp, a.xp, a.content, a.n, a.rc, a.xc, b % half, b / half);
    return;
  }
  b -= a.nA;
  if (b < a.nB) {
    const int gx = (a.ld + 63) / 64;
    col_sum_partial_kernel_body(a.pred, a.n, a.ld, a.psum, b % gx, b / gx);
    return;
  }
  b -= a.nB;
  if (b < a.nC) {
    const int half = a.nC / 2;
    palette_prepare_kernel_body(a.style, a.ns, a.ld, a.ys, a.rgb_to_yuv, a.pred, a.n, a.yp, b % half, b / half);
    return;
  }
  b -= a.nC;
  const int gx = a.ld / 32;
  center_x3_kernel_body(a.pred, a.n, a.ldc, a.ld, (const float*)nullptr, (__bf16*)nullptr, a.xt, (const float*)nullptr,
                        (float*)nullptr, b % gx, b / gx);
}
// After the grouped forward GEMMs, everything that reads their results (or the prologue's) and nothing else, in ONE launch:
// [self-similarity row statistics (n blocks) | moment-matching finalisation (1) | REMD row minima + stage 1 of the column
// minima (n + cdiv(ns, 64) * COL_CHUNKS) | palette minima (ns + n)].  Every range keeps its own blocks and arithmetic.
struct StepStatsArgs {
  const float *Dx, *Dy; int n, ldc; float sscale; float *isx, *isy, *tt, *lossrow;                 // self-similarity rows
  const float* partial; int count; const float *mx, *my; int d, ld; float *sgn, *loss_moment;       // moment finalisation
  const float* C; int ns, ldt; float *rmin, *rcnt, *pmin, *pcnt;                                   // REMD minima
  const f32x4 *ys, *yp; float *prmin, *prcnt, *pcmin, *pccnt;                                      // palette minima
};
__global__ __launch_bounds__(256) void step_losses_stats_kernel(StepStatsArgs a) {
  int b = (int)blockIdx.x;
  if (b < a.n) { selfsim_rowstat_kernel_body(a.Dx, a.Dy, a.n, a.ldc, a.sscale, a.isx, a.isy, a.tt, a.lossrow, b, 0); return; }
  b -= a.n;
  if (b == 0) { moment_finalize_kernel_body(a.partial, a.count, a.mx, a.my, a.d, a.ld, a.sgn, a.loss_moment, 0, 0); return; }
  b -= 1;
  const int nmin = a.n + ((a.ns + 63) / 64) * COL_CHUNKS;
  if (b < nmin) { row_col_min_kernel_body(a.C, a.n, a.ns, a.ldt, a.rmin, a.rcnt, a.pmin, a.pcnt, b); return; }
  b -= nmin;
  palette_minima_kernel_body(a.ys, a.ns, a.yp, a.n, a.prmin, a.prcnt, a.pcmin, a.pccnt, b);
}
// [stage 2 of the REMD column minima + branch (block 0) | palette means + branch (block 1) | the symmetrised self-similarity
// gradient matrix, four rows per 1024-thread block]: all three read only what step_losses_stats_kernel wrote.
struct SelectArgs {
  const float *pmin, *pcnt; int n, ldc; float *cmin, *ccnt; const float* rowmin; int rows, col_is_x; float* loss_out; int* sel;
  int swapped;
};
__global__ __launch_bounds__(1024) void step_losses_select_sym_kernel(SelectArgs r, SelectArgs p, SelfsimSymArgs y) {
  __shared__ float red[16];
  if (blockIdx.x < 2) {
    const SelectArgs& a = blockIdx.x == 0 ? r : p;
    col_min_final_select_kernel_body(a.pmin, a.pcnt, a.n, a.ldc, a.cmin, a.ccnt, a.rowmin, a.rows, a.col_is_x, a.loss_out,
                                     a.sel, a.swapped, red);
    return;
  }
  const int g = (int)threadIdx.x >> 8;
  selfsim_sym_body(y, ((int)blockIdx.x - 2) * 4 + g, (int)threadIdx.x & 255, red + 4 * g, blockIdx.x == 2);
}

// The same assembly for dist_metrics 'l2' / 'both' (losses.py:18-28): M = l2 or cosine + l2; S holds the l2 part with the sign
// of tf.maximum(m, 1e-6)'s gradient rule (EpiRemdCost).  dM as above, then per metric
//   l2 part:      w2 = dM * [S > 0] / (D |S|);   W2[j][i] = -w2,  q2[j] = -sum_i w2      (dy_j = y_j sum w2 - sum w2 x_i, r = 1)
//   cosine part:  W[j][i] = -dM * rs[i],  q[j] = sum_i -dM (1 - cos),  cos = M - |S|     ('both' only)
__global__ __launch_bounds__(256) void sk_assemble_metric_kernel(const float* __restrict__ Kt, const float* __restrict__ Mt,
                                                                 const float* __restrict__ S, int ns, int ldm, int T, float l,
                                                                 const float* __restrict__ U, const float* __restrict__ DA,
                                                                 const float* __restrict__ V, const float* __restrict__ DB, int n,
                                                                 const float* __restrict__ rs, float dfeat, int metric,
                                                                 float* __restrict__ W, float* __restrict__ q,
                                                                 float* __restrict__ W2, float* __restrict__ q2) {
  __shared__ float red[4];
  __shared__ float dbj[64], vpj[64];
  const int j = blockIdx.x;
  for (int t = threadIdx.x; t < T; t += 256) { dbj[t] = DB[(size_t)t * n + j]; vpj[t] = V[(size_t)t * n + j]; }
  __syncthreads();
  const float vT = V[(size_t)T * n + j];
  float qs = 0.f, qs2 = 0.f;
  for (int i = threadIdx.x; i < ldm; i += 256) {
    float w = 0.f, w2 = 0.f;
    if (i < ns) {
      const size_t o = (size_t)j * ldm + i;
      float s = 0.f;
      for (int t = 0; t < T; ++t) s += U[(size_t)t * ns + i] * dbj[t] + DA[(size_t)t * ns + i] * vpj[t];
      const float m = Mt[o], sv = S[o], l2 = fabsf(sv);
      const float dm = Kt[o] * (U[(size_t)(T - 1) * ns + i] * vT * (1.f - l * m) - l * s);
      w2 = sv > 0.f ? dm / (dfeat * l2) : 0.f;
      qs2 -= w2;
      w2 = -w2;
      if (metric == STROTSS_METRIC_BOTH) {
        w = -dm;
        qs += w * (1.f - (m - l2));
        w *= rs[i];
      }
    }
    W2[(size_t)j * ldm + i] = w2;
    if (metric == STROTSS_METRIC_BOTH) W[(size_t)j * ldm + i] = w;
  }
  const float tot2 = block_sum_256(qs2, red);
  const float tot = block_sum_256(qs, red);
  if (threadIdx.x == 0) { q2[j] = tot2; q[j] = tot; }
}

#define CHK(expr)            \
  do {                       \
    int rc__ = (expr);       \
    if (rc__ != 0) return rc__; \
  } while (0)
#define LAUNCH_OK()                                  \
  do {                                               \
    hipError_t e__ = hipGetLastError();              \
    if (e__ != hipSuccess) return (int)e__;          \
  } while (0)

bool feat_ok(int n, int d, int ld) { return n > 0 && d > 0 && ld >= d; }

// STROTSS_X3 = 0 or STROTSS_X3_MOMENT = 0 keeps the covariance GEMMs of moment_matching on the f32 MFMA.
bool moment_x3() {
  static int on = -1;
  if (on < 0) {
    const char* e = getenv("STROTSS_X3"); on = e ? atoi(e) : 1;
    const char* c = getenv("STROTSS_X3_MOMENT"); if (c && atoi(c) == 0) on = 0;
  }
  return on != 0;
}

// STROTSS_X3 = 0 or STROTSS_X3_COST = 0 keeps the cosine cost matrices on the f32 MFMA (default: bf16x3 core,
// csrc/mfma_x3.h).
bool cost_x3() {
  static int on = -1;
  if (on < 0) {
    const char* e = getenv("STROTSS_X3"); on = e ? atoi(e) : 1;
    const char* c = getenv("STROTSS_X3_COST"); if (c && atoi(c) == 0) on = 0;
  }
  return on != 0;
}

struct SelfsimWs {
  float *rp, *rc, *Dx, *Dy, *sx, *sy, *tt, *Mq, *qdot, *lossrow;
  __bf16 *xp, *xc;               // x3 panels of pred / content (ld > 0)
  __bf16 *xt, *mp;               // x3 panels of pred transposed (rows = ld, K = ldc) and of Mq (rows = n, K = ldc)
  int ldc;
  bool plan(Workspace& w, int n, int ld) {
    ldc = round_up(n, 32);
    xp = w.take<__bf16>((size_t)3 * n * ld); xc = w.take<__bf16>((size_t)3 * n * ld);
    xt = w.take<__bf16>((size_t)3 * ld * ldc); mp = w.take<__bf16>((size_t)3 * n * ldc);
    rp = w.take<float>(ldc); rc = w.take<float>(ldc);
    Dx = w.take<float>((size_t)n * ldc); Dy = w.take<float>((size_t)n * ldc);
    sx = w.take<float>(ldc); sy = w.take<float>(ldc); tt = w.take<float>(ldc);
    Mq = w.take<float>((size_t)n * ldc);
    qdot = w.take<float>(ldc); lossrow = w.take<float>(ldc);
    return w.ok();
  }
};
struct RemdWs {
  float *rp, *C, *rmin, *rcnt, *cmin, *ccnt, *pmin, *pcnt;
  f32x4 *ys, *yp;
  int* sel;
  __bf16 *xp, *xs;     // x3 panels of pred / style (cosine REMD, ld > 0)
  int ldc, ldt;        // row stride of the style-major C[i][j] (palette) / of the pred-major Ct[j][i] (cosine REMD)
  bool plan(Workspace& w, int ns, int n, int ld) {
    ldc = round_up(n, 32);
    xp = w.take<__bf16>((size_t)3 * n * ld); xs = w.take<__bf16>((size_t)3 * ns * ld);
    ldt = round_up(ns, 32);
    const int ldm = ldc > ldt ? ldc : ldt;
    rp = w.take<float>(ldc);
    C = w.take<float>((size_t)ns * ldc > (size_t)n * ldt ? (size_t)ns * ldc : (size_t)n * ldt);
    rmin = w.take<float>(ldt); rcnt = w.take<float>(ldt);
    cmin = w.take<float>(ldc); ccnt = w.take<float>(ldc);
    pmin = w.take<float>((size_t)COL_CHUNKS * ldm); pcnt = w.take<float>((size_t)COL_CHUNKS * ldm);
    ys = w.take<f32x4>(ns); yp = w.take<f32x4>(n);
    sel = w.take<int>(4);
    return w.ok();
  }
};
struct RemdGenWs {          // relaxed_emd with dist_metrics 'l2' / 'both' at any width
  float *rp, *rs, *sp, *ss, *C, *S, *rmin, *rcnt, *cmin, *ccnt, *pmin, *pcnt;
  int* sel;
  int ldt;
  bool plan(Workspace& w, int ns, int n) {
    ldt = round_up(ns, 32);
    const int ldn = round_up(n, 32), ldm = ldn > ldt ? ldn : ldt;
    rp = w.take<float>(ldn); sp = w.take<float>(ldn); rs = w.take<float>(ldt); ss = w.take<float>(ldt);
    C = w.take<float>((size_t)n * ldt); S = w.take<float>((size_t)n * ldt);
    rmin = w.take<float>(ldt); rcnt = w.take<float>(ldt); cmin = w.take<float>(ldn); ccnt = w.take<float>(ldn);
    pmin = w.take<float>((size_t)COL_CHUNKS * ldm); pcnt = w.take<float>((size_t)COL_CHUNKS * ldm);
    sel = w.take<int>(4);
    return w.ok();
  }
};
struct SinkhornWs {
  float *rp, *Mt, *Kt, *W, *U, *DA, *V, *DB, *part, *q, *cost;
  int ldm;
  bool plan(Workspace& w, int ns, int n, int T) {
    ldm = round_up(ns, 32);
    const int rows = round_up(n, 64);
    rp = w.take<float>(round_up(n, 32));
    Mt = w.take<float>((size_t)rows * ldm); Kt = w.take<float>((size_t)rows * ldm); W = w.take<float>((size_t)rows * ldm);
    U = w.take<float>((size_t)T * ns); DA = w.take<float>((size_t)T * ns);
    V = w.take<float>((size_t)(T + 1) * n); DB = w.take<float>((size_t)T * n);
    part = w.take<float>((size_t)COL_CHUNKS * ns);
    q = w.take<float>(n); cost = w.take<float>(n);
    return w.ok();
  }
};
struct MomentWs {
  float *mean, *cy, *T, *partial, *sgn, *psum;
  __bf16 *Pc, *Pt, *Tp;          // x3 panels: centred rows, their transpose, the sign matrix (one plane)
  int rows;
  bool plan(Workspace& w, int n, int ld) {
    rows = round_up(n, 32);
    Pc = w.take<__bf16>((size_t)3 * n * ld); Pt = w.take<__bf16>((size_t)3 * ld * rows); Tp = w.take<__bf16>((size_t)ld * ld);
    mean = w.take<float>(ld);
    cy = w.take<float>((size_t)rows * ld);
    T = w.take<float>((size_t)ld * ld);
    partial = w.take<float>((size_t)cdiv(ld, 64) * cdiv(ld, 64));
    sgn = w.take<float>(ld);
    psum = w.take<float>((size_t)COL_CHUNKS * ld);
    return w.ok();
  }
};

}  // namespace

extern "C" {

int strotss_row_inv_norm(const float* x, int n, int ld, float* r, void* stream) {
  ST_CHECK_ARG(x && r && n > 0 && ld > 0, STROTSS_EINVAL);
  ST_CHECK_ARG(ld % 4 == 0, STROTSS_EALIGN);
  hipLaunchKernelGGL(row_inv_norm_kernel, dim3(cdiv(n, 4)), dim3(256), 0, (hipStream_t)stream, x, n, ld, r);
  ST_LAUNCH_RET();
}

int strotss_cosine_distance(const float* x, const float* rx, int nx, const float* y, const float* ry,
                            int ny, int ld, float* C, int ldc, void* stream) {
  ST_CHECK_ARG(x && rx && y && ry && C && nx > 0 && ny > 0 && ldc >= ny, STROTSS_EINVAL);
  ST_CHECK_ARG(ld % 32 == 0 && ld > 0, STROTSS_EALIGN);
  return st_cosine_distance(x, rx, nx, y, ry, ny, ld, C, ldc, (hipStream_t)stream);
}

int strotss_l2_distance(const float* x, int nx, const float* y, int ny, int d, int ld, float* C, int ldc,
                        float* workspace, void* stream) {
  ST_CHECK_ARG(x && y && C && workspace && nx > 0 && ny > 0 && d > 0 && d <= ld && ldc >= ny, STROTSS_EINVAL);
  ST_CHECK_ARG(ld % 32 == 0, STROTSS_EALIGN);
  hipStream_t st = (hipStream_t)stream;
  float* xs = workspace;
  float* ys = workspace + nx;
  hipLaunchKernelGGL(row_sq_norm_kernel, dim3(cdiv(nx, 4)), dim3(256), 0, st, x, nx, ld, xs);
  hipLaunchKernelGGL(row_sq_norm_kernel, dim3(cdiv(ny, 4)), dim3(256), 0, st, y, ny, ld, ys);
  LAUNCH_OK();
  return st_l2_distance(x, xs, nx, y, ys, ny, ld, d, C, ldc, st);
}

int strotss_row_inv_norm_x3(const float* x, int n, int ld, float* r, void* panels, void* stream) {
  ST_CHECK_ARG(x && panels && n > 0 && ld > 0, STROTSS_EINVAL);
  ST_CHECK_ARG(ld % 32 == 0, STROTSS_EALIGN);
  hipLaunchKernelGGL(row_inv_norm_x3_kernel, dim3(cdiv(n, 4)), dim3(256), 0, (hipStream_t)stream, x, n, ld, r,
                     (__bf16*)panels, (const float*)nullptr, 0, (float*)nullptr, (__bf16*)nullptr);
  ST_LAUNCH_RET();
}

int strotss_cosine_distance_x3(const void* xp, const float* rx, int nx, const void* yp, const float* ry, int ny,
                               int ld, float* C, int ldc, void* stream) {
  ST_CHECK_ARG(xp && rx && yp && ry && C && nx > 0 && ny > 0 && ldc >= ny, STROTSS_EINVAL);
  ST_CHECK_ARG(ld % 32 == 0 && ld > 0, STROTSS_EALIGN);
  return st_cosine_distance_x3(xp, rx, nx, yp, ry, ny, ld, (xp == yp && rx == ry && nx == ny) ? 1 : 0, C, ldc, 1, 0, 0, 0,
                               (hipStream_t)stream);
}

size_t strotss_selfsim_workspace_bytes(int n, int ld) {
  Workspace w = Workspace::planner();
  SelfsimWs s;
  s.plan(w, n, ld);
  return w.off;
}

int strotss_selfsim_fwd_bwd(const float* pred, const float* content, int n, int d, int ld, float gscale,
                            float* gpred, float* loss_out, void* workspace, size_t workspace_bytes,
                            void* stream) {
  ST_CHECK_ARG(pred && content && gpred && loss_out && workspace && feat_ok(n, d, ld), STROTSS_EINVAL);
  ST_CHECK_ARG(ld % 32 == 0, STROTSS_EALIGN);
  Workspace w(workspace, workspace_bytes);
  SelfsimWs s;
  ST_CHECK_ARG(s.plan(w, n, ld), STROTSS_EINVAL);
  hipStream_t st = (hipStream_t)stream;
  const int ldc = s.ldc;
  if (cost_x3()) {      // cost matrices on the bf16x3 core: the norm pass also writes the rows' x3 panels
    hipLaunchKernelGGL(row_inv_norm_x3_kernel, dim3(cdiv(n, 4), 2), dim3(256), 0, st, pred, n, ld, s.rp, s.xp, content, n,
                       s.rc, s.xc);
    LAUNCH_OK();
    // both symmetric matrices in ONE launch (2 x 136 upper-triangular tiles at n = 1024: one round of the 512 slots)
    CHK(st_cosine_distance_x3(s.xp, s.rp, n, s.xp, s.rp, n, ld, 1, s.Dx, ldc, 2, s.xc - s.xp, s.rc - s.rp, s.Dy - s.Dx, st));
  } else {
    hipLaunchKernelGGL(row_inv_norm_kernel, dim3(cdiv(n, 4)), dim3(256), 0, st, pred, n, ld, s.rp);
    hipLaunchKernelGGL(row_inv_norm_kernel, dim3(cdiv(n, 4)), dim3(256), 0, st, content, n, ld, s.rc);
    LAUNCH_OK();
    CHK(st_cosine_distance(pred, s.rp, n, pred, s.rp, n, ld, s.Dx, ldc, st));
    CHK(st_cosine_distance(content, s.rc, n, content, s.rc, n, ld, s.Dy, ldc, st));
  }
  // loss = mean(|A-B|) * n = (1/n) sum |A-B|  ->  dL/dA = sign/n.  Two passes over the two matrices in all: the rows'
  // statistics (sx and sy hold the RECIPROCAL clamped row sums), then the symmetrised gradient + the loss
  hipLaunchKernelGGL(selfsim_rowstat_kernel, dim3(n), dim3(256), 0, st, s.Dx, s.Dy, n, ldc, 1.0f / (float)n, s.sx, s.sy,
                     s.tt, s.lossrow);
  const bool bx3 = cost_x3();
  hipLaunchKernelGGL(selfsim_sym_kernel, dim3(n), dim3(256), 0, st,
                     SelfsimSymArgs{s.Dx, s.Dy, s.sx, s.sy, s.tt, s.rp, n, ldc, ldc, 1.0f / (float)n, s.Mq, s.qdot,
                                    bx3 ? s.mp : (__bf16*)nullptr, s.lossrow, 1.0f / (float)n, loss_out});
  if (bx3)
    hipLaunchKernelGGL(center_x3_kernel, dim3(ld / 32, ldc / 32), dim3(256), 0, st, pred, n, ldc, ld, (const float*)nullptr,
                       (__bf16*)nullptr, s.xt);
  LAUNCH_OK();
  if (bx3) return st_selfsim_bwd_x3(s.mp, ldc, s.xt, pred, s.rp, s.qdot, n, ld, gscale, gpred, st);
  return st_selfsim_bwd_gemm(s.Mq, ldc, ldc, pred, pred, s.rp, s.qdot, n, ld, gscale, gpred, st);
}

size_t strotss_sinkhorn_workspace_bytes(int ns, int n, int n_iter) {
  Workspace w = Workspace::planner();
  SinkhornWs s;
  s.plan(w, ns, n, n_iter);
  return w.off;
}

// Sinkhorn scalings, cost and the reverse sweep on a given pred-major cost matrix s.Mt (n x ldm): fills s.Kt, U, V, DA, DB and
// loss_out; the caller assembles dM and the chain rule of its metric.
static int sinkhorn_iterate(SinkhornWs& s, int ns, int n, float l, int T, float* loss_out, hipStream_t st) {
  const int ldm = s.ldm;
  const float px = 1.0f / (float)ns, py = 1.0f / (float)n;
  const dim3 gcol(cdiv(ns, 64), COL_CHUNKS), gfin(cdiv(ns, 256));
  hipLaunchKernelGGL(sk_exp_kernel, dim3(min(4096, cdiv((size_t)n * ldm, 256))), dim3(256), 0, st, s.Mt,
                     (size_t)n * ldm, l, s.Kt);
  // v_0 = 1
  hipLaunchKernelGGL(sk_fill_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, s.V, n, 1.0f);
  for (int t = 1; t <= T; ++t) {                     // u_t = px / (K v_{t-1}),  v_t = py / (K^T u_t)
    hipLaunchKernelGGL(sk_coldot_partial_kernel, gcol, dim3(256), 0, st, s.Kt, (const float*)nullptr, n, ns, ldm,
                       s.V + (size_t)(t - 1) * n, (const float*)nullptr, s.part);
    hipLaunchKernelGGL(sk_col_final_kernel, gfin, dim3(256), 0, st, s.part, ns, 0, px, s.U + (size_t)(t - 1) * ns,
                       (float*)nullptr);
    hipLaunchKernelGGL(sk_rowdot_kernel, dim3(n), dim3(256), 0, st, s.Kt, (const float*)nullptr, ns, ldm,
                       s.U + (size_t)(t - 1) * ns, 0, py, s.V + (size_t)t * n, (const float*)nullptr, (float*)nullptr,
                       (float*)nullptr);
  }
  LAUNCH_OK();
  // cost = sum_j v_T[j] ((K o M)^T u_T)[j]; the same pass gives gv -> db_T
  const float* uT = s.U + (size_t)(T - 1) * ns;
  const float* vT = s.V + (size_t)T * n;
  hipLaunchKernelGGL(sk_rowdot_kernel, dim3(n), dim3(256), 0, st, s.Kt, s.Mt, ns, ldm, uT, 1, py, (float*)nullptr, vT,
                     s.DB + (size_t)(T - 1) * n, s.cost);
  hipLaunchKernelGGL(reduce_sum_kernel, dim3(1), dim3(256), 0, st, s.cost, n, 1.0f, loss_out);
  for (int t = T; t >= 1; --t) {
    // gu = [t == T] (K o M) v_T + K db_t  ->  da_t ;   gv = K^T da_t  ->  db_{t-1}
    const bool top = t == T;
    hipLaunchKernelGGL(sk_coldot_partial_kernel, gcol, dim3(256), 0, st, s.Kt, top ? s.Mt : (const float*)nullptr, n, ns,
                       ldm, s.DB + (size_t)(t - 1) * n, top ? vT : (const float*)nullptr, s.part);
    hipLaunchKernelGGL(sk_col_final_kernel, gfin, dim3(256), 0, st, s.part, ns, 1, px, s.U + (size_t)(t - 1) * ns,
                       s.DA + (size_t)(t - 1) * ns);
    if (t > 1)
      hipLaunchKernelGGL(sk_rowdot_kernel, dim3(n), dim3(256), 0, st, s.Kt, (const float*)nullptr, ns, ldm,
                         s.DA + (size_t)(t - 1) * ns, 1, py, (float*)nullptr, s.V + (size_t)(t - 1) * n,
                         s.DB + (size_t)(t - 2) * n, (float*)nullptr);
  }
  LAUNCH_OK();
  const int rows = round_up(n, 64);
  if (rows > n) CHK((int)hipMemsetAsync(s.W + (size_t)n * ldm, 0, sizeof(float) * (size_t)(rows - n) * ldm, st));
  return 0;
}

int strotss_sinkhorn_cos_fwd_bwd(const float* style, const float* rs, int ns, const float* pred, int n, int d,
                                 int ld, float l, int n_iter, float gscale, float* gpred, float* loss_out,
                                 void* workspace, size_t workspace_bytes, void* stream) {
  ST_CHECK_ARG(style && rs && pred && gpred && loss_out && workspace && ns > 0 && feat_ok(n, d, ld), STROTSS_EINVAL);
  ST_CHECK_ARG(ld % 32 == 0, STROTSS_EALIGN);
  ST_CHECK_ARG(l > 0.f && n_iter >= 1 && n_iter <= 64, STROTSS_ERANGE);
  Workspace w(workspace, workspace_bytes);
  SinkhornWs s;
  ST_CHECK_ARG(s.plan(w, ns, n, n_iter), STROTSS_EINVAL);
  hipStream_t st = (hipStream_t)stream;
  const int ldm = s.ldm, T = n_iter;
  hipLaunchKernelGGL(row_inv_norm_kernel, dim3(cdiv(n, 4)), dim3(256), 0, st, pred, n, ld, s.rp);
  LAUNCH_OK();
  CHK(st_cosine_distance(pred, s.rp, n, style, rs, ns, ld, s.Mt, ldm, st));          // Mt[j][i] = 1 - <yhat_j, xhat_i>
  CHK(sinkhorn_iterate(s, ns, n, l, T, loss_out, st));
  hipLaunchKernelGGL(sk_assemble_kernel, dim3(n), dim3(256), 0, st, s.Kt, s.Mt, ns, ldm, T, l, s.U, s.DA, s.V, s.DB, n, rs,
                     s.W, s.q);
  LAUNCH_OK();
  return st_selfsim_bwd_gemm(s.W, ldm, ldm, style, pred, s.rp, s.q, n, ld, gscale, gpred, st);
}

// sinkhorn_knopp with dist_metrics 'l2' / 'both' (losses.py:27-28, 83-105; build-defined like the cosine form): the cost
// matrix from st_remd_cost (one f32-MFMA product, the distance in its epilogue), the same scalings and reverse sweep, the
// metric's chain rule in the assembly, one (l2) or two (both) backward GEMMs against the style rows.
struct SinkhornMetricWs {
  float *rs, *sp, *ss, *S, *W2, *q2, *ones;
  bool plan(Workspace& w, int ns, int n) {
    const int ldm = round_up(ns, 32), rows = round_up(n, 64);
    rs = w.take<float>(ldm); sp = w.take<float>(round_up(n, 32)); ss = w.take<float>(ldm);
    S = w.take<float>((size_t)rows * ldm); W2 = w.take<float>((size_t)rows * ldm);
    q2 = w.take<float>(n); ones = w.take<float>(round_up(n, 32));
    return w.ok();
  }
};

size_t strotss_sinkhorn_metric_workspace_bytes(int ns, int n, int n_iter) {
  Workspace w = Workspace::planner();
  SinkhornWs s; SinkhornMetricWs m;
  s.plan(w, ns, n, n_iter); m.plan(w, ns, n);
  return w.off;
}

int strotss_sinkhorn_metric_fwd_bwd(const float* style, int ns, const float* pred, int n, int d, int ld, int metric, float l,
                                    int n_iter, float gscale, float* gpred, float* loss_out, void* workspace,
                                    size_t workspace_bytes, void* stream) {
  ST_CHECK_ARG(style && pred && gpred && loss_out && workspace && ns > 0 && feat_ok(n, d, ld), STROTSS_EINVAL);
  ST_CHECK_ARG(metric == STROTSS_METRIC_L2 || metric == STROTSS_METRIC_BOTH, STROTSS_EINVAL);
  ST_CHECK_ARG(ld % 32 == 0, STROTSS_EALIGN);
  ST_CHECK_ARG(l > 0.f && n_iter >= 1 && n_iter <= 64, STROTSS_ERANGE);
  Workspace w(workspace, workspace_bytes);
  SinkhornWs s; SinkhornMetricWs m;
  ST_CHECK_ARG(s.plan(w, ns, n, n_iter) && m.plan(w, ns, n), STROTSS_EINVAL);
  hipStream_t st = (hipStream_t)stream;
  const int ldm = s.ldm, T = n_iter;
  hipLaunchKernelGGL(row_inv_norm_kernel, dim3(cdiv(n, 4)), dim3(256), 0, st, pred, n, ld, s.rp);
  hipLaunchKernelGGL(row_inv_norm_kernel, dim3(cdiv(ns, 4)), dim3(256), 0, st, style, ns, ld, m.rs);
  hipLaunchKernelGGL(row_sq_norm_kernel, dim3(cdiv(n, 4)), dim3(256), 0, st, pred, n, ld, m.sp);
  hipLaunchKernelGGL(row_sq_norm_kernel, dim3(cdiv(ns, 4)), dim3(256), 0, st, style, ns, ld, m.ss);
  hipLaunchKernelGGL(sk_fill_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, m.ones, n, 1.0f);
  LAUNCH_OK();
  CHK(st_remd_cost(pred, s.rp, m.sp, n, style, m.rs, m.ss, ns, ld, d, metric, s.Mt, m.S, ldm, st));      // pred-major
  CHK(sinkhorn_iterate(s, ns, n, l, T, loss_out, st));
  const int rows = round_up(n, 64);
  if (rows > n) CHK((int)hipMemsetAsync(m.W2 + (size_t)n * ldm, 0, sizeof(float) * (size_t)(rows - n) * ldm, st));
  hipLaunchKernelGGL(sk_assemble_metric_kernel, dim3(n), dim3(256), 0, st, s.Kt, s.Mt, m.S, ns, ldm, T, l, s.U, s.DA, s.V, s.DB,
                     n, m.rs, (float)d, metric, s.W, s.q, m.W2, m.q2);
  LAUNCH_OK();
  CHK(st_selfsim_bwd_gemm(m.W2, ldm, ldm, style, pred, m.ones, m.q2, n, ld, gscale, gpred, st));
  if (metric == STROTSS_METRIC_BOTH) return st_selfsim_bwd_gemm(s.W, ldm, ldm, style, pred, s.rp, s.q, n, ld, gscale, gpred, st);
  return 0;
}

int strotss_rows_gemm_bwd(const float* W, int ldw, int k, const float* B, const float* x, const float* r, const float* q,
                          int n, int ld, float g, float* dx, void* stream) {
  ST_CHECK_ARG(W && B && x && r && q && dx && n > 0 && k > 0 && k <= ldw && ld > 0, STROTSS_EINVAL);
  ST_CHECK_ARG(ldw % 32 == 0 && ld % 32 == 0, STROTSS_EALIGN);
  return st_selfsim_bwd_gemm(W, ldw, ldw, B, x, r, q, n, ld, g, dx, (hipStream_t)stream);
}

size_t strotss_remd_workspace_bytes(int ns, int n, int ld) {
  Workspace w = Workspace::planner();
  RemdWs s;
  s.plan(w, ns, n, ld);
  return w.off;
}

// cosine REMD after its prologue: rp / xp = reciprocal norms / x3 panels of pred, xs = x3 panels of style (NULL panels: the
// cost matrix on the f32 MFMA from the rows themselves)
static int remd_cos_core(const float* style, const float* rs, const __bf16* xs, int ns, const float* pred, const float* rp,
                         const __bf16* xp, int n, int ld, float gscale, float* gpred, float* loss_out, RemdWs& s,
                         hipStream_t st, int swapped = 0) {
  // pred-major cost matrix Ct[j][i] (bitwise the transpose of cosine_distance(style, pred): same products, same k
  // order), so that the backward kernel's scans of "column j" are contiguous: minima over i per prediction row j
  // are row minima (cmin), minima over j per style row i column minima (rmin)
  const int ldt = s.ldt;
  if (xp && xs) CHK(st_cosine_distance_x3(xp, rp, n, xs, rs, ns, ld, 0, s.C, ldt, 1, 0, 0, 0, st));
  else CHK(st_cosine_distance(pred, rp, n, style, rs, ns, ld, s.C, ldt, st));
  hipLaunchKernelGGL(row_col_min_kernel, dim3(n + cdiv(ns, 64) * COL_CHUNKS), dim3(256), 0, st, s.C, n, ns, ldt, s.cmin,
                     s.ccnt, s.pmin, s.pcnt);
  hipLaunchKernelGGL(col_min_final_select_kernel, dim3(1), dim3(1024), 0, st, s.pmin, s.pcnt, ns, ldt, s.rmin, s.rcnt,
                     s.cmin, n, 1, loss_out, s.sel, swapped);
  hipLaunchKernelGGL(remd_cos_bwd_kernel, dim3(n), dim3(256), 0, st, s.C, ldt, style, rs, ns, pred, rp, n,
                     ld, s.rmin, s.rcnt, s.cmin, s.ccnt, s.sel, gscale, gpred);
  ST_LAUNCH_RET();
}

int strotss_remd_cos_fwd_bwd(const float* style, const float* rs, int ns, const float* pred, int n, int d,
                             int ld, float gscale, float* gpred, float* loss_out, int flags, void* workspace,
                             size_t workspace_bytes, void* stream) {
  ST_CHECK_ARG(style && rs && pred && gpred && loss_out && workspace && ns > 0 && feat_ok(n, d, ld),
               STROTSS_EINVAL);
  ST_CHECK_ARG(ld % 32 == 0, STROTSS_EALIGN);
  ST_CHECK_ARG(ns <= REMD_MAX_LIST, STROTSS_ERANGE);
  Workspace w(workspace, workspace_bytes);
  RemdWs s;
  ST_CHECK_ARG(s.plan(w, ns, n, ld), STROTSS_EINVAL);
  hipStream_t st = (hipStream_t)stream;
  const bool x3 = cost_x3();
  if (x3) {
    hipLaunchKernelGGL(row_inv_norm_x3_kernel, dim3(cdiv(n > ns ? n : ns, 4), 2), dim3(256), 0, st, pred, n, ld, s.rp, s.xp,
                       style, ns, (float*)nullptr, s.xs);
  } else {
    hipLaunchKernelGGL(row_inv_norm_kernel, dim3(cdiv(n, 4)), dim3(256), 0, st, pred, n, ld, s.rp);
  }
  LAUNCH_OK();
  return remd_cos_core(style, rs, x3 ? s.xs : nullptr, ns, pred, s.rp, x3 ? s.xp : nullptr, n, ld, gscale, gpred, loss_out, s,
                       st, flags & STROTSS_REMD_SWAPPED);
}

int strotss_selfsim_pred_panels(void* workspace, size_t workspace_bytes, int n, int ld, const float** inv_norm,
                                const void** panels) {
  ST_CHECK_ARG(workspace && inv_norm && panels && n > 0 && ld > 0 && ld % 32 == 0, STROTSS_EINVAL);
  Workspace w(workspace, workspace_bytes);
  SelfsimWs s;
  ST_CHECK_ARG(s.plan(w, n, ld), STROTSS_EINVAL);
  *inv_norm = s.rp;
  *panels = cost_x3() ? (const void*)s.xp : nullptr;
  return 0;
}

int strotss_remd_cos_fwd_bwd_panels(const float* style, const float* rs, const void* style_panels, int ns, const float* pred,
                                    const float* pred_inv_norm, const void* pred_panels, int n, int d, int ld, float gscale,
                                    float* gpred, float* loss_out, void* workspace, size_t workspace_bytes, void* stream) {
  ST_CHECK_ARG(style && rs && style_panels && pred && pred_inv_norm && pred_panels && gpred && loss_out && workspace &&
               ns > 0 && feat_ok(n, d, ld), STROTSS_EINVAL);
  ST_CHECK_ARG(ld % 32 == 0, STROTSS_EALIGN);
  ST_CHECK_ARG(ns <= REMD_MAX_LIST, STROTSS_ERANGE);
  Workspace w(workspace, workspace_bytes);
  RemdWs s;
  ST_CHECK_ARG(s.plan(w, ns, n, ld), STROTSS_EINVAL);
  return remd_cos_core(style, rs, (const __bf16*)style_panels, ns, pred, pred_inv_norm, (const __bf16*)pred_panels, n, ld,
                       gscale, gpred, loss_out, s, (hipStream_t)stream);
}

size_t strotss_remd_metric_workspace_bytes(int ns, int n) {
  Workspace w = Workspace::planner();
  RemdGenWs s;
  s.plan(w, ns, n);
  return w.off;
}

int strotss_remd_metric_fwd_bwd(const float* style, int ns, const float* pred, int n, int d, int ld, int metric,
                                float gscale, float* gpred, float* loss_out, int flags, void* workspace,
                                size_t workspace_bytes, void* stream) {
  ST_CHECK_ARG(style && pred && gpred && loss_out && workspace && ns > 0 && feat_ok(n, d, ld), STROTSS_EINVAL);
  ST_CHECK_ARG(metric == STROTSS_METRIC_L2 || metric == STROTSS_METRIC_BOTH, STROTSS_EINVAL);
  ST_CHECK_ARG(ld % 32 == 0, STROTSS_EALIGN);
  ST_CHECK_ARG(ns <= REMD_MAX_LIST, STROTSS_ERANGE);
  Workspace w(workspace, workspace_bytes);
  RemdGenWs s;
  ST_CHECK_ARG(s.plan(w, ns, n), STROTSS_EINVAL);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(row_inv_norm_kernel, dim3(cdiv(n, 4)), dim3(256), 0, st, pred, n, ld, s.rp);
  hipLaunchKernelGGL(row_inv_norm_kernel, dim3(cdiv(ns, 4)), dim3(256), 0, st, style, ns, ld, s.rs);
  hipLaunchKernelGGL(row_sq_norm_kernel, dim3(cdiv(n, 4)), dim3(256), 0, st, pred, n, ld, s.sp);
  hipLaunchKernelGGL(row_sq_norm_kernel, dim3(cdiv(ns, 4)), dim3(256), 0, st, style, ns, ld, s.ss);
  LAUNCH_OK();
  // pred-major cost matrix Ct[j][i], as in strotss_remd_cos_fwd_bwd: row minima = minima over the style rows per
  // prediction (cmin, the R_Y side), column minima = minima over the predictions per style row (rmin, the R_X side)
  const int ldt = s.ldt;
  CHK(st_remd_cost(pred, s.rp, s.sp, n, style, s.rs, s.ss, ns, ld, d, metric, s.C, s.S, ldt, st));
  hipLaunchKernelGGL(row_col_min_kernel, dim3(n + cdiv(ns, 64) * COL_CHUNKS), dim3(256), 0, st, s.C, n, ns, ldt, s.cmin,
                     s.ccnt, s.pmin, s.pcnt);
  hipLaunchKernelGGL(col_min_final_select_kernel, dim3(1), dim3(1024), 0, st, s.pmin, s.pcnt, ns, ldt, s.rmin, s.rcnt,
                     s.cmin, n, 1, loss_out, s.sel, flags & STROTSS_REMD_SWAPPED);
  hipLaunchKernelGGL(remd_generic_bwd_kernel, dim3(n), dim3(256), 0, st, s.C, s.S, ldt, style, s.rs, ns, pred, s.rp, n, ld,
                     (float)d, metric, s.rmin, s.rcnt, s.cmin, s.ccnt, s.sel, gscale, gpred);
  ST_LAUNCH_RET();
}

int strotss_palette_remd_fwd_bwd(const float* style, int ns, const float* pred, int n, int ld, int rgb_to_yuv,
                                 float gscale, float* gpred, float* loss_out, int flags, void* workspace,
                                 size_t workspace_bytes, void* stream) {
  ST_CHECK_ARG(style && pred && gpred && loss_out && workspace && ns > 0 && n > 0 && ld >= 3, STROTSS_EINVAL);
  Workspace w(workspace, workspace_bytes);
  RemdWs s;
  ST_CHECK_ARG(s.plan(w, ns, n, 0), STROTSS_EINVAL);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(palette_prepare_kernel, dim3(cdiv(n > ns ? n : ns, 256), 2), dim3(256), 0, st, style, ns, ld, s.ys,
                     rgb_to_yuv, pred, n, s.yp);
  // no N_s x N matrix: the minima kernel and the backward kernel evaluate the 3-channel cost on the fly (bitwise alike)
  hipLaunchKernelGGL(palette_minima_kernel, dim3(ns + n), dim3(256), 0, st, s.ys, ns, s.yp, n, s.rmin, s.rcnt, s.cmin,
                     s.ccnt);
  hipLaunchKernelGGL(col_min_final_select_kernel, dim3(1), dim3(1024), 0, st, (const float*)nullptr, (const float*)nullptr,
                     n, 0, s.cmin, s.ccnt, s.rmin, ns, 0, loss_out, s.sel, flags & STROTSS_REMD_SWAPPED);
  hipLaunchKernelGGL(palette_bwd_kernel, dim3(n), dim3(64), 0, st, s.ys, ns, s.yp, n, s.rmin,
                     s.rcnt, s.cmin, s.ccnt, s.sel, gscale, gpred, ld, rgb_to_yuv);
  ST_LAUNCH_RET();
}

size_t strotss_moment_workspace_bytes(int n, int ld) {
  Workspace w = Workspace::planner();
  MomentWs s;
  s.plan(w, n, ld);
  return w.off;
}

int strotss_moment_stats(const float* x, int n, int d, int ld, float* mean_out, float* cov_out,
                         void* workspace, size_t workspace_bytes, void* stream) {
  ST_CHECK_ARG(x && mean_out && cov_out && workspace && feat_ok(n, d, ld), STROTSS_EINVAL);
  ST_CHECK_ARG(ld % 32 == 0, STROTSS_EALIGN);
  Workspace w(workspace, workspace_bytes);
  MomentWs s;
  ST_CHECK_ARG(s.plan(w, n, ld), STROTSS_EINVAL);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(col_sum_partial_kernel, dim3(cdiv(ld, 64), COL_CHUNKS), dim3(256), 0, st, x, n, ld, s.psum);
  if (moment_x3()) {      // same GEMM core as the prediction side (strotss_moment_fwd_bwd): identical statistics, bit for bit
    hipLaunchKernelGGL(center_x3_kernel, dim3(ld / 32, s.rows / 32), dim3(256), 0, st, x, n, s.rows, ld,
                       (const float*)nullptr, (__bf16*)nullptr, s.Pt, (const float*)s.psum, mean_out);   // (means: in here)
    LAUNCH_OK();
    return st_gram_tn_x3(s.Pt, s.rows, ld, 1.0f / (float)n, cov_out, st);
  }
  hipLaunchKernelGGL(col_mean_final_kernel, dim3(cdiv(ld, 256)), dim3(256), 0, st, s.psum, n, ld, mean_out);
  hipLaunchKernelGGL(center_kernel, dim3(min(2048, cdiv((size_t)s.rows * ld / 4, 256))), dim3(256), 0, st, x,
                     n, s.rows, ld, mean_out, s.cy);
  LAUNCH_OK();
  return st_gram_tn(s.cy, s.rows, ld, 1.0f / (float)n, cov_out, st);
}

int strotss_moment_fwd_bwd(const float* style_mean, const float* style_cov, const float* pred, int n, int d,
                           int ld, float gscale, float* gpred, float* loss_out, void* workspace,
                           size_t workspace_bytes, void* stream) {
  ST_CHECK_ARG(style_mean && style_cov && pred && gpred && loss_out && workspace && feat_ok(n, d, ld),
               STROTSS_EINVAL);
  ST_CHECK_ARG(ld % 32 == 0, STROTSS_EALIGN);
  Workspace w(workspace, workspace_bytes);
  MomentWs s;
  ST_CHECK_ARG(s.plan(w, n, ld), STROTSS_EINVAL);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(col_sum_partial_kernel, dim3(cdiv(ld, 64), COL_CHUNKS), dim3(256), 0, st, pred, n, ld, s.psum);
  const bool x3 = moment_x3();
  int n_partial = 0;
  if (x3) {           // both GEMMs on the bf16x3 core (csrc/mfma_x3.h); the column means are finished inside the centring
    hipLaunchKernelGGL(center_x3_kernel, dim3(ld / 32, s.rows / 32), dim3(256), 0, st, pred, n, s.rows, ld,
                       (const float*)nullptr, s.Pc, s.Pt, (const float*)s.psum, s.mean);
    LAUNCH_OK();
    CHK(st_moment_fwd_x3(s.Pt, s.rows, ld, style_cov, s.Tp, 1.0f / (float)n, s.partial, &n_partial, st));
  } else {
    hipLaunchKernelGGL(col_mean_final_kernel, dim3(cdiv(ld, 256)), dim3(256), 0, st, s.psum, n, ld, s.mean);
    hipLaunchKernelGGL(center_kernel, dim3(min(2048, cdiv((size_t)s.rows * ld / 4, 256))), dim3(256), 0, st,
                       pred, n, s.rows, ld, s.mean, s.cy);
    LAUNCH_OK();
    CHK(st_moment_fwd_gemm(s.cy, s.rows, ld, style_cov, s.T, 1.0f / (float)n, s.partial, &n_partial, st));
  }
  hipLaunchKernelGGL(moment_finalize_kernel, dim3(1), dim3(256), 0, st, s.partial, n_partial, style_mean,
                     s.mean, d, ld, s.sgn, loss_out);
  LAUNCH_OK();
  // dL/dcy = cy (T + T^T) / (n d^2) = 2 cy T / (n d^2) (T symmetric); centring adjoint drops the
  // column mean of dL/dcy, which is zero up to rounding because sum_i cy[i,:] = 0.
  const float a = gscale * 2.0f / ((float)n * (float)d * (float)d);
  const float b = gscale / ((float)d * (float)n);
  if (x3) return st_moment_bwd_x3(s.Pc, n, ld, s.Tp, a, s.sgn, b, gpred, st);
  return st_moment_bwd_gemm(s.cy, n, ld, s.T, a, s.sgn, b, gpred, st);
}

// ---------------------------------------------------------------------------------------------------------------------
// The four loss terms of one train step (run_strotss.py:131-142: self_similarity + moment_matching + relaxed_emd + the YUV
// palette relaxed_emd on the same prediction rows) in ONE call whose three forward GEMMs share ONE launch (st_loss_forward_group_x3).  Every
// kernel, tile shape and summation order is that of the three separate entry points called in this order
// (strotss_selfsim_fwd_bwd, strotss_moment_fwd_bwd, strotss_remd_cos_fwd_bwd_panels, strotss_palette_remd_fwd_bwd): bit for bit the
// same losses and gradient rows; 13 launches instead of 21 (one prologue launch, one for row statistics + moment scalars).  bf16x3 core only (STROTSS_X3 / _COST /
// _MOMENT = 0: STROTSS_EINVAL, the caller takes the separate entry points).
size_t strotss_step_losses_workspace_bytes(int ns, int n, int ld) {
  Workspace w = Workspace::planner();
  SelfsimWs a; MomentWs b; RemdWs c, p;
  a.plan(w, n, ld); b.plan(w, n, ld); c.plan(w, ns, n, ld); p.plan(w, ns, n, 0);
  return w.off;
}

int strotss_step_losses_fwd_bwd(const float* pred, const float* content, int n, int d, int ld, const float* style,
                                const float* style_inv_norm, const void* style_panels, int ns, const float* style_mean,
                                const float* style_cov, float g_content, float g_moment, float g_remd, float g_palette,
                                float* gpred, float* loss_content, float* loss_moment, float* loss_remd, float* loss_palette,
                                void* workspace, size_t workspace_bytes, void* stream) {
  ST_CHECK_ARG(pred && content && style && style_inv_norm && style_panels && style_mean && style_cov && gpred && loss_content &&
               loss_moment && loss_remd && loss_palette && workspace && ns > 0 && feat_ok(n, d, ld), STROTSS_EINVAL);
  ST_CHECK_ARG(ld % 32 == 0, STROTSS_EALIGN);
  ST_CHECK_ARG(ns <= REMD_MAX_LIST, STROTSS_ERANGE);
  ST_CHECK_ARG(cost_x3() && moment_x3(), STROTSS_EINVAL);
  Workspace w(workspace, workspace_bytes);
  SelfsimWs s; MomentWs m; RemdWs r, pl;
  ST_CHECK_ARG(s.plan(w, n, ld) && m.plan(w, n, ld) && r.plan(w, ns, n, ld) && pl.plan(w, ns, n, 0), STROTSS_EINVAL);
  hipStream_t st = (hipStream_t)stream;
  const int ldc = s.ldc;
  // ---- ONE prologue launch: norms + panels of prediction and content rows | column partial sums | YUV rows | transposed panels
  StepPrologueArgs pa{pred, content, n, ld, s.rp, s.xp, s.rc, s.xc, m.psum, style, ns, pl.ys, pl.yp, 1, ldc, s.xt,
                      2 * cdiv(n, 4), cdiv(ld, 64) * COL_CHUNKS, 2 * cdiv(n > ns ? n : ns, 256)};
  hipLaunchKernelGGL(step_losses_prologue_kernel, dim3((unsigned)(pa.nA + pa.nB + pa.nC + (ld / 32) * (ldc / 32))), dim3(256), 0, st,
                     pa);
  // the centred rows (both ways round) need the column sums: their own launch
  hipLaunchKernelGGL(center_x3_kernel, dim3(ld / 32, m.rows / 32), dim3(256), 0, st, pred, n, m.rows, ld,
                     (const float*)nullptr, m.Pc, m.Pt, (const float*)m.psum, m.mean);
  LAUNCH_OK();
  // ---- the three forward products, one launch
  int n_partial = 0;
  CHK(st_loss_forward_group_x3(m.Pt, m.rows, ld, style_cov, m.Tp, 1.0f / (float)n, m.partial, &n_partial,
                               s.xp, s.rp, n, s.xc - s.xp, s.rc - s.rp, s.Dx, ldc, s.Dy - s.Dx,
                               style_panels, style_inv_norm, ns, r.C, r.ldt, st));
  // ---- every statistic of the forward products: self-similarity rows | moment scalars + mean-gradient signs | REMD minima |
  // palette minima (run_strotss.py:37-39, on the YUV rows the prologue made) -- one launch
  const int ldt = r.ldt;
  StepStatsArgs sa{s.Dx, s.Dy, n, ldc, 1.0f / (float)n, s.sx, s.sy, s.tt, s.lossrow,
                   m.partial, n_partial, style_mean, m.mean, d, ld, m.sgn, loss_moment,
                   r.C, ns, ldt, r.cmin, r.ccnt, r.pmin, r.pcnt,
                   pl.ys, pl.yp, pl.rmin, pl.rcnt, pl.cmin, pl.ccnt};
  hipLaunchKernelGGL(step_losses_stats_kernel, dim3((unsigned)(n + 1 + n + cdiv(ns, 64) * COL_CHUNKS + ns + n)), dim3(256), 0, st, sa);
  // ---- the two branch selections (+ stage 2 of the REMD column minima) | the self-similarity gradient matrix: one launch
  hipLaunchKernelGGL(step_losses_select_sym_kernel, dim3((unsigned)(2 + cdiv(n, 4))), dim3(1024), 0, st,
                     SelectArgs{r.pmin, r.pcnt, ns, ldt, r.rmin, r.rcnt, r.cmin, n, 1, loss_remd, r.sel, 0},
                     SelectArgs{nullptr, nullptr, n, 0, pl.cmin, pl.ccnt, pl.rmin, ns, 0, loss_palette, pl.sel, 0},
                     SelfsimSymArgs{s.Dx, s.Dy, s.sx, s.sy, s.tt, s.rp, n, ldc, ldc, 1.0f / (float)n, s.Mq, s.qdot, s.mp,
                                    s.lossrow, 1.0f / (float)n, loss_content});
  LAUNCH_OK();
  CHK(st_selfsim_bwd_x3(s.mp, ldc, s.xt, pred, s.rp, s.qdot, n, ld, g_content, gpred, st));
  CHK(st_moment_bwd_x3(m.Pc, n, ld, m.Tp, g_moment * 2.0f / ((float)n * (float)d * (float)d), m.sgn,
                       g_moment / ((float)d * (float)n), gpred, st));
  // ---- relaxed EMD and palette: sparse backward kernels
  hipLaunchKernelGGL(remd_cos_bwd_kernel, dim3(n), dim3(256), 0, st, r.C, ldt, style, style_inv_norm, ns, pred, s.rp, n,
                     ld, r.rmin, r.rcnt, r.cmin, r.ccnt, r.sel, g_remd, gpred);
  hipLaunchKernelGGL(palette_bwd_kernel, dim3(n), dim3(64), 0, st, pl.ys, ns, pl.yp, n, pl.rmin, pl.rcnt, pl.cmin, pl.ccnt,
                     pl.sel, g_palette, gpred, ld, 1);
  ST_LAUNCH_RET();
}

}  // extern "C"
