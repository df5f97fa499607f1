// Internal (non-ABI) launchers shared between the translation units of libstrotss_hip.so.
#pragma once
#include "common.h"

// gemm.hip
int st_cosine_distance(const float* x, const float* rx, int nx, const float* y, const float* ry, int ny,
                       int ld, float* C, int ldc, hipStream_t s);
int st_l2_distance(const float* x, const float* xs, int nx, const float* y, const float* ys, int ny, int ld, int d,
                   float* C, int ldc, hipStream_t s);
int st_remd_cost(const float* x, const float* rx, const float* sx, int nx, const float* y, const float* ry, const float* sy,
                 int ny, int ld, int d, int metric, float* C, float* S, int ldc, hipStream_t s);
int st_gram_tn(const float* A, int krows, int ld, float alpha, float* C, hipStream_t s);
int st_moment_fwd_gemm(const float* cy, int krows, int ld, const float* Sx, float* T, float inv_n,
                       float* partial, int* n_partial, hipStream_t s);
int st_moment_bwd_gemm(const float* cy, int n, int ld, const float* T, float alpha, const float* bias,
                       float bias_scale, float* dY, hipStream_t s);
int st_selfsim_bwd_gemm(const float* Mq, int ldm, int kpad, const float* bmat, const float* x, const float* r,
                        const float* q, int n, int ld, float g, float* dx, hipStream_t s);

int st_gemm_nt_batched(const float* A, int lda, long long strideA, const float* B, int ldb, long long strideB,
                       float* C, int ldc, long long strideC, int M, int N, int K, int batch, hipStream_t s);

// winograd_fused.hip: F(4x4,3x3) in one kernel (cin % 16 == 0, cout % 32 == 0)
bool st_winograd43_fused_enabled(int h, int w, int cout);
int st_winograd43_pack(const float* u_prk, int rows, int k, float* u_packed, hipStream_t st);
int st_winograd43_fused(const float* in, int h, int w, int cin, const float* U, const float* bias, int cout,
                        const float* mask, int relu, float* out, float* pool_out, unsigned char* pool_code,
                        const unsigned* bits_in, unsigned* bits_out, hipStream_t st, int accumulate = 0);
int st_maxpool2_fwd(const float* in, int h, int w, int c, float* out, unsigned char* code, hipStream_t st);

// f32 GEMM cores on the bf16 MFMA (mfma_x3.h): operands as "x3 panels" (3 * rows * K bf16 per batch entry)
int st_x3_split_rows(const float* x, int rows, int ld, int K, long long stride_in, void* panels, int batch,
                     hipStream_t s);
int st_cosine_distance_x3(const void* xp, const float* rx, int nx, const void* yp, const float* ry, int ny, int K,
                          int symm, float* C, int ldc, int batch, long long pstride, long long rstride,
                          long long cstride, hipStream_t s);
int st_selfsim_bwd_x3(const void* Mp, int kpad, const void* Xt, const float* x, const float* r, const float* q, int n,
                      int ld, float g, float* dx, hipStream_t s);
int st_gram_tn_x3(const void* Pt, int npad, int ld, float alpha, float* C, hipStream_t s);
int st_moment_fwd_x3(const void* Pt, int npad, int ld, const float* Sx, void* Tp, float inv_n, float* partial,
                     int* n_partial, hipStream_t s);
int st_loss_forward_group_x3(const void* Pt, int npad, int ld, const float* Sx, void* Tp, float inv_n, float* partial,
                             int* n_partial, const void* xp, const float* rp, int n, long long pstride, long long rstride,
                             float* Dx, int ldc, long long dstride, const void* xs, const float* rs, int ns, float* Ct,
                             int ldt, hipStream_t s);
int st_moment_bwd_x3(const void* Pc, int n, int ld, const void* Tp, float alpha, const float* bias, float bias_scale,
                     float* dY, hipStream_t s);
int st_gemm_x3_batched(const void* A, const void* B, float* C, int ldc, long long strideC, int M, int N, int K,
                       int batch, hipStream_t s, long min_tiles128 = 0);

static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

// Bump allocator over a caller-provided workspace (256-byte aligned slices).
struct Workspace {
  char* base; size_t size; size_t off;
  Workspace(void* p, size_t n) : base((char*)p), size(n), off(0) {}
  template <class T> T* take(size_t count) {
    size_t bytes = (count * sizeof(T) + 255) & ~(size_t)255;
    if (off + bytes > size) { failed = true; return nullptr; }
    T* r = reinterpret_cast<T*>(base + off);  // base == nullptr: planning pass, never dereferenced
    off += bytes;
    return r;
  }
  bool failed = false;
  bool ok() const { return !failed; }
  static Workspace planner() { return Workspace(nullptr, (size_t)1 << 60); }
};
static inline size_t ws_slice(size_t count, size_t elt) { return (count * elt + 255) & ~(size_t)255; }
