// F(4x4, 3x3) Winograd layers with 256 output channels and more (nn/model.py:44-48 of the reference: the VGG trunk's
// block3 .. block5 convolutions), two-kernel form: the input transform writes V ONCE as x3 panels (winograd.hip:
// winograd43_in_x3_kernel), and THIS kernel does the 36 transform-domain GEMMs AND the output transform -- the
// transform-domain products M never exist in memory (the three-kernel form writes and re-reads 2.25 x the activations
// for them, and its output transform is a third, HBM-bound launch).
//
// Work item = 64 Winograd tiles x 64 output channels for ALL 36 positions, one 256-thread workgroup (2 x 2 waves, a
// 32 x 32 MFMA tile each) per CU.  The positions run one after the other through ONE continuous K pipeline: the x3
// panels of consecutive positions are contiguous in memory (mfma_x3.h: (kb, plane, row, k)), so the whole sweep
// p = 0..35, kb = 0..K/32-1 is a linear walk of both operands; operands go global -> LDS by LDS-DMA into a ring of NS
// stages (one K-block of 32 per stage: 6 plane images of [64 rows][64 B], XOR-swizzled as in mfma_x3.h), NS - 1 tiles
// in flight, one raw barrier per K-step, counted vmcnt.  The bf16x3 products are those of mfma_x3.h (six exact
// partial products per f32 product on v_mfma_f32_32x32x16_bf16, f32 accumulation).
//
// Output transform in registers, separably:  Y = A^T M A  with M[r][q] the finished accumulator of position 6 r + q:
//     after position (r, q):   Z[j]    += A^T[j][q] * M[r][q]      (j = 0..3;  Z: 4 accumulator tiles)
//     after row r (q == 5):    Y[i][j] += A^T[i][r] * Z[j]         (i = 0..3;  Y: 16 accumulator tiles, mostly AGPRs)
// A^T = [[1,1,1,1,1,0],[0,1,-1,2,-2,0],[0,1,1,4,4,0],[0,1,-1,8,-8,1]].  A lane then owns, for each of its 16 tiles and
// its output channel, the 4 x 4 output pixels: bias + ReLU (forward) or the ReLU mask of the layer's input (data
// gradient) are applied on the way out, 32 consecutive channels (128 B) per half wave and store instruction.
//
// Algorithmic bytes per launch: input X (in-transform) + V written and read once (3.375 X as x3 panels) + output
// (+ mask): 8.75 X against the three-kernel form's 13.25 X; HBM-bound part left: the in-transform.
#include <stdlib.h>

#include "internal.h"
#include "mfma_x3.h"

namespace {

constexpr int GO_PL = 64 * 64;           // bytes of one plane image ([64 rows][64 B])
constexpr int GO_STAGE = 6 * GO_PL;      // 24 KiB: A h, m, l; B h, m, l

// column r of A^T: the four coefficients applied to transform-domain index r
__device__ __forceinline__ void at_column(int r, float (&c)[4]) {
  const float b = r == 1 ? 1.f : r == 2 ? -1.f : r == 3 ? 2.f : -2.f;
  const bool first = r == 0, last = r == 5, edge = first || last;
  c[0] = last ? 0.f : 1.f;
  c[1] = edge ? 0.f : b;
  c[2] = edge ? 0.f : b * b;
  c[3] = first ? 0.f : last ? 1.f : b * b * b;
}

// one x3 panel operand as seen by a wave's DMA lanes (16 rows x 64 B per wave instruction), walking the K-blocks of
// all 36 positions linearly; past the last K-block the walk stays on it (the pipeline's look-ahead loads stay in bounds)
struct GoOperand {
  const char* base;
  unsigned off, plane, cur, last;
  __device__ __forceinline__ GoOperand(const __bf16* p, int rows, int row0, unsigned total_steps) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    base = reinterpret_cast<const char*>(p);
    plane = (unsigned)rows * 64u;
    cur = 0;
    last = (total_steps - 1) * 3u * plane;
    const int chunk = (lane & 3) ^ ((lane >> 4) & 3);           // source-side swizzle, see mfma_x3.h
    const int row = min(row0 + wave * 16 + (lane >> 2), rows - 1);
    off = (unsigned)row * 64u + (unsigned)chunk * 16u;
  }
  __device__ __forceinline__ void dma(int pl, unsigned char* lds_dst) const {
    const char* src = base + (size_t)(cur + pl * plane) + off;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
  }
  __device__ __forceinline__ void advance() { cur = min(cur + 3u * plane, last); }
};

#define GO_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
template <int N> __device__ __forceinline__ void go_wait_vm() {
  static_assert(N == 0 || N == 6 || N == 12 || N == 18 || N == 24, "6 DMA pieces per tile");
  if constexpr (N == 0) GO_WAIT_VM(0); else if constexpr (N == 6) GO_WAIT_VM(6); else if constexpr (N == 12) GO_WAIT_VM(12);
  else if constexpr (N == 18) GO_WAIT_VM(18); else GO_WAIT_VM(24);
}

// V: x3 panels of the transformed input, 36 x (T rows x K);  U: x3 panels of the weights, 36 x (N rows x K);
// out / mask: (H, W, N) NHWC.  !MASK: out = relu ? max(Y + bias, 0) : Y + bias;  MASK: out = mask > 0 ? Y : 0.
template <bool MASK, int NS>
__global__ __launch_bounds__(256) void winograd43_gemm_out_kernel(const __bf16* __restrict__ V, const __bf16* __restrict__ U,
                                                                  int T, int K, int N, int H, int W, int TW,
                                                                  const float* __restrict__ bias,
                                                                  const float* __restrict__ mask, int relu,
                                                                  const unsigned* __restrict__ bits_in,
                                                                  unsigned* __restrict__ bits_out,
                                                                  float* __restrict__ out, int nitems, int nb) {
  __shared__ __attribute__((aligned(1024))) unsigned char lds[NS * GO_STAGE];
  constexpr int D = NS - 1;                                      // tiles in flight
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, hh = lane >> 5;
  const int KS = K >> 5;
  const unsigned total_steps = 36u * (unsigned)KS;
  const int f = (l31 >> 2) & 3;
  const int a_rd = (wm * 32 + l31) * 64 + ((hh ^ f) << 4);       // k-chunk 0; chunk 1 is this ^ 32
  const int b_rd = 3 * GO_PL + (wn * 32 + l31) * 64 + ((hh ^ f) << 4);
  constexpr int PA[6] = {2, 0, 1, 1, 0, 0};                      // partial products, smallest terms first
  constexpr int PB[6] = {0, 2, 1, 0, 1, 0};

  for (int it = blockIdx.x; it < nitems; it += gridDim.x) {
    // XCD-aware order inside a round: the cout blocks of one tile block run on one XCD and share V in its L2
    const int round0 = it - (int)blockIdx.x;
    const int in_round = min((int)gridDim.x, nitems - round0);
    const int item = round0 + ((in_round & 7) == 0 ? (int)xcd_swizzle(blockIdx.x, in_round) : (int)blockIdx.x);
    const int m0 = (item / nb) * 64, n0 = (item % nb) * 64;
    GoOperand oa(V, T, m0, total_steps), ob(U, N, n0, total_steps);
    auto dma_piece = [&](int j, unsigned char* stage) {           // j 0..2: A planes, 3..5: B planes; 16 rows per wave
#ifdef GO_ABL_NO_DMA                                            // timing ablations (tools/go_ablate.sh): results are wrong
      return;
#endif
      unsigned char* dst = stage + j * GO_PL + wave * 1024;
      if (j < 3) oa.dma(j, dst); else ob.dma(j - 3, dst);
    };
    bf16x8 fa[2][3], fb[2][3];
    auto read_frags = [&](const unsigned char* stage, int kc, int slot) {
      const unsigned char* pa = stage + (a_rd ^ (kc << 5));
      const unsigned char* pb = stage + (b_rd ^ (kc << 5));
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        fa[slot][p] = *reinterpret_cast<const bf16x8*>(pa + p * GO_PL);
        fb[slot][p] = *reinterpret_cast<const bf16x8*>(pb + p * GO_PL);
      }
    };
    f32x16 acc, Z[4], Y[4][4];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      acc[e] = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        Z[j][e] = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) Y[i][j][e] = 0.f;
      }
    }
    // prologue: tiles 0 .. D-1 into stages 0 .. D-1
#pragma unroll
    for (int d = 0; d < D; ++d) {
#pragma unroll
      for (int j = 0; j < 6; ++j) dma_piece(j, lds + d * GO_STAGE);
      oa.advance(); ob.advance();
    }
    go_wait_vm<6 * (D - 1)>();
    __builtin_amdgcn_s_barrier();
    int cur = 0;                                                   // stage of the current tile
    read_frags(lds, 0, 0);

    for (int p = 0; p < 36; ++p) {
      for (int kb = 0; kb < KS; ++kb) {
        unsigned char* s_cur = lds + cur * GO_STAGE;
        const int prev = cur == 0 ? NS - 1 : cur - 1;              // free since the barrier of the previous step
        unsigned char* s_new = lds + prev * GO_STAGE;
        const int nxt = cur == NS - 1 ? 0 : cur + 1;
        unsigned char* s_nxt = lds + nxt * GO_STAGE;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
#pragma unroll
          for (int q = 0; q < 6; ++q) {
            if (c == 0 && q == 1) {                                // second k-chunk's fragments behind the first product
              __builtin_amdgcn_sched_barrier(0);
              read_frags(s_cur, 1, 1);
              __builtin_amdgcn_sched_barrier(0);
            }
#ifndef GO_ABL_NO_MFMA
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[c][PA[q]], fb[c][PB[q]], acc, 0, 0, 0);
#else
            acc[0] += (float)fa[c][PA[q]][0] + (float)fb[c][PB[q]][0];
#endif
            if (c == 0) {                                          // tile s + D: one DMA piece behind each MFMA
              __builtin_amdgcn_sched_barrier(0);
              dma_piece(q, s_new);
              __builtin_amdgcn_sched_barrier(0);
            }
            if (c == 1 && q == 0) {
              __builtin_amdgcn_sched_barrier(0);
              go_wait_vm<6 * (D - 1)>();                           // tile s + 1 has landed, D - 1 tiles stay in flight
              asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
              __builtin_amdgcn_s_barrier();
              asm volatile("" ::: "memory");
              read_frags(s_nxt, 0, 0);
              __builtin_amdgcn_sched_barrier(0);
            }
          }
        }
        oa.advance(); ob.advance();
        cur = nxt;
      }
      // position (r, q) finished: fold its accumulator into the column stage, after a row into the output tiles
#ifdef GO_ABL_NO_FOLD
      if (p < 35) continue;
#endif
      const int r = p / 6, q = p - 6 * r;
      float cq[4];
      at_column(q, cq);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) Z[j][e] = fmaf(cq[j], acc[e], Z[j][e]);
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] = 0.f;
      if (q == 5) {
        float cr[4];
        at_column(r, cr);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) Y[i][j][e] = fmaf(cr[i], Z[j][e], Y[i][j][e]);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e) Z[j][e] = 0.f;
      }
    }
    GO_WAIT_VM(0);                                                 // the look-ahead tiles past the end
    // epilogue: lane = (16 tiles, one output channel); 32 consecutive channels per half wave and store
    const int n = n0 + wn * 32 + l31;
    const float bv = (MASK || !bias) ? 0.f : bias[n];                // data-gradient without a mask: no bias either
#pragma unroll
    for (int e = 0; e < 16; ++e) {
#ifdef GO_ABL_NO_EPI
      if (e > 0) continue;
#endif
      const int tile = m0 + wm * 32 + 8 * (e >> 2) + 4 * hh + (e & 3);
      if (tile >= T) continue;
      const int ty = tile / TW, tx = tile - ty * TW;
      // sign words of the tile grid (include/strotss_hip.h: relu_bits): one per (tile, channel), byte i = row i
      const unsigned kw = (MASK && bits_in) ? bits_in[(size_t)tile * N + n] : 0u;
      unsigned ow = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int y = 4 * ty + i;
        if (y >= H) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int x = 4 * tx + j;
          if (x >= W) continue;
          const size_t o = ((size_t)y * W + x) * N + n;
          float v = Y[i][j][e];
          if (MASK) v = (bits_in ? ((kw >> (8 * i + j)) & 1u) != 0u : mask[o] > 0.f) ? v : 0.f;
          else { v += bv; ow |= (v > 0.f ? 1u : 0u) << (8 * i + j); if (relu) v = fmaxf(v, 0.f); }
          out[o] = v;
        }
      }
      if (!MASK && bits_out) bits_out[(size_t)tile * N + n] = ow;
    }
    __builtin_amdgcn_s_barrier();                                  // every wave is done with the ring before it is refilled
  }
}

int go_cus() {
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
    cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  return cus;
}

}  // namespace

// 0: never; 1 (default): where a launch has at least STROTSS_WINO_GEMM_OUT_MIN_ITEMS work items (default 192 = three
// quarters of the CUs -- every item is 36 GEMMs of 64 x 64 x K, so fewer items leave CUs idle for the whole launch);
// 2: wherever the shapes allow (tests).
bool st_winograd43_gemm_out_enabled(size_t T, int cin, int cout) {
  static int mode = -1, min_items = 192;
  if (mode < 0) {
    const char* e = getenv("STROTSS_WINO_GEMM_OUT"); mode = e ? atoi(e) : 0;
    const char* m = getenv("STROTSS_WINO_GEMM_OUT_MIN_ITEMS"); if (m) min_items = atoi(m);
  }
  if (mode == 0 || cin % 32 != 0 || cout % 64 != 0) return false;
  if ((size_t)36 * 3 * T * cin * 2 >= ((size_t)1 << 32) || (size_t)36 * 3 * cout * cin * 2 >= ((size_t)1 << 32)) return false;
  const size_t items = ((T + 63) / 64) * (size_t)(cout / 64);
  return mode == 2 || items >= (size_t)min_items;
}

int st_winograd43_gemm_out(const void* V, const void* Ux3, size_t T, int cin, int cout, int h, int w, int TW,
                           const float* bias, const float* mask, int relu, float* out, const unsigned* bits_in,
                           unsigned* bits_out, hipStream_t st) {
  const int nb = cout / 64;
  const size_t items = ((T + 63) / 64) * (size_t)nb;
  if (items >= ((size_t)1 << 30)) return STROTSS_ERANGE;
  const int nitems = (int)items;
  int grid = go_cus();
  if (grid > nitems) grid = nitems;
  const __bf16* v = reinterpret_cast<const __bf16*>(V);
  const __bf16* u = reinterpret_cast<const __bf16*>(Ux3);
  if (mask || bits_in)
    hipLaunchKernelGGL((winograd43_gemm_out_kernel<true, 5>), dim3((unsigned)grid), dim3(256), 0, st, v, u, (int)T, cin, cout,
                       h, w, TW, bias, mask, relu, bits_in, bits_out, out, nitems, nb);
  else
    hipLaunchKernelGGL((winograd43_gemm_out_kernel<false, 5>), dim3((unsigned)grid), dim3(256), 0, st, v, u, (int)T, cin, cout,
                       h, w, TW, bias, mask, relu, bits_in, bits_out, out, nitems, nb);
  ST_LAUNCH_RET();
}
