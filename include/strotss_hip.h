/*
 * strotss_hip.h -- C ABI of libstrotss_hip.so: the MI355X (gfx950) kernels behind the
 * STROTSS optimisation inner loop of interaction-lab-uh/STROTSS-tensorflow.
 *
 * The reference has NO native boundary: its hot path is a chain of TensorFlow op calls made
 * from Python (run_strotss.py:131-148).  Each entry point below replaces the TF op(s) at the
 * cited reference call site; the Python package strotss-tensorflow_amd/nn (same module and
 * function names as the reference's nn/) binds them with ctypes (nn/_hip.py).
 *
 * Conventions (all entry points):
 *   - every pointer is a CALLER-OWNED DEVICE pointer (float32 unless stated), never freed or
 *     allocated by the library; workspaces are passed in explicitly;
 *   - `stream` is the hipStream_t the work is enqueued on (as a void*); no call synchronises;
 *   - return 0 on success, a negative STROTSS_E* for a bad argument (nothing is launched), or a
 *     positive hipError_t from the launch;
 *   - images / feature maps are NHWC with batch 1: (H, W, C) row-major, C contiguous;
 *   - sampled feature matrices are (rows, ld) row-major with ld >= D, ld % 32 == 0, columns
 *     [D, ld) and rows [n, rows) ZERO (the library keeps them zero); rows % 32 == 0;
 *   - n-by-n cost matrices are (rows, ldc) row-major, ldc % 4 == 0.
 */
#ifndef STROTSS_HIP_H
#define STROTSS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define STROTSS_OK 0
#define STROTSS_EINVAL (-1)   /* bad size / null pointer */
#define STROTSS_EALIGN (-2)   /* leading dimension or K not a multiple of what the kernel needs */
#define STROTSS_ERANGE (-3)   /* too many maps / tensors for the fixed-size descriptor */

#define STROTSS_MAX_MAPS 12
#define STROTSS_MAX_DIVS 8
#define STROTSS_MAX_TENSORS 8

/* library identity; also used by the loader to check the build. */
int strotss_abi_version(void);
const char* strotss_build_info(void);

/* ---------------------------------------------------------------------------------------
 * Images: tf.image.resize(bilinear, half-pixel centres, no antialias)
 *   replaces nn/utils.py:37,41 and nn/strotss_utils.py:142-143,162 (tf.image.resize)
 * --------------------------------------------------------------------------------------- */
/* out(oh,ow,c) = alpha*resize(in(ih,iw,c)) + (add ? add(oh,ow,c) : 0) ; alpha is +1 or -1 in
 * practice: fold uses (add = residual, alpha = 1); make_laplacian uses (add = x, alpha = -1). */
int strotss_resize_bilinear(const float* in, int ih, int iw, int c, float* out, int oh, int ow,
                            float alpha, const float* add, void* stream);
/* fold_laplacian_pyramid (strotss_utils.py:159-163) of a 3-channel pyramid in ONE launch:
 *   img = var[0] + up(var[1] + up(var[2] + ... up(var[n-1]))),   up = TF2 bilinear resize to the next finer level's size.
 * Level k is (h[k], w[k], 3); every level at most as large as the one above it, halving or faster from level 1 on
 * (what make_laplacian_pyramid builds); STROTSS_ERANGE otherwise (use strotss_resize_bilinear level by level).
 * Same taps and arithmetic as n-1 calls of strotss_resize_bilinear(..., alpha = 1, add = var[k]). */
#define STROTSS_MAX_LEVELS 8
typedef struct {
  int n_levels;
  int h[STROTSS_MAX_LEVELS], w[STROTSS_MAX_LEVELS];
  const float* var[STROTSS_MAX_LEVELS];
} strotss_pyramid_t;
int strotss_fold_pyramid(const strotss_pyramid_t* pyr, float* img, void* stream);
/* adjoint of the above w.r.t. `in`: gin(ih,iw,c) = resize^T(gout(oh,ow,c)).  Deterministic
 * gather form (no atomics).  Replaces the TF gradient of strotss_utils.py:162. */
int strotss_resize_bilinear_adjoint(const float* gout, int oh, int ow, int c, float* gin, int ih,
                                    int iw, void* stream);
/* The adjoint of the whole fold (3 channels): var[0] = the gradient of the folded image (input), var[k] for k >= 1 receive
 * var[k] = resize^T(var[k-1]) -- the gradients of the pyramid levels (written through the const pointers of the
 * descriptor).  Two levels per launch where each level halves the one above it to within a pixel (a workgroup recomputes
 * the few middle-level rows its tile reads), strotss_resize_bilinear_adjoint level by level otherwise; bit for bit the
 * same values either way. */
int strotss_fold_pyramid_adjoint(const strotss_pyramid_t* gpyr, void* stream);

/* ---------------------------------------------------------------------------------------
 * VGG16 trunk (frozen): nn/model.py:44-55 -- Keras Conv2D(3x3,'same',relu) / MaxPooling2D(2,2)
 * --------------------------------------------------------------------------------------- */
/* First layer with the ImageNet preprocess fused (model.py:50-51):
 *   out(h,w,cout) = relu(conv3x3((img-mean)/std zero-padded, w_kio) + bias), img(h,w,3) in [0,1].
 *   w_kio: (27, cout) = HWIO kernel flattened, k = (dy*3+dx)*3+ci.
 *   mean3/std3 are the ONLY host pointers of this ABI: 3 floats each, the constants of
 *   model.py:34-35, read at call time and passed to the kernel by value.
 *   relu_bits_out (may be NULL): also writes the sign words of `out`, see strotss_relu_bits. */
int strotss_conv3x3_c3_fwd(const float* img, int h, int w, const float* w_kio, const float* bias,
                           int cout, const float* mean3, const float* std3, float* out,
                           unsigned int* relu_bits_out, void* stream);
/* Sign words of a conv layer's output on the F(4x4,3x3) tile grid: what the data-gradient of the NEXT layer needs of it
 * (the ReLU mask, model.py:44-48 differentiated) in 4 bytes per (4 x 4 tile, channel) instead of 64:
 *   relu_bits[(ty * TW + tx) * c + ch], TW = (w + 3) / 4, byte r (0..3), bit q (0..3) = act[4 ty + r][4 tx + q][ch] > 0
 * (pixels outside the image: unspecified).  strotss_relu_bits_bytes(h, w, c) bytes.  The forward entry points write
 * them from registers (relu_bits_out); strotss_relu_bits derives them from a finished activation tensor (h, w, c). */
size_t strotss_relu_bits_bytes(int h, int w, int c);
int strotss_relu_bits(const float* act, int h, int w, int c, unsigned int* relu_bits, void* stream);
/* Generic layer, cin % 32 == 0, cout % 64 == 0:
 *   out(h,w,cout) = relu(conv3x3(in(h,w,cin)) + bias);  w_tok: (9, cout, cin), tap = dy*3+dx. */
/* workspace (may be NULL; strotss_conv3x3_workspace_bytes(h, w, cin, cout) bytes, 0 for big maps): with it, layers of at
 * most 128 output tiles of 64 x 64 (the small maps of the 64 ... 256 px scales) split K over up to 256 workgroups
 * and a finish kernel adds the partial tiles in a fixed order (same result class, bitwise reproducible).
 * (Round 3 measured the finish INSIDE the split-K kernel -- last workgroup to arrive at a tile, agent-scope release /
 * ticket / acquire: 1.32 ms per 64-px step against 0.835 ms with the finish launch; 256 workgroups each writing back
 * their XCD's L2 cost far more than the 24 kernel boundaries they replace.  Not kept.) */
size_t strotss_conv3x3_workspace_bytes(int h, int w, int cin, int cout);
int strotss_conv3x3_relu_fwd(const float* in, int h, int w, int cin, const float* w_tok,
                             const float* bias, int cout, float* out, void* workspace, size_t workspace_bytes,
                             void* stream);
/* Data gradient of the generic layer (no weight gradient: the net is frozen, model.py:45):
 *   gin(h,w,cin) = conv3x3^T(gout(h,w,cout)) [* (act_in > 0) if act_in != NULL]
 *   gout must already carry the ReLU mask of ITS layer.  w_tik: (9, cin, cout) spatially
 *   flipped kernel, tap' = (2-dy)*3+(2-dx).  act_in = the layer's (post-ReLU) input or NULL
 *   when the input came from a max-pool. */
/* accumulate != 0: gin += the (masked) data gradient instead of being overwritten -- the buffer then already holds the
 * hypercolumn taps' contributions of this layer (scattered in one launch for all maps before the backward pass). */
/* Block ends at the split-K scales (ABI 8): the finish kernel of a split-K layer also does the 2x2/2 max-pool that follows it
 * (forward: out AND pool_out = strotss_maxpool2_fwd(out) with its argmax codes pool_code, may be NULL) or precedes it
 * (data-gradient of a layer whose input came from the pool: gin_full(full_h, full_w, cin) (+)= strotss_maxpool2_bwd's result
 * from the codes, full_h / 2 == h, full_w / 2 == w; the pooled gradient itself is never stored) -- one launch less per block
 * and direction, bit for bit the two-launch results.  Split-K layers only (strotss_conv3x3_workspace_bytes(...) > 0 and that
 * workspace given): STROTSS_EINVAL otherwise. */
int strotss_conv3x3_relu_pool_fwd(const float* in, int h, int w, int cin, const float* w_tok, const float* bias, int cout,
                                  float* out, float* pool_out, unsigned char* pool_code, void* workspace,
                                  size_t workspace_bytes, void* stream);
int strotss_conv3x3_dgrad_unpool(const float* gout, int h, int w, int cout, const float* w_tik, int cin,
                                 const unsigned char* pool_code, float* gin_full, int full_h, int full_w, int accumulate,
                                 void* workspace, size_t workspace_bytes, void* stream);
int strotss_conv3x3_dgrad(const float* gout, int h, int w, int cout, const float* w_tik, int cin,
                          const float* act_in, float* gin, int accumulate, void* workspace, size_t workspace_bytes,
                          void* stream);
/* Data gradient of the first layer down to the pixels, preprocess adjoint fused:
 *   gimg(h,w,3) (+)= conv3x3^T(gout(h,w,cout)) / std.   w_tic: (9, 3, cout) flipped kernel.
 *   accumulate != 0 adds to gimg (the hypercolumn scatter of map 0 lands there first).
 *   std3: HOST pointer, 3 floats. */
int strotss_conv3x3_c3_dgrad(const float* gout, int h, int w, int cout, const float* w_tic,
                             const float* std3, float* gimg, int accumulate, void* stream);
/* Winograd form of the two generic-layer entry points above (same results up to fp32 rounding of the
 * transforms).  tile_m = 2: F(2x2,3x3), P = 16 transform-domain GEMMs, 2.25x fewer MACs than direct;
 * tile_m = 4: F(4x4,3x3), P = 36 GEMMs, 4x fewer MACs (f32 error ~1e-5 of the output range instead of ~5e-7).
 * u_pok: (P, cout, cin) = (G g G^T)[p] of the forward kernel, u_pik: (P, cin, cout) of the spatially flipped
 * kernel; both pre-computed once from the frozen weights.
 * workspace >= strotss_conv3x3_winograd_workspace_bytes(h, w, cin, cout, tile_m).
 * u_packed (tile_m = 4 only, may be NULL): the same weights in the MFMA-fragment order written by
 * strotss_conv3x3_winograd_pack (same number of floats).  With it, layers the three-kernel form would run bound
 * by its transform traffic (few output channels, many tiles) run as ONE persistent kernel that keeps the
 * transforms and the 36 GEMMs on chip (csrc/winograd_fused.hip); the workspace is then not touched.
 * u_x3 (tile_m = 4 only, may be NULL): the same weights as "x3 panels" written by strotss_conv3x3_winograd_x3pack
 * (strotss_conv3x3_winograd_x3_bytes bytes).  With it, the 36 GEMMs of the three-kernel form run on the bf16 MFMA by
 * EXACT 3-way operand splitting (every f32 value = h + m + l in bf16, six exact partial products, f32 accumulation:
 * f32-class results at 6/16 of the f32-MFMA cost, csrc/mfma_x3.h).  Default on (STROTSS_X3_CONV=0 switches it off).
 * (Round 2's two bf16x3 forms of the FUSED kernel, both measured slower than its f32-MFMA form, left the library and
 * this ABI in round 3: tools/experiments/winograd_fused_x3.hip.) */
size_t strotss_conv3x3_winograd_workspace_bytes(int h, int w, int cin, int cout, int tile_m);
/* The Winograd weight transform itself: g_nk33 (n, k, 3, 3) = kernel as [out-row][in-col][r][q] -> u_pnk (P, n, k) with
 * u_pnk[a * (tile_m + 2) + b] = (G g G^T)[a][b], P = (tile_m + 2)^2, computed in float64 and rounded once. */
int strotss_conv3x3_winograd_weights(const float* g_nk33, int n, int k, int tile_m, float* u_pnk, void* stream);
/* u_prk: (36, rows, k) -> x3 panels: per position p, element (r, c) split into bf16 planes h, m, l at
 * ((p * (k/32) + c/32) * 3 + plane) * rows * 32 + r * 32 + c % 32   (bf16 units).  k % 32 == 0. */
size_t strotss_conv3x3_winograd_x3_bytes(int rows, int k);
int strotss_conv3x3_winograd_x3pack(const float* u_prk, int rows, int k, void* u_x3, void* stream);
/* u_prk: (36, rows, k) -> u_packed[p][rows/32][k/8][2][32][4]: element (p, r, c) at
 * ((((p * (rows/32) + r/32) * (k/8) + c/8) * 2 + (c%8)/4) * 32 + r%32) * 4 + c%4.  rows % 32 == 0, k % 8 == 0. */
int strotss_conv3x3_winograd_pack(const float* u_prk, int rows, int k, float* u_packed, void* stream);
/* relu_bits_out (fwd, may be NULL, tile_m == 4 only): also writes the sign words of `out` (strotss_relu_bits).
 * relu_bits (dgrad, may be NULL, tile_m == 4 only): the sign words of the layer's INPUT activation; when given, the ReLU
 * mask comes from them and act_in is not read (same result bit for bit: 4 bytes per tile and channel instead of 64).
 * pool_out (may be NULL): also writes strotss_maxpool2_fwd(out) = the (h/2, w/2, cout) input of the next block --
 * from the registers of the fused kernel's epilogue where that kernel runs, by a pooling launch otherwise;
 * pool_code (may be NULL, needs pool_out): the argmax codes of that pooling, see strotss_maxpool2_fwd.
 * accumulate (dgrad, tile_m == 4 with act_in or relu_bits): gin += the masked data-gradient instead of gin = (round 4, ABI 7:
 * the taps of a tapped layer may be scattered into a zeroed gin before the backward pass -- one scatter launch per step --
 * when every producer of such a gradient adds, like strotss_conv3x3_dgrad and strotss_maxpool2_bwd). */
int strotss_conv3x3_winograd_fwd(const float* in, int h, int w, int cin, const float* u_pok,
                                 const float* u_packed, const void* u_x3, const float* bias,
                                 int cout, int tile_m, float* out,
                                 float* pool_out, unsigned char* pool_code, unsigned int* relu_bits_out,
                                 void* workspace, size_t workspace_bytes, void* stream);
int strotss_conv3x3_winograd_dgrad(const float* gout, int h, int w, int cout, const float* u_pik,
                                   const float* u_packed, const void* u_x3, int cin, int tile_m,
                                   const float* act_in, const unsigned int* relu_bits, float* gin, int accumulate,
                                   void* workspace, size_t workspace_bytes, void* stream);
/* Which kernels strotss_conv3x3_winograd_fwd / _dgrad run for a layer shape (the routing is a size policy with
 * environment switches, read once per process): what bench.py names in its roofline.  has_packed / has_x3: whether the
 * caller passes u_packed / u_x3 for this layer. */
#define STROTSS_ROUTE_F2_GEMM_F32 0     /* F(2x2,3x3): input transform, 16 f32-MFMA GEMMs, output transform            */
#define STROTSS_ROUTE_F4_FUSED_F32 1    /* F(4x4,3x3), one persistent kernel, f32 MFMA (csrc/winograd_fused.hip)       */
#define STROTSS_ROUTE_F4_GEMM_F32 2     /* F(4x4,3x3), three kernels, 36 f32-MFMA GEMMs                                 */
#define STROTSS_ROUTE_F4_X3_GEMM_128 3  /* three kernels, 36 bf16x3 GEMMs, 128 x 128 tiles (K16 ring, two workgroups/CU)*/
#define STROTSS_ROUTE_F4_X3_GEMM_64 4   /* the same on 64 x 64 tiles                                                    */
int strotss_conv3x3_winograd_route(int h, int w, int cin, int cout, int tile_m, int has_packed, int has_x3);
/* MEASUREMENT HOOK, process-wide and not thread-safe: which stages of the F(4x4,3x3) three-kernel form are launched
 * from now on (bit 0 input transform, bit 1 GEMMs, bit 2 output transform; 7 = all, the default); returns the previous
 * mask.  bench.py times a layer with masks 1, 3, 7 to split its time into transform and GEMM time (results are
 * meaningless under a partial mask). */
int strotss_debug_winograd_stages(int mask);
/* 2x2/2 VALID max-pool: out(h/2, w/2, c).  code (may be NULL): (h/2, w/2, c) bytes, the index 0..3 of the FIRST
 * max of each window in scan order (0,0),(0,1),(1,0),(1,1), or 4 when that max is not positive. */
int strotss_maxpool2_fwd(const float* in, int h, int w, int c, float* out, unsigned char* code, void* stream);
/* gin(h,w,c) = route gout(h/2,w/2,c) to the first max of each window, times (act > 0) where
 * act(h,w,c) is the pooled layer's input (post-ReLU).  Overwrites gin.  With code != NULL (from the forward pass)
 * act is not read (may be NULL): 1 byte instead of 16 per pooled element.  accumulate != 0: gin += instead of gin =. */
int strotss_maxpool2_bwd(const float* act, int h, int w, int c, const float* gout, float* gin,
                         const unsigned char* code, int accumulate, void* stream);

/* ---------------------------------------------------------------------------------------
 * Sampling._sample: hypercolumn gather  (nn/strotss_utils.py:25-81) and its adjoint
 * --------------------------------------------------------------------------------------- */
typedef struct {
  int n_maps;
  int h[STROTSS_MAX_MAPS], w[STROTSS_MAX_MAPS], c[STROTSS_MAX_MAPS];
  int n_div[STROTSS_MAX_MAPS];                    /* how many entries of div[] apply to map k */
  float div[STROTSS_MAX_DIVS];                    /* cumulative `indices /= y` chain (float32) */
  const float* map[STROTSS_MAX_MAPS];             /* gather: sources; scatter: activations   */
  float* gmap[STROTSS_MAX_MAPS];                  /* scatter only: gradient buffers           */
  /* Row window (spatially sharded trunk): when rows[k] > 0 the buffers of map k hold only rows
   * [row0[k], row0[k] + rows[k]) of a map whose full height is h[k]; coordinates and clipping stay those of
   * the full map (strotss_utils.py:43-64), the taps are then shifted into the window.  rows[k] == 0: whole map. */
  int row0[STROTSS_MAX_MAPS], rows[STROTSS_MAX_MAPS];
  /* scatter (and its plan) only, windowed maps only: != 0 DROPS a tap whose row lies outside the window instead of
   * clamping it to the window's edge -- halo-exchange strips: every rank scatters ALL samples and keeps what lands in
   * the rows it holds.  The gather always clamps (a rank gathers its own samples, whose taps lie inside). */
  int window_drop;
  /* gather, scatter and plan: NULL, or a DEVICE pointer to two ints {begin, end}: only the samples begin <= s < end of
   * the n given are processed (the launch still spans all n; the gather leaves the other rows of `out` untouched).  The
   * block of samples a rank owns changes every step (parallel.sort_indices_by_strip); read from device memory it can
   * change between replays of a captured graph. */
  const int* sample_range;
} strotss_maps_t;
/* out(rows, ld): row s < n = concat_k sample(map_k, idx[s]); bilinear != 0 -> 4-tap weights of
 * strotss_utils.py:43-70, else truncating nearest (72-75).  idx: (n,2) float32 (row, col). */
int strotss_hypercol_gather(const strotss_maps_t* maps, const float* idx, int n, int bilinear,
                            float* out, int ld, void* stream);
/* strotss_hypercol_gather(maps_a, ..., out_a) and strotss_hypercol_gather(maps_b, ..., out_b) at the same n positions in ONE
 * launch (the content and the prediction features of a train step, run_strotss.py:131-137), and, with zero != NULL, a zero
 * fill of the zero_rows x ld matrix `zero` (the step's gradient rows).  maps_b may carry a sample_range. */
int strotss_hypercol_gather2(const strotss_maps_t* maps_a, const strotss_maps_t* maps_b, const float* idx, int n,
                             int bilinear, float* out_a, float* out_b, int ld, float* zero, int zero_rows,
                             void* stream);
/* Adjoint (bilinear only): gmap_k[pixel, c] += w * gfeat[s, off_k + c] * (relu_mask ? map_k>0 : 1)
 * for the maps k in [map_begin, map_end) only (the backward pass of the trunk needs the taps'
 * contributions one layer at a time); gmap[k] may be NULL outside that range.
 * relu_mask_from: maps with index >= relu_mask_from are post-ReLU activations (mask applied).
 * Float atomics, except for maps of at most 64 pixels (the 4 x 4 / 8 x 8 maps of the 64 / 128-px scales, where hundreds of
 * samples share an address): there one workgroup owns a (pixel, channel chunk) and adds its samples in sample order --
 * no atomics, the same bits on every run for those maps (STROTSS_SCATTER_DENSE=0: atomics everywhere). */
int strotss_hypercol_scatter(const strotss_maps_t* maps, const float* idx, int n,
                             const float* gfeat, int ld, int relu_mask_from, int map_begin,
                             int map_end, void* stream);

/* Deterministic form of the same adjoint (no float atomics; bitwise reproducible): strotss_hypercol_scatter_plan orders
 * the (sample, tap) entries of EVERY map by destination pixel once per index set (n <= 1024) into `plan`
 * (strotss_hypercol_scatter_plan_bytes(n_maps) bytes); strotss_hypercol_scatter_sorted then adds, per destination pixel
 * of the maps [map_begin, map_end), the entries' contributions in plan order with one plain read-modify-write. */
size_t strotss_hypercol_scatter_plan_bytes(int n_maps);
int strotss_hypercol_scatter_plan(const strotss_maps_t* maps, const float* idx, int n, void* plan, size_t plan_bytes,
                                  void* stream);
int strotss_hypercol_scatter_sorted(const strotss_maps_t* maps, const void* plan, int n, const float* gfeat, int ld,
                                    int relu_mask_from, int map_begin, int map_end, void* stream);

/* ---------------------------------------------------------------------------------------
 * Box calibration (bench.py only; nothing on the product path calls these).  strotss_calib_mfma: `blocks` workgroups of
 * four waves, each wave 16 * iters register-only v_mfma_f32_32x32x2_f32 (bf16 == 0: 4096 FLOP each) or
 * v_mfma_f32_32x32x16_bf16 (bf16 != 0: 32768 FLOP each); sink: blocks * 256 floats; clocks: per workgroup
 * {s_memtime ticks, s_memrealtime ticks (100 MHz)} of the loop -> the clock the device held.  bf16 == 2: the LOADED loop that
 * tells boxes apart -- 512-thread workgroups, 4 * iters bf16 MFMAs per wave whose operands are re-read from pseudo-random LDS
 * data every trip (sink: blocks * 512 floats).  strotss_calib_copy: a plain
 * 16-byte-per-lane streaming copy of `bytes` (a multiple of 16).
 * --------------------------------------------------------------------------------------- */
int strotss_calib_mfma(int bf16, int blocks, int iters, float* sink, unsigned long long* clocks, void* stream);
int strotss_calib_copy(const void* src, void* dst, size_t bytes, void* stream);
/* dependent-load latency: workgroup b starts at element b * start_stride of `next` (a random cycle over n elements, element i
 * at next[16 * i]: 64-byte stride) and follows `steps` links with one lane; clocks[2b], clocks[2b + 1] = s_memtime /
 * s_memrealtime ticks of its chase */
int strotss_calib_chase(const unsigned* next, unsigned start_stride, int blocks, int steps, unsigned* sink,
                        unsigned long long* clocks, void* stream);

/* ---------------------------------------------------------------------------------------
 * Sample coordinates drawn on the device  (Sampling._make_indices, nn/strotss_utils.py:83-121; the reference draws
 * them inside the traced train_step, run_strotss.py:136 / 115).  Region r (one workgroup each) reads its draw number
 * t = counter[r], draws the grid offsets (k = 0 rows, 1 columns:  philox4x32-10(ctr = (0, 1, t, 0), key = seed)[k] mod
 * step), lists the candidates (off_x + a*step_x, off_y + b*step_y) in tf.meshgrid('xy') order (a fastest), keeps those
 * whose mask byte is nonzero (mask[r] == NULL: all), gives position j of the kept list the shuffle key
 * philox(ctr = (j >> 2, 0, t, 0))[j & 3], writes the min(sample_size, kept) positions with the smallest (key, j) in
 * ascending order as float32 (row, col) pairs to idx[r] (rows beyond that count: zeros) and the count to n_out[r], and
 * stores counter[r] = t + counter_stride.  nn/rand.py:PhiloxStream is the host twin (same numbers through a
 * NumPy-Generator interface), so the oracle's make_indices reproduces every draw.  sample_size <= 1024; the grid has at
 * most 32768 candidates for any image (strotss_index_draw_max_candidates: STROTSS_ERANGE beyond that).
 * --------------------------------------------------------------------------------------- */
#define STROTSS_DRAW_MAX_REGIONS 16
typedef struct strotss_draw_t {
  int h, w;                       /* the scale's image size                                        */
  int step_x, step_y;             /* strotss_utils.py:89-90: max(1, floor / ceil(sqrt(h*w // 128^2))) */
  int sample_size;
  int n_regions;
  unsigned seed_lo, seed_hi;      /* Philox key                                                    */
  unsigned counter_stride;        /* one stream whose draws interleave over the regions: n_regions */
  const unsigned char* mask[STROTSS_DRAW_MAX_REGIONS];   /* (h, w) bytes at THIS scale, or NULL    */
  float* idx[STROTSS_DRAW_MAX_REGIONS];                  /* (sample_size, 2) float32               */
  unsigned* counter;              /* n_regions draw numbers in device memory                       */
  int* n_out;                     /* n_regions counts, or NULL                                     */
  int debug_flags;                /* bit 0: take the kernel's general selection path (tests)       */
} strotss_draw_t;
int strotss_index_draw_max_candidates(int h, int w, int step_x, int step_y);
int strotss_index_draw(const strotss_draw_t* d, void* stream);

/* ---------------------------------------------------------------------------------------
 * Pairwise / moment losses  (nn/losses.py:12-80, run_strotss.py:21-40).
 * Every loss entry computes the loss value AND d(loss)/d(pred) scaled by `gscale`, added
 * into gpred (which the caller zero-fills once per step).  Scalars land in a device array.
 * --------------------------------------------------------------------------------------- */
/* r[i] = rsqrt(max(sum_k x[i,k]^2, 1e-12)) for i < n  (tf.nn.l2_normalize, losses.py:13-14) */
int strotss_row_inv_norm(const float* x, int n, int ld, float* r, void* stream);
/* C[i,j] = 1 - <x_i,y_j> * rx[i]*ry[j], i < nx, j < ny   (losses.py:12-15) */
int strotss_cosine_distance(const float* x, const float* rx, int nx, const float* y,
                            const float* ry, int ny, int ld, float* C, int ldc, void* stream);
/* The same two entry points on the bf16x3 GEMM core (csrc/mfma_x3.h): every f32 value is split EXACTLY into three
 * bf16 values, six exact partial products, f32 accumulation -- f32-class results at 6/16 of the f32-MFMA cost.
 * strotss_row_inv_norm_x3 also writes the rows as "x3 panels" (3 * n * ld bf16; element (i, k), plane p at
 * ((k/32 * 3 + p) * n + i) * 32 + k%32); r may be NULL.  strotss_cosine_distance_x3 takes the panels of x and y;
 * x == y gives an exactly symmetric matrix.  The loss entry points below use this core unless STROTSS_X3=0. */
/* C[i][j] = sqrt(max(|x_i|^2 + |y_j|^2 - 2 x_i.y_j, 1e-6) / d)   (nn/losses.py:18-24, l2_distance); x (nx, ld), y (ny, ld)
 * row-major with columns >= d zero, ld % 32 == 0; C (nx, ldc); workspace: nx + ny floats. */
int strotss_l2_distance(const float* x, int nx, const float* y, int ny, int d, int ld, float* C, int ldc,
                        float* workspace, void* stream);
int strotss_row_inv_norm_x3(const float* x, int n, int ld, float* r, void* panels, void* stream);
int strotss_cosine_distance_x3(const void* x_panels, const float* rx, int nx, const void* y_panels, const float* ry,
                               int ny, int ld, float* C, int ldc, void* stream);
/* Backward of a pairwise distance matrix w.r.t. one of its row sets (what tape.gradient does to nn/losses.py:12-24):
 *   dx[i, :] += g * r[i] * ( sum_{j < k} W[i, j] * B[j, :]  -  x[i, :] * r[i] * q[i] ),   i < n,
 * W (n rows, ldw >= k floats per row, ldw % 32 == 0, entries j >= k zero), B (>= ldw rows of ld floats, rows >= k zero or
 * finite), x / dx (n, ld), ld % 32 == 0; r, q: n floats.  cosine_distance(x, y) with upstream gradient G: W = -G * ry[j],
 * B = y, r = the reciprocal norms of x, q[i] = -sum_j G[i, j] (1 - C[i, j]);  l2_distance: W = -2 G' (G' = G / (2 D C) where
 * the clamp passes), B = y, r = 1, q[i] = -2 sum_j G'[i, j].  f32 MFMA (the loss path's own backward GEMMs run on the
 * bf16x3 core from pre-split panels; this entry is the operator surface's). */
int strotss_rows_gemm_bwd(const float* W, int ldw, int k, const float* B, const float* x, const float* r, const float* q,
                          int n, int ld, float g, float* dx, void* stream);
size_t strotss_selfsim_workspace_bytes(int n, int ld);
/* loss_out[0] = self_similarity(pred, content) (losses.py:55-66);
 * gpred += gscale * dloss/dpred.  pred/content: (rows >= n, ld). */
int strotss_selfsim_fwd_bwd(const float* pred, const float* content, int n, int d, int ld,
                            float gscale, float* gpred, float* loss_out, void* workspace,
                            size_t workspace_bytes, void* stream);
/* Sinkhorn-Knopp transport cost between the style rows (ns) and the prediction rows (n) on the cosine cost matrix.
 * BUILD-DEFINED: the reference's sinkhorn_knopp (losses.py:83-105) is marked untested, is never called and cannot
 * execute (`tf.ones_like` of a Python tuple); this implements its evident intent:
 *   M = cosine_distance(style, pred), K = exp(-l M), v_0 = 1,
 *   n_iter times:  u = (1/ns) / max(K v, 1e-12),  v = (1/n) / max(K^T u, 1e-12),     loss = sum(u * ((K o M) v)),
 * and gpred += gscale * dloss/dpred, differentiated THROUGH the iterations as autodiff would (checked against a
 * float64 autograd restatement, oracle/strotss_oracle.py: sinkhorn_knopp).  n_iter <= 64. */
size_t strotss_sinkhorn_workspace_bytes(int ns, int n, int n_iter);
int strotss_sinkhorn_cos_fwd_bwd(const float* style, const float* rs, int ns, const float* pred, int n, int d,
                                 int ld, float l, int n_iter, float gscale, float* gpred, float* loss_out,
                                 void* workspace, size_t workspace_bytes, void* stream);
/* ld = row stride of the feature matrices (strotss_remd_cos_fwd_bwd), 0 for strotss_palette_remd_fwd_bwd */
/* The same with dist_metrics 'l2' (STROTSS_METRIC_L2) or 'both' (STROTSS_METRIC_BOTH) as the cost (losses.py:27-28): cost matrix
 * on the f32 MFMA with the distance in its epilogue, the scalings and their reverse sweep as above, the clamp of l2_distance
 * passing gradient where m >= 1e-6. */
size_t strotss_sinkhorn_metric_workspace_bytes(int ns, int n, int n_iter);
int strotss_sinkhorn_metric_fwd_bwd(const float* style, int ns, const float* pred, int n, int d, int ld, int metric, float l,
                                    int n_iter, float gscale, float* gpred, float* loss_out, void* workspace,
                                    size_t workspace_bytes, void* stream);
size_t strotss_remd_workspace_bytes(int ns, int n, int ld);
/* loss_out[0] = relaxed_emd(style, pred, 'cosine') (losses.py:69-80); gpred += gscale*dloss/dpred.
 * rs = row_inv_norm(style) (constant per scale). */
/* flags (this entry, strotss_remd_metric_fwd_bwd, strotss_palette_remd_fwd_bwd): STROTSS_REMD_SWAPPED = the caller wants the
 * gradient w.r.t. the reference's FIRST argument x and therefore passed x as `pred` and y as `style` (every metric is symmetric,
 * the value is the same); tf.maximum(R_X, R_Y) sends an exact tie to R_X, which is then the kernel's column branch. */
#define STROTSS_REMD_SWAPPED 1
int strotss_remd_cos_fwd_bwd(const float* style, const float* rs, int ns, const float* pred, int n,
                             int d, int ld, float gscale, float* gpred, float* loss_out, int flags,
                             void* workspace, size_t workspace_bytes, void* stream);
/* The same with its prologue already done (a train step computes the prediction rows' reciprocal norms and x3 panels for
 * the content loss anyway, and the style rows' panels do not change within a scale): pred_inv_norm / pred_panels as
 * strotss_row_inv_norm_x3(pred) writes them -- e.g. the ones strotss_selfsim_fwd_bwd left in ITS workspace
 * (strotss_selfsim_pred_panels: pointers into that workspace, valid until it is reused; *panels == NULL when the cost
 * matrices run on the f32 MFMA, STROTSS_X3=0) -- and style_panels = strotss_row_inv_norm_x3(style).  Same result bit for
 * bit, one launch and two passes over the rows less. */
int strotss_selfsim_pred_panels(void* workspace, size_t workspace_bytes, int n, int ld, const float** inv_norm,
                                const void** panels);
int strotss_remd_cos_fwd_bwd_panels(const float* style, const float* rs, const void* style_panels, int ns,
                                    const float* pred, const float* pred_inv_norm, const void* pred_panels, int n,
                                    int d, int ld, float gscale, float* gpred, float* loss_out, void* workspace,
                                    size_t workspace_bytes, void* stream);
/* loss_out[0] = relaxed_emd(yuv(style[:, :3]), yuv(pred[:, :3]), 'both') (run_strotss.py:37-39);
 * gpred[:, :3] += gscale*dloss/dpred[:, :3].  style/pred are the full (rows, ld) matrices, of
 * which only the first three columns are read.  rgb_to_yuv != 0 applies convert_rgb_to_yuv
 * (strotss_utils.py:166-167) to them first; 0 takes them as they are (losses.relaxed_emd 'both').
 * Workspace: strotss_remd_workspace_bytes(ns, n, 0). */
int strotss_palette_remd_fwd_bwd(const float* style, int ns, const float* pred, int n, int ld,
                                 int rgb_to_yuv, float gscale, float* gpred, float* loss_out, int flags,
                                 void* workspace, size_t workspace_bytes, void* stream);
/* relaxed_emd(style, pred, distance) for the other two entries of dist_metrics (losses.py:27-28) at ANY width d:
 * metric STROTSS_METRIC_L2 -> l2_distance (losses.py:18-24), STROTSS_METRIC_BOTH -> cosine + l2.  Same reductions and
 * tie rules as strotss_remd_cos_fwd_bwd (tf.reduce_min splits among ties, tf.maximum -> first argument);
 * tf.maximum(m, 1e-6) inside l2_distance passes gradient where m >= 1e-6.  gpred += gscale*dloss/dpred. */
#define STROTSS_METRIC_L2 1
#define STROTSS_METRIC_BOTH 2
size_t strotss_remd_metric_workspace_bytes(int ns, int n);
int strotss_remd_metric_fwd_bwd(const float* style, int ns, const float* pred, int n, int d, int ld, int metric,
                                float gscale, float* gpred, float* loss_out, int flags, void* workspace,
                                size_t workspace_bytes, void* stream);
/* The four loss terms of one train step on the same prediction rows (run_strotss.py:131-142, 33-40) in ONE call:
 * loss_content[0] = self_similarity(pred, content), loss_moment[0] = moment_matching(style, pred) (style side given by
 * strotss_moment_stats), loss_remd[0] = relaxed_emd(style, pred) (cosine; style_inv_norm / style_panels =
 * strotss_row_inv_norm_x3(style)), loss_palette[0] = relaxed_emd(yuv(style[:, :3]), yuv(pred[:, :3]), 'both'),
 * gpred += g_content * d(content)/d(pred) + g_moment * ... + g_remd * ... + g_palette * ....  Bit for bit
 * strotss_selfsim_fwd_bwd, strotss_moment_fwd_bwd, strotss_remd_cos_fwd_bwd_panels and strotss_palette_remd_fwd_bwd called in
 * this order; the three forward GEMMs (two symmetric cost matrices, covariance, prediction x style cost matrix) share ONE
 * launch, the prologues of all four terms another, every statistic of the forward products (self-similarity rows, moment
 * scalars, REMD and palette minima) a third, the two branch selections + the self-similarity gradient matrix a fourth:
 * 9 launches instead of 21.
 * bf16x3 core only: STROTSS_EINVAL when STROTSS_X3 / _COST / _MOMENT switch it off (take the separate entry points). */
size_t strotss_step_losses_workspace_bytes(int ns, int n, int ld);
int strotss_step_losses_fwd_bwd(const float* pred, const float* content, int n, int d, int ld, const float* style,
                                const float* style_inv_norm, const void* style_panels, int ns, const float* style_mean,
                                const float* style_cov, float g_content, float g_moment, float g_remd, float g_palette,
                                float* gpred, float* loss_content, float* loss_moment, float* loss_remd, float* loss_palette,
                                void* workspace, size_t workspace_bytes, void* stream);
size_t strotss_moment_workspace_bytes(int n, int ld);
/* style side of moment_matching, once per scale: mean_out(ld), cov_out(ld,ld) = biased covariance */
int strotss_moment_stats(const float* x, int n, int d, int ld, float* mean_out, float* cov_out,
                         void* workspace, size_t workspace_bytes, void* stream);
/* loss_out[0] = moment_matching(style, pred) (losses.py:39-52) with the style statistics
 * cached; gpred += gscale*dloss/dpred. */
int strotss_moment_fwd_bwd(const float* style_mean, const float* style_cov, const float* pred,
                           int n, int d, int ld, float gscale, float* gpred, float* loss_out,
                           void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------
 * Optimiser + output
 * --------------------------------------------------------------------------------------- */
typedef struct {
  int n_tensors;
  float* var[STROTSS_MAX_TENSORS];
  float* rms[STROTSS_MAX_TENSORS];
  const float* grad[STROTSS_MAX_TENSORS];
  int64_t numel[STROTSS_MAX_TENSORS];
} strotss_tensors_t;
/* Keras RMSprop (momentum 0, not centred), run_strotss.py:63,148, all tensors in ONE launch:
 *   rms = rho*rms + (1-rho)*g*g ;  var -= lr * g / (sqrt(rms) + eps) */
int strotss_rmsprop_step(const strotss_tensors_t* t, float lr, float rho, float eps, void* stream);
/* postprocess (strotss_utils.py:170-175): clip[0,1], -min, /max, *255, truncate to uint8.
 * workspace: >= 2*1024 floats. */
int strotss_postprocess(const float* img, int64_t numel, uint8_t* out, float* workspace,
                        void* stream);

#ifdef __cplusplus
}
#endif
#endif /* STROTSS_HIP_H */
