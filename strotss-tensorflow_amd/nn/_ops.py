"""Tensor-level wrappers over the C ABI (one Python function per entry point of
include/strotss_hip.h).  Tensors are torch HIP tensors used purely as device memory."""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence, Tuple

import torch

from . import _hip
from ._hip import check, ptr, require, stream_ptr

IMAGENET_MEAN = (0.485, 0.456, 0.406)      # reference nn/model.py:34
IMAGENET_STD = (0.229, 0.224, 0.225)       # reference nn/model.py:35
_MEAN3 = (C.c_float * 3)(*IMAGENET_MEAN)
_STD3 = (C.c_float * 3)(*IMAGENET_STD)


def pad32(v: int) -> int:
    return (v + 31) // 32 * 32


def canonical_device(device) -> torch.device:
    """torch.device with an explicit index ("cuda" -> "cuda:<current>"), comparable with Tensor.device."""
    d = torch.device(device)
    if d.type == "cuda" and d.index is None:
        d = torch.device("cuda", torch.cuda.current_device())
    return d


def hwc(t: torch.Tensor) -> Tuple[int, int, int]:
    return int(t.shape[-3]), int(t.shape[-2]), int(t.shape[-1])


# ------------------------------------------------------------------ images
def resize_bilinear(x: torch.Tensor, oh: int, ow: int, alpha: float = 1.0,
                    add: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    require(x, "resize input")
    ih, iw, c = hwc(x)
    shape = (*x.shape[:-3], oh, ow, c)
    if out is None:
        out = torch.empty(shape, dtype=torch.float32, device=x.device)
    if add is not None:
        require(add, "resize addend")
        assert add.numel() == out.numel()
    check(_hip.lib().strotss_resize_bilinear(ptr(x), ih, iw, c, ptr(out), oh, ow, alpha, ptr(add),
                                             stream_ptr()), "resize_bilinear")
    return out


def fold_pyramid(variables: Sequence[torch.Tensor], out: torch.Tensor) -> Optional[torch.Tensor]:
    """out = fold_laplacian_pyramid(variables) in one launch (reference strotss_utils.py:159-163); None when the pyramid's
    shape is not one the kernel takes (the caller folds level by level)."""
    p = _hip.PyramidT()
    p.n_levels = len(variables)
    for k, v in enumerate(variables):
        require(v, "pyramid level")
        h, w, c = hwc(v)
        if c != 3 or len(variables) > 8:
            return None
        p.h[k], p.w[k], p.var[k] = h, w, v.data_ptr()
    rc = _hip.lib().strotss_fold_pyramid(C.byref(p), ptr(out), stream_ptr())
    if rc == -3:          # STROTSS_ERANGE: not a shrinking, at-least-halving pyramid
        return None
    check(rc, "fold_pyramid")
    return out


def fold_pyramid_adjoint(gvars: Sequence[torch.Tensor]) -> bool:
    """gvars[k] = resize_bilinear_adjoint(gvars[k-1]) for k >= 1 (gvars[0] = gradient of the folded image), two levels per
    launch where the pyramid halves; False when the tensors are not a 3-channel pyramid of at most 8 levels (caller: level by
    level)."""
    if len(gvars) < 2 or len(gvars) > 8 or any(int(g.shape[-1]) != 3 for g in gvars):
        return False
    p = _hip.PyramidT()
    p.n_levels = len(gvars)
    for k, g in enumerate(gvars):
        require(g, "pyramid gradient")
        p.h[k], p.w[k], p.var[k] = int(g.shape[-3]), int(g.shape[-2]), g.data_ptr()
    check(_hip.lib().strotss_fold_pyramid_adjoint(C.byref(p), stream_ptr()), "fold_pyramid_adjoint")
    return True


def resize_bilinear_adjoint(gout: torch.Tensor, ih: int, iw: int,
                            out: Optional[torch.Tensor] = None) -> torch.Tensor:
    require(gout, "resize adjoint input")
    oh, ow, c = hwc(gout)
    if out is None:
        out = torch.empty((*gout.shape[:-3], ih, iw, c), dtype=torch.float32, device=gout.device)
    check(_hip.lib().strotss_resize_bilinear_adjoint(ptr(gout), oh, ow, c, ptr(out), ih, iw, stream_ptr()),
          "resize_bilinear_adjoint")
    return out


# ------------------------------------------------------------------ VGG layers
def _f3(v, default):
    return default if v is None else (C.c_float * 3)(*v)


def relu_bits_buffer(h: int, w: int, c: int, device) -> torch.Tensor:
    """Room for the sign words of an (h, w, c) activation: one int32 per (4x4 tile, channel), see include/strotss_hip.h."""
    return torch.empty(((h + 3) // 4) * ((w + 3) // 4), c, dtype=torch.int32, device=device)


def relu_bits(act, out=None):
    """Sign words of a finished activation tensor (1, h, w, c): byte r, bit q of word (tile, ch) = act[4ty+r, 4tx+q, ch] > 0."""
    require(act, "activation"); h, w, c = hwc(act)
    if out is None:
        out = relu_bits_buffer(h, w, c, act.device)
    check(_hip.lib().strotss_relu_bits(ptr(act), h, w, c, ptr(out), stream_ptr()), "relu_bits")
    return out


def conv3x3_c3_fwd(img, w_kio, bias, out=None, mean=None, std=None, relu_bits_out=None):
    require(img, "image"); h, w, c = hwc(img)
    assert c == 3
    cout = bias.numel()
    if out is None:
        out = torch.empty((1, h, w, cout), dtype=torch.float32, device=img.device)
    check(_hip.lib().strotss_conv3x3_c3_fwd(ptr(img), h, w, ptr(w_kio), ptr(bias), cout, _f3(mean, _MEAN3),
                                            _f3(std, _STD3), ptr(out), ptr(relu_bits_out), stream_ptr()), "conv3x3_c3_fwd")
    return out


def conv3x3_relu_fwd(x, w_tok, bias, out=None, pool_out=None, pool_code=None):
    """pool_out (split-K layers only, conv3x3_direct_splits): the finish kernel also writes maxpool2_fwd(out) (+ pool_code)."""
    require(x, "conv input"); h, w, cin = hwc(x)
    cout = bias.numel()
    if out is None:
        out = torch.empty((1, h, w, cout), dtype=torch.float32, device=x.device)
    nb = _hip.lib().strotss_conv3x3_workspace_bytes(h, w, cin, cout)
    ws = workspaces.get("conv_splitk", nb, x.device) if nb else None
    if pool_out is not None:
        check(_hip.lib().strotss_conv3x3_relu_pool_fwd(ptr(x), h, w, cin, ptr(w_tok), ptr(bias), cout, ptr(out), ptr(pool_out),
                                                       ptr(pool_code), ptr(ws), nb, stream_ptr()), "conv3x3_relu_pool_fwd")
        return out
    check(_hip.lib().strotss_conv3x3_relu_fwd(ptr(x), h, w, cin, ptr(w_tok), ptr(bias), cout, ptr(out), ptr(ws), nb,
                                              stream_ptr()), "conv3x3_relu_fwd")
    return out


def conv3x3_direct_splits(h: int, w: int, cin: int, cout: int) -> bool:
    """True when a (h, w, cin -> cout) layer runs as the split-K direct kernel (which can also ADD to its output)."""
    return _hip.load_library().strotss_conv3x3_workspace_bytes(h, w, cin, cout) > 0


def conv3x3_dgrad(gout, w_tik, cin, act_in=None, out=None, accumulate=False):
    require(gout, "conv grad"); h, w, cout = hwc(gout)
    if out is None:
        out = torch.empty((1, h, w, cin), dtype=torch.float32, device=gout.device)
    nb = _hip.lib().strotss_conv3x3_workspace_bytes(h, w, cout, cin)
    ws = workspaces.get("conv_splitk", nb, gout.device) if nb else None
    check(_hip.lib().strotss_conv3x3_dgrad(ptr(gout), h, w, cout, ptr(w_tik), cin, ptr(act_in), ptr(out), int(accumulate),
                                           ptr(ws), nb, stream_ptr()), "conv3x3_dgrad")
    return out


def conv3x3_dgrad_unpool(gout, w_tik, cin, pool_code, out_full, accumulate=False):
    """Data gradient of a split-K layer whose input came from the 2x2/2 max-pool, written through the pool's adjoint:
    out_full (1, H, W, cin) (+)= maxpool2_bwd(code=pool_code, conv^T(gout)); the pooled gradient is never stored."""
    require(gout, "conv grad"); h, w, cout = hwc(gout)
    require(out_full, "gradient in front of the pool"); fh, fw, fc = hwc(out_full)
    assert fc == cin and fh // 2 == h and fw // 2 == w, (out_full.shape, gout.shape, cin)
    nb = _hip.lib().strotss_conv3x3_workspace_bytes(h, w, cout, cin)
    ws = workspaces.get("conv_splitk", nb, gout.device) if nb else None
    check(_hip.lib().strotss_conv3x3_dgrad_unpool(ptr(gout), h, w, cout, ptr(w_tik), cin, ptr(pool_code), ptr(out_full), fh, fw,
                                                  int(accumulate), ptr(ws), nb, stream_ptr()), "conv3x3_dgrad_unpool")
    return out_full


def conv3x3_c3_dgrad(gout, w_tic, gimg=None, accumulate=False, std=None):
    require(gout, "conv grad"); h, w, cout = hwc(gout)
    if gimg is None:
        gimg = torch.empty((1, h, w, 3), dtype=torch.float32, device=gout.device)
        accumulate = False
    check(_hip.lib().strotss_conv3x3_c3_dgrad(ptr(gout), h, w, cout, ptr(w_tic), _f3(std, _STD3), ptr(gimg),
                                              int(accumulate), stream_ptr()), "conv3x3_c3_dgrad")
    return gimg


def _wino_ws(h, w, cin, cout, tile_m, device):
    nb = _hip.lib().strotss_conv3x3_winograd_workspace_bytes(h, w, cin, cout, tile_m)
    return workspaces.get("winograd", nb, device), nb


def _tile_m(u: torch.Tensor) -> int:
    return {16: 2, 36: 4}[int(u.shape[0])]


_packed = {}       # id(u) -> (weakref(u), packed): fragment-major copies of frozen F(4x4,3x3) weights


def winograd_packed(u: torch.Tensor):
    """The fragment-major copy of a (36, rows, k) Winograd weight tensor for the fused kernel (made once per
    tensor object; the weights are frozen), or None where the fused kernel does not apply."""
    import weakref
    if not winograd_packed_wanted(int(u.shape[0]), int(u.shape[1]), int(u.shape[2])):
        return None
    hit = _packed.get(id(u))
    if hit is not None and hit[0]() is u:
        return hit[1]
    require(u, "winograd weights")
    up = torch.empty_like(u)
    check(_hip.lib().strotss_conv3x3_winograd_pack(ptr(u), int(u.shape[1]), int(u.shape[2]), ptr(up), stream_ptr()),
          "conv3x3_winograd_pack")
    for k in [k for k, v in _packed.items() if v[0]() is None]:
        del _packed[k]
    _packed[id(u)] = (weakref.ref(u), up)
    return up


_x3 = {}           # id(u) -> (weakref(u), panels): bf16x3 "x3 panels" of frozen F(4x4,3x3) weights


def _x3_min_tiles() -> int:
    import os
    return int(os.environ.get("STROTSS_X3_MIN_TILES", "1024"))


def winograd_x3_wanted(p: int, rows: int, k: int, h: int, w: int) -> bool:
    """Whether the library would run an (h, w) layer with (p, rows, k) Winograd weights on the bf16x3 GEMM core: same policy
    as csrc/winograd.hip x3_enabled (at least STROTSS_X3_MIN_TILES 64 x 64 GEMM tiles) on the layers the fused kernel
    does not take.  Pure host arithmetic (tests/test_route_table.py pins it without a GPU)."""
    import os
    tiles = -(-(-(-h // 4) * -(-w // 4)) // 64) * -(-rows // 64) * 36       # 64 x 64 tiles (csrc/winograd.hip x3_enabled)
    if p != 36 or k % 32 or os.environ.get("STROTSS_X3", "1") == "0" \
            or os.environ.get("STROTSS_X3_CONV", "1") == "0":           # default on, see csrc/winograd.hip x3_enabled
        return False
    tiles128 = -(-(-(-h // 4) * -(-w // 4)) // 128) * -(-rows // 128) * 36
    fused_takes_it = (rows <= int(os.environ.get("STROTSS_WINO_FUSED_MAX_COUT", "256"))
                      and (rows < int(os.environ.get("STROTSS_X3_MIN_COUT", "256")) or tiles128 < _x3_min_tiles())
                      and os.environ.get("STROTSS_WINO_FUSED", "1") != "0")
    return not (tiles < _x3_min_tiles() or fused_takes_it)


def winograd_packed_wanted(p: int, rows: int, k: int) -> bool:
    return p == 36 and rows % 32 == 0 and k % 32 == 0


def winograd_x3(u: torch.Tensor, h: int, w: int):
    """The x3 panels (three bf16 planes per f32 weight, K-blocked; csrc/mfma_x3.h) of a (36, rows, k) Winograd weight
    tensor for the bf16x3 GEMM core (made once per tensor object; the weights are frozen), or None where the library
    would not use them for an (h, w) layer (`winograd_x3_wanted`)."""
    import weakref
    if not winograd_x3_wanted(int(u.shape[0]), int(u.shape[1]), int(u.shape[2]), h, w):
        return None
    hit = _x3.get(id(u))
    if hit is not None and hit[0]() is u:
        return hit[1]
    require(u, "winograd weights")
    rows, k = int(u.shape[1]), int(u.shape[2])
    nb = _hip.lib().strotss_conv3x3_winograd_x3_bytes(rows, k)
    up = torch.empty(nb // 2, dtype=torch.bfloat16, device=u.device)
    check(_hip.lib().strotss_conv3x3_winograd_x3pack(ptr(u), rows, k, ptr(up), stream_ptr()), "conv3x3_winograd_x3pack")
    for key in [key for key, v in _x3.items() if v[0]() is None]:
        del _x3[key]
    _x3[id(u)] = (weakref.ref(u), up)
    return up


def conv3x3_winograd_fwd(x, u_pok, bias, out=None, pool_out=None, pool_code=None, relu_bits_out=None):
    """u_pok: (16, cout, cin) -> F(2x2,3x3), (36, cout, cin) -> F(4x4,3x3).  pool_out: (1, h//2, w//2, cout) buffer
    that also receives the 2x2/2 max-pool of the result.  relu_bits_out (F(4x4) only): relu_bits_buffer that receives the
    sign words of the result, for the next layer's conv3x3_winograd_dgrad."""
    require(x, "conv input"); h, w, cin = hwc(x)
    cout = bias.numel()
    if out is None:
        out = torch.empty((1, h, w, cout), dtype=torch.float32, device=x.device)
    m = _tile_m(u_pok)
    assert u_pok.device == x.device, (u_pok.device, x.device)
    ws, nb = _wino_ws(h, w, cin, cout, m, x.device)
    check(_hip.lib().strotss_conv3x3_winograd_fwd(ptr(x), h, w, cin, ptr(u_pok), ptr(winograd_packed(u_pok)),
                                                  ptr(winograd_x3(u_pok, h, w)), ptr(bias),
                                                  cout, m, ptr(out), ptr(pool_out), ptr(pool_code), ptr(relu_bits_out), ptr(ws), nb,
                                                  stream_ptr()),
          "conv3x3_winograd_fwd")
    return out


def conv3x3_winograd_dgrad(gout, u_pik, cin, act_in=None, out=None, relu_bits=None, accumulate=False):
    """relu_bits (F(4x4) only): the sign words of the layer's input activation; the ReLU mask then comes from them instead of
    act_in (same result, 1/16 of the bytes)."""
    require(gout, "conv grad"); h, w, cout = hwc(gout)
    if out is None:
        out = torch.empty((1, h, w, cin), dtype=torch.float32, device=gout.device)
    m = _tile_m(u_pik)
    assert u_pik.device == gout.device, (u_pik.device, gout.device)
    ws, nb = _wino_ws(h, w, cout, cin, m, gout.device)
    check(_hip.lib().strotss_conv3x3_winograd_dgrad(ptr(gout), h, w, cout, ptr(u_pik), ptr(winograd_packed(u_pik)),
                                                    ptr(winograd_x3(u_pik, h, w)), cin,
                                                    m, ptr(act_in), ptr(relu_bits), ptr(out), int(accumulate), ptr(ws), nb,
                                                    stream_ptr()),
          "conv3x3_winograd_dgrad")
    return out


def winograd_weights(g: torch.Tensor, tile_m: int = 2, device=None) -> torch.Tensor:
    """g: (N, K, 3, 3) kernel as [out-channel][in-channel][r][q] -> U (P, N, K) float32 with
    U[a*(m+2)+b] = (G g G^T)[a, b], computed in float64 on the device (P = 16 for tile_m = 2, 36 for tile_m = 4).
    `device`: where a host-held `g` goes (the model's device; default: the current one)."""
    if not g.is_cuda and torch.cuda.is_available():
        g = g.to(canonical_device(device if device is not None else "cuda"))
    g = g.float().contiguous()
    require(g, "conv kernel")
    n, k = int(g.shape[0]), int(g.shape[1])
    u = torch.empty(((tile_m + 2) ** 2, n, k), dtype=torch.float32, device=g.device)
    with torch.cuda.device(g.device):          # stream_ptr() is the CURRENT device's stream: make that device g's
        check(_hip.lib().strotss_conv3x3_winograd_weights(ptr(g), n, k, tile_m, ptr(u), stream_ptr()),
              "conv3x3_winograd_weights")
    return u


def maxpool2_fwd(x, out=None, code=None):
    """code: optional (1, h//2, w//2, c) uint8 buffer receiving the argmax codes for maxpool2_bwd."""
    require(x, "pool input"); h, w, c = hwc(x)
    if out is None:
        out = torch.empty((1, h // 2, w // 2, c), dtype=torch.float32, device=x.device)
    check(_hip.lib().strotss_maxpool2_fwd(ptr(x), h, w, c, ptr(out), ptr(code), stream_ptr()), "maxpool2_fwd")
    return out


def maxpool2_bwd(act, gout, out=None, code=None, accumulate=False):
    """With `code` (from the forward pass) the activations are not read.  accumulate: out += instead of out =."""
    require(act, "pool act"); require(gout, "pool grad"); h, w, c = hwc(act)
    if out is None:
        out = torch.empty_like(act)
        accumulate = False
    check(_hip.lib().strotss_maxpool2_bwd(ptr(act), h, w, c, ptr(gout), ptr(out), ptr(code), int(accumulate), stream_ptr()),
          "maxpool2_bwd")
    return out


# ------------------------------------------------------------------ hypercolumns
def map_divisors(shapes: Sequence[Tuple[int, int]]) -> List[List[float]]:
    """Divisor chain per map: reference nn/strotss_utils.py:31-37 (`indices /= y`, cumulative, the
    axis chosen once from whether log2 of the first shrunk height is an integer)."""
    import math
    chains, cur, index = [], [], None
    for i, (h, w) in enumerate(shapes):
        if i > 0 and h < shapes[i - 1][0]:
            if index is None:
                index = 0 if not (math.log2(h) % 1) else 1
            cur = cur + [shapes[i - 1][index] / shapes[i][index]]
        chains.append(list(cur))
    return chains


def hypercol_gather(maps: Sequence[torch.Tensor], idx: torch.Tensor, bilinear: bool,
                    out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """-> (pad32(n), pad32(D)) feature buffer, rows >= n and columns >= D zero."""
    require(idx, "indices")
    n = idx.shape[0]
    d = sum(int(m.shape[-1]) for m in maps)
    if out is None:
        out = torch.zeros((pad32(n), pad32(d)), dtype=torch.float32, device=idx.device)
    mt = _hip.make_maps(maps, map_divisors([hwc(m)[:2] for m in maps]))
    check(_hip.lib().strotss_hypercol_gather(C.byref(mt), ptr(idx), n, int(bilinear), ptr(out),
                                             out.shape[1], stream_ptr()), "hypercol_gather")
    return out


def hypercol_scatter(maps: Sequence[torch.Tensor], gmaps: Sequence[Optional[torch.Tensor]],
                     idx: torch.Tensor, gfeat: torch.Tensor, relu_mask_from: int = 1,
                     map_begin: int = 0, map_end: Optional[int] = None, maps_t=None):
    """gmaps[k] += adjoint-gather of gfeat's columns of map k, for k in [map_begin, map_end)
    (float atomics).  `maps_t` may carry a prebuilt descriptor (engine hot loop)."""
    require(idx, "indices"); require(gfeat, "feature grads")
    if map_end is None:
        map_end = len(maps)
    mt = maps_t if maps_t is not None else _hip.make_maps(
        maps, map_divisors([hwc(m)[:2] for m in maps]), gmaps)
    check(_hip.lib().strotss_hypercol_scatter(C.byref(mt), ptr(idx), idx.shape[0], ptr(gfeat),
                                              gfeat.shape[1], relu_mask_from, map_begin, map_end,
                                              stream_ptr()), "hypercol_scatter")


def hypercol_scatter_plan(maps_t, idx: torch.Tensor, plan: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Plan of the deterministic scatter for one index set (all maps of `maps_t`); returns the plan buffer."""
    require(idx, "indices")
    nb = _hip.lib().strotss_hypercol_scatter_plan_bytes(int(maps_t.n_maps))
    if plan is None:
        plan = torch.empty(nb, dtype=torch.uint8, device=idx.device)
    check(_hip.lib().strotss_hypercol_scatter_plan(C.byref(maps_t), ptr(idx), idx.shape[0], plan.data_ptr(), plan.numel(),
                                                   stream_ptr()), "hypercol_scatter_plan")
    return plan


def hypercol_scatter_sorted(maps_t, plan: torch.Tensor, n: int, gfeat: torch.Tensor, relu_mask_from: int = 1,
                            map_begin: int = 0, map_end: Optional[int] = None):
    """gmaps[k] += adjoint-gather of gfeat's columns of map k, k in [map_begin, map_end), in plan order (no atomics)."""
    require(gfeat, "feature grads")
    if map_end is None:
        map_end = int(maps_t.n_maps)
    check(_hip.lib().strotss_hypercol_scatter_sorted(C.byref(maps_t), plan.data_ptr(), n, ptr(gfeat), gfeat.shape[1],
                                                     relu_mask_from, map_begin, map_end, stream_ptr()),
          "hypercol_scatter_sorted")


# ------------------------------------------------------------------ losses
def index_draw(h: int, w: int, sample_size: int, seed: int, counters: torch.Tensor, out_idx, masks=None, n_out=None,
               stride=None, general_path: bool = False) -> None:
    """One launch of strotss_index_draw (csrc/draw.hip; reference: Sampling._make_indices, strotss_utils.py:83-121): region r's
    next draw of its stream (draw number counters[r], advanced by `stride` afterwards) -> out_idx[r] (sample_size, 2) float32.
    masks: per region a (h, w) uint8 device tensor at THIS scale (nonzero = keep) or None; n_out: int32 (R,) or None.
    nn/rand.py:PhiloxStream(seed, t) is the host twin of draw number t."""
    from .strotss_utils import sampling_steps
    R = len(out_idx)
    assert 0 < R <= _hip.MAX_DRAW_REGIONS and counters.dtype == torch.int32 and counters.numel() >= R and counters.is_cuda
    d = _hip.DrawT()
    d.h, d.w = int(h), int(w)
    d.step_x, d.step_y = sampling_steps(int(h), int(w))
    d.sample_size, d.n_regions = int(sample_size), R
    d.seed_lo, d.seed_hi = int(seed) & 0xFFFFFFFF, (int(seed) >> 32) & 0xFFFFFFFF
    d.counter_stride = R if stride is None else int(stride)
    for r in range(R):
        o = out_idx[r]
        assert o.is_cuda and o.dtype == torch.float32 and o.is_contiguous() and o.numel() >= 2 * sample_size
        d.idx[r] = o.data_ptr()
        m = None if masks is None else masks[r]
        if m is not None:
            assert m.is_cuda and m.dtype == torch.uint8 and m.is_contiguous() and tuple(m.shape) == (int(h), int(w))
            d.mask[r] = m.data_ptr()
    d.counter = counters.data_ptr()
    d.debug_flags = 1 if general_path else 0     # (tests: the selection path that is exact for ANY key distribution)
    if n_out is not None:
        assert n_out.dtype == torch.int32 and n_out.numel() >= R and n_out.is_cuda
        d.n_out = n_out.data_ptr()
    check(_hip.lib().strotss_index_draw(C.byref(d), stream_ptr()), "index_draw")


def index_draw_counts(h: int, w: int, masks=None):
    """(max candidates of the grid, min over all offset pairs and regions of the number of candidates that survive the
    mask): what decides whether a scale can draw on the device with a fixed sample count.  masks: boolean (h, w) host arrays
    or None.  Pure host arithmetic."""
    import numpy as np
    from .strotss_utils import sampling_steps
    sx, sy = sampling_steps(int(h), int(w))
    most = -(-h // sx) * -(-w // sy)
    least = None
    for ox in range(sx):
        for oy in range(sy):
            if masks is None or all(m is None for m in masks):
                cnt = len(range(ox, h, sx)) * len(range(oy, w, sy))
            else:
                cnt = min(int(np.asarray(m)[ox::sx, oy::sy].sum()) if m is not None else
                          len(range(ox, h, sx)) * len(range(oy, w, sy)) for m in masks)
            least = cnt if least is None else min(least, cnt)
    return most, least


class _WsCache:
    """Grow-only workspace per (device, tag): the C ABI never allocates.  Growth REALLOCATES, which a captured hipGraph
    must never see (the graph holds the old pointer): StepEngine.capture_graph therefore runs one full eager step on a
    side stream first, so every workspace of the step has its final size before capture; sizes depend on the engine's
    shapes only, never on the data."""

    def __init__(self):
        self.bufs = {}

    def get(self, tag: str, nbytes: int, device) -> torch.Tensor:
        key = (tag, str(device))
        b = self.bufs.get(key)
        if b is None or b.numel() < nbytes:
            b = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=device)
            self.bufs[key] = b
        return b


workspaces = _WsCache()


def row_inv_norm(x: torch.Tensor, n: int) -> torch.Tensor:
    require(x, "features")
    r = torch.zeros(x.shape[0], dtype=torch.float32, device=x.device)
    check(_hip.lib().strotss_row_inv_norm(ptr(x), n, x.shape[1], ptr(r), stream_ptr()), "row_inv_norm")
    return r


def cosine_distance(x, rx, nx, y, ry, ny) -> torch.Tensor:
    ldc = pad32(ny)
    Cm = torch.empty((nx, ldc), dtype=torch.float32, device=x.device)
    check(_hip.lib().strotss_cosine_distance(ptr(x), ptr(rx), nx, ptr(y), ptr(ry), ny, x.shape[1], ptr(Cm),
                                             ldc, stream_ptr()), "cosine_distance")
    return Cm


def l2_distance(x, nx, y, ny, d) -> torch.Tensor:
    """sqrt(max(|x_i|^2 + |y_j|^2 - 2 x_i.y_j, 1e-6) / d) on the f32 MFMA; x, y: zero-padded (rows, ld) buffers."""
    ldc = pad32(ny)
    Cm = torch.empty((nx, ldc), dtype=torch.float32, device=x.device)
    ws = torch.empty(nx + ny, dtype=torch.float32, device=x.device)
    check(_hip.lib().strotss_l2_distance(ptr(x), nx, ptr(y), ny, d, x.shape[1], ptr(Cm), ldc, ptr(ws), stream_ptr()),
          "l2_distance")
    return Cm


def row_inv_norm_x3(x, n):
    """(r, panels): the row norms' reciprocals and the rows as x3 panels (three bf16 planes per value) for
    cosine_distance_x3."""
    require(x, "feature matrix")
    r = torch.empty(pad32(n), dtype=torch.float32, device=x.device)
    panels = torch.empty(3 * n * x.shape[1], dtype=torch.bfloat16, device=x.device)
    check(_hip.lib().strotss_row_inv_norm_x3(ptr(x), n, x.shape[1], ptr(r), ptr(panels), stream_ptr()), "row_inv_norm_x3")
    return r, panels


def cosine_distance_x3(xp, rx, nx, yp, ry, ny, ld) -> torch.Tensor:
    """cosine_distance on the bf16x3 GEMM core from the x3 panels of x and y (row_inv_norm_x3)."""
    ldc = pad32(ny)
    Cm = torch.empty((nx, ldc), dtype=torch.float32, device=rx.device)
    check(_hip.lib().strotss_cosine_distance_x3(ptr(xp), ptr(rx), nx, ptr(yp), ptr(ry), ny, ld, ptr(Cm), ldc,
                                                stream_ptr()), "cosine_distance_x3")
    return Cm


def selfsim_fwd_bwd(pred, content, n, d, gscale, gpred, loss_out):
    l = _hip.lib()
    nb = l.strotss_selfsim_workspace_bytes(n, pred.shape[1])
    ws = workspaces.get("selfsim", nb, pred.device)
    check(l.strotss_selfsim_fwd_bwd(ptr(pred), ptr(content), n, d, pred.shape[1], gscale, ptr(gpred),
                                    ptr(loss_out), ptr(ws), nb, stream_ptr()), "selfsim_fwd_bwd")
    # what the workspace now holds (reciprocal norms + x3 panels of exactly these prediction rows), for the borrower below
    global _selfsim_record
    _selfsim_record = (ws.data_ptr(), nb, ptr(pred), n, int(pred.shape[1]), stream_ptr())


_selfsim_record = None
remd_borrow_stats = {"borrowed": 0, "plain": 0}      # which path remd_cos_fwd_bwd_after_selfsim took (tests read it)


def remd_cos_fwd_bwd(style, rs, ns, pred, n, d, gscale, gpred, loss_out, swapped=False):
    """swapped: `pred` is the reference's FIRST argument (gradient to the target side; STROTSS_REMD_SWAPPED)"""
    l = _hip.lib()
    nb = l.strotss_remd_workspace_bytes(ns, n, pred.shape[1])
    ws = workspaces.get("remd", nb, pred.device)
    check(l.strotss_remd_cos_fwd_bwd(ptr(style), ptr(rs), ns, ptr(pred), n, d, pred.shape[1], gscale,
                                     ptr(gpred), ptr(loss_out), int(swapped), ptr(ws), nb, stream_ptr()), "remd_cos_fwd_bwd")


def remd_cos_fwd_bwd_after_selfsim(style, rs, style_panels, ns, pred, n, d, gscale, gpred, loss_out):
    """remd_cos_fwd_bwd for the prediction rows that selfsim_fwd_bwd has just processed (checked against the record that
    call leaves; a mismatch -- other rows, a regrown workspace, another stream -- falls back to the plain call): their reciprocal norms and x3 panels are taken from that workspace, the style rows' panels
    from `style_panels` (row_inv_norm_x3(style)[1], constant within a scale) -- bit for bit remd_cos_fwd_bwd, one launch and
    two passes over the rows less.  style_panels None or the cost matrices on the f32 MFMA: the plain call."""
    l = _hip.lib()
    ld = pred.shape[1]
    nb = l.strotss_selfsim_workspace_bytes(n, ld)
    ws = workspaces.get("selfsim", nb, pred.device)
    # borrow only what selfsim_fwd_bwd is KNOWN to have left there: same buffer (not regrown or re-used since), same rows,
    # same count and stride, same stream -- anything else takes the plain call, which makes its own norms and panels
    if _selfsim_record != (ws.data_ptr(), nb, ptr(pred), n, int(ld), stream_ptr()):
        remd_borrow_stats["plain"] += 1
        return remd_cos_fwd_bwd(style, rs, ns, pred, n, d, gscale, gpred, loss_out)
    rp, xp = C.c_void_p(), C.c_void_p()
    check(l.strotss_selfsim_pred_panels(ptr(ws), nb, n, ld, C.byref(rp), C.byref(xp)), "selfsim_pred_panels")
    if style_panels is None or not xp.value:
        remd_borrow_stats["plain"] += 1
        return remd_cos_fwd_bwd(style, rs, ns, pred, n, d, gscale, gpred, loss_out)
    remd_borrow_stats["borrowed"] += 1
    nbr = l.strotss_remd_workspace_bytes(ns, n, ld)
    wsr = workspaces.get("remd", nbr, pred.device)
    check(l.strotss_remd_cos_fwd_bwd_panels(ptr(style), ptr(rs), ptr(style_panels), ns, ptr(pred), rp.value, xp.value, n, d, ld,
                                            gscale, ptr(gpred), ptr(loss_out), ptr(wsr), nbr, stream_ptr()),
          "remd_cos_fwd_bwd_panels")


def step_losses_available() -> bool:
    import os
    return all(os.environ.get(k, "1") != "0" for k in ("STROTSS_X3", "STROTSS_X3_COST", "STROTSS_X3_MOMENT")) \
        and os.environ.get("STROTSS_GROUPED_LOSSES", "1") != "0"


def step_losses_fwd_bwd(pred, content, n, d, style, rs, style_panels, ns, style_mean, style_cov, g_content, g_moment, g_remd,
                        g_palette, gpred, loss_content, loss_moment, loss_remd, loss_palette):
    """self_similarity + moment_matching + relaxed_emd (cosine) + the YUV palette relaxed_emd of one train step in ONE call
    (strotss_step_losses_fwd_bwd): one prologue launch, the three forward GEMMs in one launch, 13 launches in all; bit for
    bit selfsim_fwd_bwd, moment_fwd_bwd, remd_cos_fwd_bwd_after_selfsim, palette_remd_fwd_bwd in this order."""
    l = _hip.lib()
    ld = int(pred.shape[1])
    nb = l.strotss_step_losses_workspace_bytes(ns, n, ld)
    ws = workspaces.get("step_losses", nb, pred.device)
    check(l.strotss_step_losses_fwd_bwd(ptr(pred), ptr(content), n, d, ld, ptr(style), ptr(rs), ptr(style_panels), ns,
                                        ptr(style_mean), ptr(style_cov), float(g_content), float(g_moment), float(g_remd),
                                        float(g_palette), ptr(gpred), ptr(loss_content), ptr(loss_moment), ptr(loss_remd),
                                        ptr(loss_palette), ptr(ws), nb, stream_ptr()), "step_losses_fwd_bwd")


def sinkhorn_cos_fwd_bwd(style, rs, ns, pred, n, d, l, n_iter, gscale, gpred, loss_out):
    lib = _hip.lib()
    nb = lib.strotss_sinkhorn_workspace_bytes(ns, n, n_iter)
    ws = workspaces.get("sinkhorn", nb, pred.device)
    check(lib.strotss_sinkhorn_cos_fwd_bwd(ptr(style), ptr(rs), ns, ptr(pred), n, d, pred.shape[1], float(l), int(n_iter),
                                           gscale, ptr(gpred), ptr(loss_out), ptr(ws), nb, stream_ptr()),
          "sinkhorn_cos_fwd_bwd")


def sinkhorn_metric_fwd_bwd(style, ns, pred, n, d, metric: str, l, n_iter, gscale, gpred, loss_out):
    lib = _hip.lib()
    nb = lib.strotss_sinkhorn_metric_workspace_bytes(ns, n, n_iter)
    ws = workspaces.get("sinkhorn", nb, pred.device)
    check(lib.strotss_sinkhorn_metric_fwd_bwd(ptr(style), ns, ptr(pred), n, d, pred.shape[1], REMD_METRICS[metric], float(l),
                                              int(n_iter), gscale, ptr(gpred), ptr(loss_out), ptr(ws), nb, stream_ptr()),
          "sinkhorn_metric_fwd_bwd")


def palette_remd_fwd_bwd(style, ns, pred, n, gscale, gpred, loss_out, rgb_to_yuv=True, swapped=False):
    l = _hip.lib()
    nb = l.strotss_remd_workspace_bytes(ns, n, 0)
    ws = workspaces.get("remd", nb, pred.device)
    assert style.shape[1] == pred.shape[1]
    check(l.strotss_palette_remd_fwd_bwd(ptr(style), ns, ptr(pred), n, pred.shape[1], int(rgb_to_yuv), gscale,
                                         ptr(gpred), ptr(loss_out), int(swapped), ptr(ws), nb, stream_ptr()), "palette_remd_fwd_bwd")


REMD_METRICS = {"l2": 1, "both": 2}      # STROTSS_METRIC_L2 / STROTSS_METRIC_BOTH (include/strotss_hip.h)


def remd_metric_fwd_bwd(style, ns, pred, n, d, metric: str, gscale, gpred, loss_out, swapped=False):
    """relaxed_emd with dist_metrics 'l2' / 'both' at any width (reference losses.py:18-28, 69-80)."""
    l = _hip.lib()
    nb = l.strotss_remd_metric_workspace_bytes(ns, n)
    ws = workspaces.get("remd_metric", nb, pred.device)
    assert style.shape[1] == pred.shape[1]
    check(l.strotss_remd_metric_fwd_bwd(ptr(style), ns, ptr(pred), n, d, pred.shape[1], REMD_METRICS[metric], gscale,
                                        ptr(gpred), ptr(loss_out), int(swapped), ptr(ws), nb, stream_ptr()), "remd_metric_fwd_bwd")


def rows_gemm_bwd(W, k, B, x, r, q, n, g, dx):
    """dx[i, :] += g * r[i] * (sum_j W[i, j] B[j, :] - x[i, :] * r[i] * q[i])   for i < n, j < k  (strotss_rows_gemm_bwd):
    the backward of a pairwise distance matrix w.r.t. one of its two row sets (reference losses.py:12-24 under
    tape.gradient) -- W = d(loss)/d(product) scaled by the other side's factors, q = the normalisation's rank-one term."""
    assert W.is_contiguous() and B.is_contiguous() and x.is_contiguous() and dx.is_contiguous()
    assert int(W.shape[1]) % 32 == 0 and int(B.shape[1]) == int(x.shape[1]) == int(dx.shape[1]) and int(B.shape[0]) >= int(W.shape[1])
    check(_hip.lib().strotss_rows_gemm_bwd(ptr(W), int(W.shape[1]), int(k), ptr(B), ptr(x), ptr(r), ptr(q), n, int(x.shape[1]),
                                           float(g), ptr(dx), stream_ptr()), "rows_gemm_bwd")


def moment_stats(x, n, d):
    l = _hip.lib()
    ld = x.shape[1]
    nb = l.strotss_moment_workspace_bytes(n, ld)
    ws = workspaces.get("moment", nb, x.device)
    mean = torch.empty(ld, dtype=torch.float32, device=x.device)
    cov = torch.empty((ld, ld), dtype=torch.float32, device=x.device)
    check(l.strotss_moment_stats(ptr(x), n, d, ld, ptr(mean), ptr(cov), ptr(ws), nb, stream_ptr()),
          "moment_stats")
    return mean, cov


def moment_fwd_bwd(style_mean, style_cov, pred, n, d, gscale, gpred, loss_out):
    l = _hip.lib()
    ld = pred.shape[1]
    nb = l.strotss_moment_workspace_bytes(n, ld)
    ws = workspaces.get("moment", nb, pred.device)
    check(l.strotss_moment_fwd_bwd(ptr(style_mean), ptr(style_cov), ptr(pred), n, d, ld, gscale, ptr(gpred),
                                   ptr(loss_out), ptr(ws), nb, stream_ptr()), "moment_fwd_bwd")


# ------------------------------------------------------------------ optimiser / output
def rmsprop_step(variables, rms, grads, lr: float, rho: float = 0.99, eps: float = 1e-8):
    t = _hip.TensorsT()
    t.n_tensors = len(variables)
    for k, (v, r, g) in enumerate(zip(variables, rms, grads)):
        require(v, "variable"); require(r, "rms"); require(g, "grad")
        assert v.numel() == r.numel() == g.numel()
        t.var[k], t.rms[k], t.grad[k], t.numel[k] = v.data_ptr(), r.data_ptr(), g.data_ptr(), v.numel()
    check(_hip.lib().strotss_rmsprop_step(C.byref(t), lr, rho, eps, stream_ptr()), "rmsprop_step")


def postprocess(img: torch.Tensor) -> torch.Tensor:
    require(img, "image")
    out = torch.empty(img.shape, dtype=torch.uint8, device=img.device)
    ws = torch.empty(2048, dtype=torch.float32, device=img.device)
    check(_hip.lib().strotss_postprocess(ptr(img), img.numel(), ptr(out), ptr(ws), stream_ptr()), "postprocess")
    return out
