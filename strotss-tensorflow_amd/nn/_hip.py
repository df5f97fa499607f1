"""ctypes binding of libstrotss_hip.so (C ABI: include/strotss_hip.h).

This is the ONLY compute backend of the package: there is no CPU fallback.  Importing this
module never fails (so argument parsing, the oracle-free host logic and the symbol-export test
work on a machine without a GPU), but the first kernel call raises `StrotssHipError` when the
library is missing or no MI355X is visible.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Tuple, Optional, Sequence

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("STROTSS_HIP_LIB") or os.path.join(os.path.dirname(_HERE), "libstrotss_hip.so")

MAX_MAPS, MAX_DIVS, MAX_TENSORS = 12, 8, 8
ABI_VERSION = 8          # must equal strotss_abi_version() of the loaded library (argument lists change with it)


class StrotssHipError(RuntimeError):
    pass


class MapsT(C.Structure):
    _fields_ = [("n_maps", C.c_int),
                ("h", C.c_int * MAX_MAPS), ("w", C.c_int * MAX_MAPS), ("c", C.c_int * MAX_MAPS),
                ("n_div", C.c_int * MAX_MAPS),
                ("div", C.c_float * MAX_DIVS),
                ("map", C.c_void_p * MAX_MAPS),
                ("gmap", C.c_void_p * MAX_MAPS),
                ("row0", C.c_int * MAX_MAPS), ("rows", C.c_int * MAX_MAPS),
                ("window_drop", C.c_int), ("sample_range", C.c_void_p)]


MAX_DRAW_REGIONS = 16


class DrawT(C.Structure):
    """strotss_draw_t (include/strotss_hip.h): the device-side draw of a step's sample coordinates"""
    _fields_ = [("h", C.c_int), ("w", C.c_int), ("step_x", C.c_int), ("step_y", C.c_int), ("sample_size", C.c_int),
                ("n_regions", C.c_int), ("seed_lo", C.c_uint), ("seed_hi", C.c_uint), ("counter_stride", C.c_uint),
                ("mask", C.c_void_p * MAX_DRAW_REGIONS), ("idx", C.c_void_p * MAX_DRAW_REGIONS),
                ("counter", C.c_void_p), ("n_out", C.c_void_p), ("debug_flags", C.c_int)]


class PyramidT(C.Structure):
    _fields_ = [("n_levels", C.c_int), ("h", C.c_int * 8), ("w", C.c_int * 8), ("var", C.c_void_p * 8)]


class TensorsT(C.Structure):
    _fields_ = [("n_tensors", C.c_int),
                ("var", C.c_void_p * MAX_TENSORS), ("rms", C.c_void_p * MAX_TENSORS),
                ("grad", C.c_void_p * MAX_TENSORS), ("numel", C.c_int64 * MAX_TENSORS)]


_P, _I, _F, _Z, _L = C.c_void_p, C.c_int, C.c_float, C.c_size_t, C.c_int64
# name -> (restype, argtypes); must list EVERY symbol include/strotss_hip.h declares
SIGNATURES = {
    "strotss_abi_version": (_I, []),
    "strotss_build_info": (C.c_char_p, []),
    "strotss_resize_bilinear": (_I, [_P, _I, _I, _I, _P, _I, _I, _F, _P, _P]),
    "strotss_fold_pyramid": (_I, [C.POINTER(PyramidT), _P, _P]),
    "strotss_fold_pyramid_adjoint": (_I, [C.POINTER(PyramidT), _P]),
    "strotss_resize_bilinear_adjoint": (_I, [_P, _I, _I, _I, _P, _I, _I, _P]),
    "strotss_conv3x3_c3_fwd": (_I, [_P, _I, _I, _P, _P, _I, C.POINTER(_F), C.POINTER(_F), _P, _P, _P]),
    "strotss_relu_bits_bytes": (_Z, [_I, _I, _I]),
    "strotss_relu_bits": (_I, [_P, _I, _I, _I, _P, _P]),
    "strotss_conv3x3_workspace_bytes": (_Z, [_I, _I, _I, _I]),
    "strotss_conv3x3_relu_fwd": (_I, [_P, _I, _I, _I, _P, _P, _I, _P, _P, _Z, _P]),
    "strotss_conv3x3_dgrad": (_I, [_P, _I, _I, _I, _P, _I, _P, _P, _I, _P, _Z, _P]),
    "strotss_conv3x3_relu_pool_fwd": (_I, [_P, _I, _I, _I, _P, _P, _I, _P, _P, _P, _P, _Z, _P]),
    "strotss_conv3x3_dgrad_unpool": (_I, [_P, _I, _I, _I, _P, _I, _P, _P, _I, _I, _I, _P, _Z, _P]),
    "strotss_conv3x3_c3_dgrad": (_I, [_P, _I, _I, _I, _P, C.POINTER(_F), _P, _I, _P]),
    "strotss_conv3x3_winograd_workspace_bytes": (_Z, [_I, _I, _I, _I, _I]),
    "strotss_conv3x3_winograd_weights": (_I, [_P, _I, _I, _I, _P, _P]),
    "strotss_conv3x3_winograd_pack": (_I, [_P, _I, _I, _P, _P]),
    "strotss_conv3x3_winograd_x3_bytes": (_Z, [_I, _I]),
    "strotss_conv3x3_winograd_x3pack": (_I, [_P, _I, _I, _P, _P]),
    "strotss_conv3x3_winograd_fwd": (_I, [_P, _I, _I, _I, _P, _P, _P, _P, _I, _I, _P, _P, _P, _P, _P, _Z, _P]),
    "strotss_conv3x3_winograd_dgrad": (_I, [_P, _I, _I, _I, _P, _P, _P, _I, _I, _P, _P, _P, _I, _P, _Z, _P]),
    "strotss_conv3x3_winograd_route": (_I, [_I, _I, _I, _I, _I, _I, _I]),
    "strotss_debug_winograd_stages": (_I, [_I]),
    "strotss_maxpool2_fwd": (_I, [_P, _I, _I, _I, _P, _P, _P]),
    "strotss_maxpool2_bwd": (_I, [_P, _I, _I, _I, _P, _P, _P, _I, _P]),
    "strotss_hypercol_gather": (_I, [C.POINTER(MapsT), _P, _I, _I, _P, _I, _P]),
    "strotss_hypercol_gather2": (_I, [C.POINTER(MapsT), C.POINTER(MapsT), _P, _I, _I, _P, _P, _I, _P, _I, _P]),
    "strotss_hypercol_scatter": (_I, [C.POINTER(MapsT), _P, _I, _P, _I, _I, _I, _I, _P]),
    "strotss_hypercol_scatter_plan_bytes": (_Z, [_I]),
    "strotss_hypercol_scatter_plan": (_I, [C.POINTER(MapsT), _P, _I, _P, _Z, _P]),
    "strotss_hypercol_scatter_sorted": (_I, [C.POINTER(MapsT), _P, _I, _P, _I, _I, _I, _I, _P]),
    "strotss_calib_mfma": (_I, [_I, _I, _I, _P, _P, _P]),
    "strotss_calib_copy": (_I, [_P, _P, _Z, _P]),
    "strotss_calib_chase": (_I, [_P, C.c_uint, _I, _I, _P, _P, _P]),
    "strotss_index_draw_max_candidates": (_I, [_I, _I, _I, _I]),
    "strotss_index_draw": (_I, [C.POINTER(DrawT), _P]),
    "strotss_row_inv_norm": (_I, [_P, _I, _I, _P, _P]),
    "strotss_cosine_distance": (_I, [_P, _P, _I, _P, _P, _I, _I, _P, _I, _P]),
    "strotss_l2_distance": (_I, [_P, _I, _P, _I, _I, _I, _P, _I, _P, _P]),
    "strotss_row_inv_norm_x3": (_I, [_P, _I, _I, _P, _P, _P]),
    "strotss_cosine_distance_x3": (_I, [_P, _P, _I, _P, _P, _I, _I, _P, _I, _P]),
    "strotss_rows_gemm_bwd": (_I, [_P, _I, _I, _P, _P, _P, _P, _I, _I, _F, _P, _P]),
    "strotss_selfsim_workspace_bytes": (_Z, [_I, _I]),
    "strotss_selfsim_fwd_bwd": (_I, [_P, _P, _I, _I, _I, _F, _P, _P, _P, _Z, _P]),
    "strotss_sinkhorn_workspace_bytes": (_Z, [_I, _I, _I]),
    "strotss_sinkhorn_cos_fwd_bwd": (_I, [_P, _P, _I, _P, _I, _I, _I, _F, _I, _F, _P, _P, _P, _Z, _P]),
    "strotss_sinkhorn_metric_workspace_bytes": (_Z, [_I, _I, _I]),
    "strotss_sinkhorn_metric_fwd_bwd": (_I, [_P, _I, _P, _I, _I, _I, _I, _F, _I, _F, _P, _P, _P, _Z, _P]),
    "strotss_remd_workspace_bytes": (_Z, [_I, _I, _I]),
    "strotss_remd_cos_fwd_bwd": (_I, [_P, _P, _I, _P, _I, _I, _I, _F, _P, _P, _I, _P, _Z, _P]),
    "strotss_selfsim_pred_panels": (_I, [_P, _Z, _I, _I, C.POINTER(_P), C.POINTER(_P)]),
    "strotss_remd_cos_fwd_bwd_panels": (_I, [_P, _P, _P, _I, _P, _P, _P, _I, _I, _I, _F, _P, _P, _P, _Z, _P]),
    "strotss_palette_remd_fwd_bwd": (_I, [_P, _I, _P, _I, _I, _I, _F, _P, _P, _I, _P, _Z, _P]),
    "strotss_remd_metric_workspace_bytes": (_Z, [_I, _I]),
    "strotss_remd_metric_fwd_bwd": (_I, [_P, _I, _P, _I, _I, _I, _I, _F, _P, _P, _I, _P, _Z, _P]),
    "strotss_step_losses_workspace_bytes": (_Z, [_I, _I, _I]),
    "strotss_step_losses_fwd_bwd": (_I, [_P, _P, _I, _I, _I, _P, _P, _P, _I, _P, _P, _F, _F, _F, _F, _P, _P, _P, _P, _P, _P, _Z, _P]),
    "strotss_moment_workspace_bytes": (_Z, [_I, _I]),
    "strotss_moment_stats": (_I, [_P, _I, _I, _I, _P, _P, _P, _Z, _P]),
    "strotss_moment_fwd_bwd": (_I, [_P, _P, _P, _I, _I, _I, _F, _P, _P, _P, _Z, _P]),
    "strotss_rmsprop_step": (_I, [C.POINTER(TensorsT), _F, _F, _F, _P]),
    "strotss_postprocess": (_I, [_P, _L, _P, _P, _P]),
}

_lib: Optional[C.CDLL] = None


def load_library(path: str = LIB_PATH) -> C.CDLL:
    """dlopen the library and type every entry point.  Needs no GPU."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(path):
        raise StrotssHipError(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"or `make -C strotss-tensorflow_amd/csrc` (there is no CPU fallback)")
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)            # AttributeError -> missing export
        fn.restype = res
        fn.argtypes = args
    got = lib.strotss_abi_version()
    if got != ABI_VERSION:
        raise StrotssHipError(f"{path} has ABI version {got}, this binding needs {ABI_VERSION}: rebuild it "
                              f"(`make -C strotss-tensorflow_amd/csrc`); a stale library would be called with shifted arguments")
    _lib = lib
    return lib


def lib() -> C.CDLL:
    """The library, ready to launch kernels: fails loudly without a GPU."""
    l = load_library()
    if not torch.cuda.is_available():
        raise StrotssHipError("no MI355X visible: libstrotss_hip kernels cannot run (no CPU fallback)")
    return l


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def check(rc: int, what: str):
    if rc != 0:
        kind = {-1: "EINVAL", -2: "EALIGN", -3: "ERANGE"}.get(rc, f"hipError {rc}")
        raise StrotssHipError(f"{what} failed: {kind}")


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    if t is None:
        return None
    return t.data_ptr()


def require(t: torch.Tensor, name: str = "tensor") -> torch.Tensor:
    if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
        raise StrotssHipError(f"{name} must be a contiguous float32 CUDA/HIP tensor, got "
                              f"{t.dtype} {t.device} contiguous={t.is_contiguous()}")
    return t


def make_maps(maps: Sequence[torch.Tensor], divs_per_map: Sequence[Sequence[float]],
              gmaps: Optional[Sequence[torch.Tensor]] = None,
              windows: Optional[Sequence[Tuple[int, int]]] = None, window_drop: bool = False) -> MapsT:
    """maps: list of (1,h,w,c) or (h,w,c) tensors; divs_per_map[k]: the divisor chain for map k
    (each chain is a prefix of the longest one, as the reference's cumulative `indices /= y`).
    windows[k] = (row0, full_height): map k holds rows [row0, row0 + its height) of a map that is
    `full_height` tall (spatially sharded trunk); None: whole maps.  window_drop: the scatter drops taps whose row lies
    outside the window instead of clamping them (halo-exchange strips)."""
    if len(maps) > MAX_MAPS:
        raise StrotssHipError("too many maps")
    m = MapsT()
    m.n_maps = len(maps)
    m.window_drop = 1 if (window_drop and windows is not None) else 0
    longest = max(divs_per_map, key=len)
    if len(longest) > MAX_DIVS:
        raise StrotssHipError("divisor chain too long")
    for i, d in enumerate(longest):
        m.div[i] = d
    for k, t in enumerate(maps):
        require(t, f"map {k}")
        h, w, c = t.shape[-3], t.shape[-2], t.shape[-1]
        m.h[k], m.w[k], m.c[k] = h, w, c
        if windows is not None and windows[k] is not None:
            row0, full_h = windows[k]
            assert 0 <= row0 and row0 + h <= full_h
            m.h[k], m.row0[k], m.rows[k] = full_h, row0, h
        chain = list(divs_per_map[k])
        assert chain == list(longest[:len(chain)])
        m.n_div[k] = len(chain)
        m.map[k] = t.data_ptr()
        if gmaps is not None and gmaps[k] is not None:
            require(gmaps[k], f"gmap {k}")
            assert gmaps[k].numel() == t.numel()
            m.gmap[k] = gmaps[k].data_ptr()
    return m
