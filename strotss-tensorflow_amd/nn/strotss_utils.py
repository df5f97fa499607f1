"""Sampling (hypercolumn gather), Laplacian pyramid, YUV, postprocess, masks -- mirrors the
reference's nn/strotss_utils.py:12-201 on torch HIP tensors and the kernels of libstrotss_hip.so.

The optimisation loop itself goes through `nn.engine` (pre-allocated buffers, fused backward);
the functions here are the reference's operator surface, differentiable where the reference's are
(`fold_laplacian_pyramid`, `Sampling.bilinear`) through small `torch.autograd.Function` bridges."""
from __future__ import annotations

import math
from functools import partialmethod
from typing import List, Optional, Sequence, Tuple, Union

import numpy as np
import torch

from . import _ops, rand, utils

RGB2YUV = ((0.299, -0.14714119, 0.61497538),       # tf.image.rgb_to_yuv kernel, rgb @ M
           (0.587, -0.28886916, -0.51496512),
           (0.114, 0.43601035, -0.10001026))


# ----------------------------------------------------------------------------- autograd bridges
class _ResizeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, oh, ow):
        ctx.in_hw = (int(x.shape[-3]), int(x.shape[-2]))
        return _ops.resize_bilinear(x.detach().contiguous(), oh, ow)

    @staticmethod
    def backward(ctx, g):
        return _ops.resize_bilinear_adjoint(g.contiguous(), *ctx.in_hw), None, None


def _resize(x: torch.Tensor, oh: int, ow: int) -> torch.Tensor:
    if x.requires_grad:
        return _ResizeFn.apply(x, oh, ow)
    return _ops.resize_bilinear(x.contiguous(), oh, ow)


class _GatherFn(torch.autograd.Function):
    """feats(n, D) = bilinear hypercolumns of the maps at idx; gradient to every map that needs it."""

    @staticmethod
    def forward(ctx, idx, n_maps, *maps):
        maps_c = [m.detach().contiguous() for m in maps]
        out = _ops.hypercol_gather(maps_c, idx, True)
        ctx.maps = maps_c
        ctx.idx = idx
        d = sum(int(m.shape[-1]) for m in maps_c)
        ctx.nd = (idx.shape[0], d)
        return out[:idx.shape[0], :d]

    @staticmethod
    def backward(ctx, g):
        n, d = ctx.nd
        gbuf = torch.zeros((_ops.pad32(n), _ops.pad32(d)), dtype=torch.float32, device=g.device)
        gbuf[:n, :d] = g
        gm = [torch.zeros_like(m) for m in ctx.maps]
        # the maps are plain inputs here (any ReLU mask belongs to whoever produced them)
        _ops.hypercol_scatter(ctx.maps, gm, ctx.idx, gbuf, relu_mask_from=len(gm))
        return (None, None, *gm)


# ----------------------------------------------------------------------------- Sampling
def sampling_steps(h: int, w: int) -> Tuple[int, int]:
    """reference strotss_utils.py:89-90"""
    area = math.sqrt((h * w) // (128 ** 2))
    return max(1, math.floor(area)), max(1, math.ceil(area))


def make_indices_np(h: int, w: int, bilinear_sampling: bool, sample_size: int, rng: np.random.Generator,
                    mask_hw: Optional[np.ndarray] = None) -> np.ndarray:
    """reference strotss_utils.py:83-121 given the ALREADY resized+thresholded boolean mask (h,w).
    Strided grid with random offsets (bilinear mode) or every pixel, optional mask filter, joint
    shuffle of the (row, col) pairs, first `sample_size`, float32."""
    if bilinear_sampling:
        step_x, step_y = sampling_steps(h, w)
        off_x = int(rng.integers(0, step_x))
        off_y = int(rng.integers(0, step_y))
        X = np.arange(h)[off_x::step_x]
        Y = np.arange(w)[off_y::step_y]
    else:
        X, Y = np.arange(h), np.arange(w)
    XX, YY = np.meshgrid(X, Y)                      # tf.meshgrid default 'xy'
    ret = np.stack([XX.reshape(-1), YY.reshape(-1)], axis=1)
    if mask_hw is not None:
        ret = ret[mask_hw[ret[:, 0], ret[:, 1]]]
    if hasattr(rng, "permutation_head"):                # PhiloxStream: the same first `sample_size` entries, without the full sort
        ret = ret[rng.permutation_head(ret.shape[0], sample_size)]
    else:
        ret = ret[rng.permutation(ret.shape[0])][:sample_size]
    return ret.astype(np.float32)


def mask_at_scale(mask: torch.Tensor, h: int, w: int) -> np.ndarray:
    """reference strotss_utils.py:105-110: bilinear resize of the (H,W,1) float mask to the scale,
    `> 0.5` (or all-true if the resized mask is all < 0.1).  Boolean (h,w) host array."""
    m = mask.to(utils.device()).float()
    if m.dim() == 2:
        m = m[..., None]
    m = _ops.resize_bilinear(m.contiguous(), h, w)[..., 0]
    if float(m.max()) < 0.1:
        keep = (m + 1) > 0.5
    else:
        keep = m > 0.5
    return keep.cpu().numpy()


class Sampling:
    """reference strotss_utils.py:20-136.  `rng` defaults to nn.rand.index_rng."""

    def __init__(self, sample_size: int, rng: Optional[np.random.Generator] = None, **kwargs):
        self.sample_size = sample_size
        self.rng = rng

    def _rng(self):
        return self.rng if self.rng is not None else rand.index_rng

    def _sample(self, xs: List[torch.Tensor], indices: torch.Tensor, bilinear_sampling: bool) -> torch.Tensor:
        n = indices.shape[0]
        d = sum(int(x.shape[-1]) for x in xs)
        if bilinear_sampling and any(x.requires_grad for x in xs):
            return _GatherFn.apply(indices, len(xs), *xs)
        out = _ops.hypercol_gather([x.detach().contiguous() for x in xs], indices, bilinear_sampling)
        return out[:n, :d]

    def _make_indices(self, base_tensor: torch.Tensor, bilinear_sampling: bool,
                      mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        _, h, w, *_ = base_tensor.shape
        mk = mask_at_scale(mask, int(h), int(w)) if mask is not None else None
        idx = make_indices_np(int(h), int(w), bilinear_sampling, self.sample_size, self._rng(), mk)
        return torch.from_numpy(idx).to(base_tensor.device)

    def __call__(self, xs: List[torch.Tensor], ys: Optional[List[torch.Tensor]] = None,
                 mask: Optional[torch.Tensor] = None, bilinear_sampling: bool = False,
                 indices: Optional[torch.Tensor] = None
                 ) -> Union[torch.Tensor, Tuple[torch.Tensor, torch.Tensor]]:
        if indices is None:
            indices = self._make_indices(xs[0], bilinear_sampling, mask)
        ret = self._sample(xs, indices, bilinear_sampling)
        if ys:
            ret_y = self._sample(ys, indices, bilinear_sampling)
            return ret, ret_y
        return ret

    bilinear = partialmethod(__call__, bilinear_sampling=True)


# ----------------------------------------------------------------------------- Laplacian pyramid
def make_laplacian(x: torch.Tensor, return_downscale: bool = False):
    """reference strotss_utils.py:139-146"""
    h, w = int(x.shape[1]), int(x.shape[2])
    hd, wd = max(h // 2, 1), max(w // 2, 1)
    xc = x.contiguous()
    temp = _ops.resize_bilinear(xc, hd, wd)
    pyr = _ops.resize_bilinear(temp, h, w, -1.0, xc)          # x - up(down(x)) in one kernel
    if return_downscale:
        return pyr, temp
    return pyr


def make_laplacian_pyramid(x: torch.Tensor, levels: int = 5) -> List[torch.Tensor]:
    """reference strotss_utils.py:149-156"""
    xs = []
    curx = x
    for _ in range(levels):
        pyr, curx = make_laplacian(curx, return_downscale=True)
        xs.append(pyr)
    xs.append(curx)
    return xs


def fold_laplacian_pyramid(xs: Sequence[torch.Tensor]) -> torch.Tensor:
    """reference strotss_utils.py:159-163 (differentiable w.r.t. every level)."""
    ret = xs[-1]
    needs_grad = any(x.requires_grad for x in xs)
    for x in reversed(xs[:-1]):
        h, w = int(x.shape[1]), int(x.shape[2])
        if needs_grad:
            ret = x + _resize(ret, h, w)
        else:
            ret = _ops.resize_bilinear(ret.contiguous(), h, w, 1.0, x.contiguous())
    return ret


def convert_rgb_to_yuv(x: torch.Tensor) -> torch.Tensor:
    """reference strotss_utils.py:166-167"""
    # three axpys per output channel instead of a (N, 3) x (3, 3) library GEMM
    r, g, b = x[:, 0:1], x[:, 1:2], x[:, 2:3]
    return torch.cat([r * RGB2YUV[0][k] + g * RGB2YUV[1][k] + b * RGB2YUV[2][k] for k in range(3)], dim=1)


def postprocess(final: torch.Tensor) -> torch.Tensor:
    """reference strotss_utils.py:170-175 -> uint8 (H,W,3)"""
    return _ops.postprocess(final.detach().float().contiguous())[0]


def _colour_keys(path: str, max_size: Optional[int], pixel_threth: int) -> np.ndarray:
    """(H, W) int64 key per pixel of a colour-coded region image: its channels floored to multiples of `pixel_threth`
    (uint8 as decoded, or float32 when `max_size` made load_image resize it) and packed so that ascending keys are
    ascending (r, g, b) triples."""
    img = utils.load_image(path, max_size, dtype=torch.uint8, batch_expand=False).cpu().numpy()
    q = (np.floor_divide(img, pixel_threth) * pixel_threth).astype(np.int64)
    return (q[..., 0] << 32) | (q[..., 1] << 16) | q[..., 2]


def load_mask(content_path: str, style_path: str, max_size: Optional[int],
              pixel_threth: int = 255, sample_threth: int = 10000):
    """reference strotss_utils.py:178-201: paired (H, W, 1) float 0/1 region masks from two colour-coded images -- one
    pair per colour that covers at least `sample_threth` content pixels and occurs in the style image too, in ascending
    (r, g, b) order; bare Exception('No mask found') when there is none."""
    c_keys = _colour_keys(content_path, max_size, pixel_threth)
    s_keys = _colour_keys(style_path, max_size, pixel_threth)
    colours, counts = np.unique(c_keys, return_counts=True)
    chosen = colours[(counts >= sample_threth) & np.isin(colours, s_keys)]
    if chosen.size == 0:
        raise Exception('No mask found')

    def region(keys: np.ndarray, colour) -> torch.Tensor:
        return torch.from_numpy((keys == colour).astype(np.float32))[..., None]
    return [region(c_keys, c) for c in chosen], [region(s_keys, c) for c in chosen]
