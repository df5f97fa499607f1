"""Distance matrices and the three STROTSS losses -- mirrors the reference's nn/losses.py:4-105.

Each loss is ONE fused forward+backward pass over the HIP kernels (cost-matrix GEMMs on the bf16x3
core of csrc/mfma_x3.h -- exact three-way bf16 split of the f32 operands, f32 accumulation; STROTSS_X3=0: the f32
MFMA --, reductions, sparse backward); the autograd bridge stores d(loss)/d(prediction) computed in
that pass and scales it by the incoming scalar gradient.  As in `run_strotss.py`, gradients flow
to the *prediction* argument only: `x` for `self_similarity(x, y)`, `y` for
`moment_matching(x, y)` and `relaxed_emd(x, y)`; asking for the other side raises."""
from __future__ import annotations

import torch

from . import _ops


def mse(x: torch.Tensor, y: torch.Tensor, axis=None, keepdims=False) -> torch.Tensor:
    d = (x - y) ** 2
    return d.mean() if axis is None else d.mean(dim=axis, keepdim=keepdims)


def mae(x: torch.Tensor, y: torch.Tensor, axis=None, keepdims=False) -> torch.Tensor:
    d = (x - y).abs()
    return d.mean() if axis is None else d.mean(dim=axis, keepdim=keepdims)


def _buf(x: torch.Tensor) -> torch.Tensor:
    """(n, d) -> zero-padded (pad32(n), pad32(d)) feature buffer the kernels want."""
    x = reshape_2d(x).detach()
    n, d = x.shape
    b = torch.zeros((_ops.pad32(n), _ops.pad32(d)), dtype=torch.float32, device=x.device)
    b[:n, :d] = x
    return b


def cosine_distance(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """1 - l2_normalize(x) @ l2_normalize(y)^T  (reference losses.py:12-15); forward only."""
    bx, by = _buf(x), _buf(y)
    nx, ny = x.shape[0], y.shape[0]
    return _ops.cosine_distance(bx, _ops.row_inv_norm(bx, nx), nx, by, _ops.row_inv_norm(by, ny), ny)[:, :ny]


def l2_distance(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """sqrt(max(|x|^2 + |y|^2 - 2 x y^T, 1e-6) / D)  (reference losses.py:18-24); forward only, on the f32 MFMA."""
    bx, by = _buf(x), _buf(y)
    nx, ny = x.shape[0], y.shape[0]
    return _ops.l2_distance(bx, nx, by, ny, int(reshape_2d(x).shape[1]))[:, :ny]


dist_metrics = {'cosine': cosine_distance, 'l2': l2_distance,
                'both': lambda x, y: cosine_distance(x, y) + l2_distance(x, y)}


def reshape_2d(x: torch.Tensor, channel_axis: int = -1) -> torch.Tensor:
    """reference losses.py:31-36: squeeze, then (-1, C)."""
    if x.dim() == 2:
        return x
    x = x.squeeze()
    return x.reshape(-1, x.shape[channel_axis])


class _FusedLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, runner):
        p = reshape_2d(pred)
        n, d = p.shape
        bp = _buf(p)
        g = torch.zeros_like(bp)
        loss = torch.zeros(1, dtype=torch.float32, device=p.device)
        runner(bp, n, d, g, loss)
        ctx.save_for_backward(g)
        ctx.shape = (n, d, tuple(pred.shape))
        return loss[0]

    @staticmethod
    def backward(ctx, gl):
        (g,) = ctx.saved_tensors
        n, d, shape = ctx.shape
        return (g[:n, :d] * gl).reshape(shape), None


def _no_grad_side(t: torch.Tensor, what: str):
    if t.requires_grad:
        raise NotImplementedError(f"{what}: gradients flow to the prediction argument only")


def moment_matching(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """mae(cov x, cov y) + mae(mean x, mean y)  (reference losses.py:39-52); x = target, y = prediction."""
    _no_grad_side(x, "moment_matching(x, y)")
    bx = _buf(x)
    nx, d = reshape_2d(x).shape
    mean, cov = _ops.moment_stats(bx, nx, d)
    return _FusedLoss.apply(y, lambda bp, n, dd, g, loss: _ops.moment_fwd_bwd(mean, cov, bp, n, dd, 1.0, g, loss))


def self_similarity(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """reference losses.py:55-66; x = prediction, y = target (ContentLoss swaps them)."""
    _no_grad_side(y, "self_similarity(x, y)")
    by = _buf(y)
    return _FusedLoss.apply(x, lambda bp, n, d, g, loss: _ops.selfsim_fwd_bwd(bp, by, n, d, 1.0, g, loss))


def relaxed_emd(x: torch.Tensor, y: torch.Tensor, distance: str = 'cosine') -> torch.Tensor:
    """max(mean_i min_j C, mean_j min_i C)  (reference losses.py:69-80); x = target, y = prediction.
    Every entry of `dist_metrics` at any width, as the reference (losses.py:27-28, 74): 'cosine' on the bf16x3 cost
    GEMM, 'l2' and 'both' on the f32-MFMA cost GEMM with the distance in its epilogue; width 3 with 'both' (the palette
    term of run_strotss.py:36-39) takes the VALU kernel that never stores the cost matrix."""
    if distance not in dist_metrics:
        raise KeyError(distance)
    _no_grad_side(x, "relaxed_emd(x, y)")
    bx = _buf(x)
    ns, d = reshape_2d(x).shape
    if distance == 'cosine':
        rs = _ops.row_inv_norm(bx, ns)
        return _FusedLoss.apply(y, lambda bp, n, dd, g, loss: _ops.remd_cos_fwd_bwd(bx, rs, ns, bp, n, dd, 1.0, g, loss))
    if distance == 'both' and d == 3:
        return _FusedLoss.apply(y, lambda bp, n, dd, g, loss: _ops.palette_remd_fwd_bwd(bx, ns, bp, n, 1.0, g, loss,
                                                                                       rgb_to_yuv=False))
    return _FusedLoss.apply(y, lambda bp, n, dd, g, loss: _ops.remd_metric_fwd_bwd(bx, ns, bp, n, dd, distance, 1.0, g,
                                                                                  loss))


def sinkhorn_knopp(x: torch.Tensor, y: torch.Tensor, distance: str = 'cosine', l: int = 10,
                   N_iter: int = 30) -> torch.Tensor:
    """Sinkhorn-Knopp transport cost between x (target) and y (prediction), differentiable w.r.t. y.
    BUILD-DEFINED: reference losses.py:83-105 is marked `# TODO: untested`, is never called and cannot execute
    (`tf.ones_like(shape)` on a Python tuple), so there is no behaviour to match; this is its evident intent
    (K = exp(-l M), uniform marginals, N_iter alternating scalings from v = 1, cost sum(u * ((K*M) v)), gradient
    through the iterations), pinned by the float64 autograd restatement oracle.strotss_oracle.sinkhorn_knopp.
    Only the cosine cost is provided on the HIP path."""
    if not l > 0:
        raise ValueError("l must be greater than 0")
    if distance != 'cosine':
        raise NotImplementedError(f"sinkhorn_knopp(distance={distance!r}): the HIP path covers 'cosine'")
    _no_grad_side(x, "sinkhorn_knopp(x, y)")
    bx = _buf(x)
    ns, _ = reshape_2d(x).shape
    rs = _ops.row_inv_norm(bx, ns)
    return _FusedLoss.apply(y, lambda bp, n, dd, g, loss: _ops.sinkhorn_cos_fwd_bwd(bx, rs, ns, bp, n, dd, l, N_iter, 1.0,
                                                                                     g, loss))
