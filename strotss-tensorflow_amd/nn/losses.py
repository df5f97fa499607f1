"""Distance matrices and the three STROTSS losses -- mirrors the reference's nn/losses.py:4-105.

Each loss is ONE fused forward+backward pass over the HIP kernels (cost-matrix GEMMs on the bf16x3
core of csrc/mfma_x3.h -- exact three-way bf16 split of the f32 operands, f32 accumulation; STROTSS_X3=0: the f32
MFMA --, reductions, sparse backward); the autograd bridge stores d(loss)/d(prediction) computed in
that pass and scales it by the incoming scalar gradient.  `run_strotss.py` needs the gradient of the
*prediction* argument only (`x` of `self_similarity(x, y)`, `y` of `moment_matching(x, y)` and `relaxed_emd(x, y)`);
TF differentiates both sides, and so do these (round 4): the three losses are symmetric in their two row sets, the
target side's gradient is the same fused pass with the roles exchanged.  `cosine_distance` / `l2_distance` are
differentiable in both arguments too (`strotss_rows_gemm_bwd`).  `sinkhorn_knopp` (build-defined, see its docstring) keeps
the prediction-side gradient only: its alternating scalings are not symmetric in the two sides."""
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


class _PairwiseDistance(torch.autograd.Function):
    """cosine_distance / l2_distance with gradients to BOTH row sets (TF differentiates both sides of losses.py:12-24).
    Forward: the cost-matrix GEMM with the distance in its epilogue.  Backward: per side one strotss_rows_gemm_bwd
    (dX = r * (W B - xhat q), the self-similarity backward GEMM with the roles of the operands set by hand); W and q are
    element-wise functions of the upstream gradient and the forward matrix."""

    @staticmethod
    def forward(ctx, x, y, kind):
        x2, y2 = reshape_2d(x), reshape_2d(y)
        bx, by = _buf(x2), _buf(y2)
        nx, ny, d = int(x2.shape[0]), int(y2.shape[0]), int(x2.shape[1])
        if kind == 'cosine':
            rx, ry = _ops.row_inv_norm(bx, nx), _ops.row_inv_norm(by, ny)
            C = _ops.cosine_distance(bx, rx, nx, by, ry, ny)[:, :ny]
        else:
            rx = ry = None
            C = _ops.l2_distance(bx, nx, by, ny, d)[:, :ny]
        ctx.kind, ctx.dims, ctx.shapes = kind, (nx, ny, d), (tuple(x.shape), tuple(y.shape))
        ctx.save_for_backward(bx, by, C, *([rx, ry] if kind == 'cosine' else []))
        return C.clone()

    @staticmethod
    def backward(ctx, G):
        bx, by, C = ctx.saved_tensors[:3]
        nx, ny, d = ctx.dims
        G = G.contiguous().float()
        if ctx.kind == 'cosine':
            rx, ry = ctx.saved_tensors[3:]
            # C = 1 - xhat.yhat: dL/dxhat_i = -sum_j G_ij yhat_j; q_i = xhat_i . dL/dxhat_i = -sum_j G_ij (1 - C_ij)
            Wx, Wy = -G * ry[:ny][None, :], -G.t() * rx[:nx][None, :]
            qx, qy = -(G * (1.0 - C)).sum(1), -(G * (1.0 - C)).sum(0)
            sx, sy = rx, ry
        else:
            # C = sqrt(max(m, 1e-6) / D), m = |x|^2 + |y|^2 - 2 x.y: dL/dm = G / (2 D C) where the clamp passes (m >= 1e-6;
            # tf.maximum sends a tie to its first argument); dm/dx_i = 2 x_i - 2 y_j
            passes = (C * C * float(d)) > 1e-6 * (1.0 + 1e-6)
            Gm = torch.where(passes, G / (2.0 * float(d) * C), torch.zeros_like(G))
            Wx, Wy = -2.0 * Gm, -2.0 * Gm.t()
            qx, qy = -2.0 * Gm.sum(1), -2.0 * Gm.sum(0)
            sx = torch.ones(bx.shape[0], dtype=torch.float32, device=bx.device)
            sy = torch.ones(by.shape[0], dtype=torch.float32, device=by.device)
        out = []
        for need, W, q, rows, other, r, n, k, shape in ((ctx.needs_input_grad[0], Wx, qx, bx, by, sx, nx, ny, ctx.shapes[0]),
                                                       (ctx.needs_input_grad[1], Wy, qy, by, bx, sy, ny, nx, ctx.shapes[1])):
            if not need:
                out.append(None)
                continue
            Wp = torch.zeros((_ops.pad32(n) + 32, int(other.shape[0])), dtype=torch.float32, device=G.device)
            Wp[:n, :k] = W
            qp = torch.zeros(_ops.pad32(n), dtype=torch.float32, device=G.device); qp[:n] = q
            dx = torch.zeros_like(rows)
            _ops.rows_gemm_bwd(Wp, k, other, rows, r, qp, n, 1.0, dx)
            out.append(dx[:n, :d].reshape(shape))
        return out[0], out[1], None


def cosine_distance(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """1 - l2_normalize(x) @ l2_normalize(y)^T  (reference losses.py:12-15); differentiable in both arguments."""
    if x.requires_grad or y.requires_grad:
        return _PairwiseDistance.apply(x, y, 'cosine')
    bx, by = _buf(x), _buf(y)
    nx, ny = x.shape[0], y.shape[0]
    return _ops.cosine_distance(bx, _ops.row_inv_norm(bx, nx), nx, by, _ops.row_inv_norm(by, ny), ny)[:, :ny]


def l2_distance(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """sqrt(max(|x|^2 + |y|^2 - 2 x y^T, 1e-6) / D)  (reference losses.py:18-24), on the f32 MFMA; differentiable in both
    arguments."""
    if x.requires_grad or y.requires_grad:
        return _PairwiseDistance.apply(x, y, 'l2')
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
    """loss(pred) for a fixed other side: ONE fused forward+backward pass stores d(loss)/d(pred)."""

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


class _TwoSidedLoss(torch.autograd.Function):
    """loss(first, second) with gradients to whichever arguments need them (TF differentiates both sides of
    losses.py:39-80).  `run(other_buf, n_other, pred_buf, n, d, g, loss, swapped)` is the fused forward+backward pass of
    the loss for the side passed as `pred`.  All three losses are symmetric functions of their two row sets, so the
    gradient w.r.t. the side the kernels call the target is the same pass with the roles exchanged -- `swapped` tells
    the relaxed-EMD kernels so (tf.maximum's tie rule is the one asymmetry)."""

    @staticmethod
    def forward(ctx, first, second, run, pred_is_first):
        f2, s2 = reshape_2d(first), reshape_2d(second)
        bf, bs = _buf(f2), _buf(s2)
        nf, ns_, d = int(f2.shape[0]), int(s2.shape[0]), int(f2.shape[1])
        loss = torch.zeros(1, dtype=torch.float32, device=f2.device)
        gf = gs = None
        # the natural direction first (its value is the one returned), then the exchanged one if asked for
        order = [(True, False), (False, True)] if pred_is_first else [(False, False), (True, True)]
        value = None
        for first_is_pred, swapped in order:
            need = first.requires_grad if first_is_pred else second.requires_grad
            if value is not None and not need:
                continue
            g = torch.zeros_like(bf if first_is_pred else bs)
            l = torch.zeros(1, dtype=torch.float32, device=f2.device)
            if first_is_pred:
                run(bs, ns_, bf, nf, d, g, l, swapped)
                gf = g
            else:
                run(bf, nf, bs, ns_, d, g, l, swapped)
                gs = g
            if value is None:
                value = l
        ctx.dims = (nf, ns_, d, tuple(first.shape), tuple(second.shape))
        ctx.have = (gf is not None, gs is not None)
        ctx.save_for_backward(*[t for t in (gf, gs) if t is not None])
        return value[0]

    @staticmethod
    def backward(ctx, gl):
        nf, ns_, d, shf, shs = ctx.dims
        saved = list(ctx.saved_tensors)
        gf = saved.pop(0) if ctx.have[0] else None
        gs = saved.pop(0) if ctx.have[1] else None
        return (None if gf is None or not ctx.needs_input_grad[0] else (gf[:nf, :d] * gl).reshape(shf),
                None if gs is None or not ctx.needs_input_grad[1] else (gs[:ns_, :d] * gl).reshape(shs), None, None)


def moment_matching(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """mae(cov x, cov y) + mae(mean x, mean y)  (reference losses.py:39-52); x = target, y = prediction in run_strotss.py,
    differentiable in both."""
    def run(other, n_other, pred, n, d, g, loss, swapped):
        mean, cov = _ops.moment_stats(other, n_other, d)
        _ops.moment_fwd_bwd(mean, cov, pred, n, d, 1.0, g, loss)
    return _TwoSidedLoss.apply(x, y, run, False)


def self_similarity(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """reference losses.py:55-66; x = prediction, y = target (ContentLoss swaps them); differentiable in both."""
    def run(other, n_other, pred, n, d, g, loss, swapped):
        _ops.selfsim_fwd_bwd(pred, other, n, d, 1.0, g, loss)
    return _TwoSidedLoss.apply(x, y, run, True)


def relaxed_emd(x: torch.Tensor, y: torch.Tensor, distance: str = 'cosine') -> torch.Tensor:
    """max(mean_i min_j C, mean_j min_i C)  (reference losses.py:69-80); x = target, y = prediction in run_strotss.py,
    differentiable in both.  Every entry of `dist_metrics` at any width, as the reference (losses.py:27-28, 74): 'cosine' on
    the bf16x3 cost GEMM, 'l2' and 'both' on the f32-MFMA cost GEMM with the distance in its epilogue; width 3 with 'both'
    (the palette term of run_strotss.py:36-39) takes the VALU kernel that never stores the cost matrix."""
    if distance not in dist_metrics:
        raise KeyError(distance)

    def run(other, n_other, pred, n, d, g, loss, swapped):
        if distance == 'cosine':
            _ops.remd_cos_fwd_bwd(other, _ops.row_inv_norm(other, n_other), n_other, pred, n, d, 1.0, g, loss, swapped=swapped)
        elif distance == 'both' and d == 3:
            _ops.palette_remd_fwd_bwd(other, n_other, pred, n, 1.0, g, loss, rgb_to_yuv=False, swapped=swapped)
        else:
            _ops.remd_metric_fwd_bwd(other, n_other, pred, n, d, distance, 1.0, g, loss, swapped=swapped)
    return _TwoSidedLoss.apply(x, y, run, False)


def _no_grad_side(t: torch.Tensor, what: str):
    if t.requires_grad:
        raise NotImplementedError(f"{what}: gradients flow to the prediction argument only")


def sinkhorn_knopp(x: torch.Tensor, y: torch.Tensor, distance: str = 'cosine', l: int = 10,
                   N_iter: int = 30) -> torch.Tensor:
    """Sinkhorn-Knopp transport cost between x (target) and y (prediction), differentiable w.r.t. y.
    BUILD-DEFINED: reference losses.py:83-105 is marked `# TODO: untested`, is never called and cannot execute
    (`tf.ones_like(shape)` on a Python tuple), so there is no behaviour to match; this is its evident intent
    (K = exp(-l M), uniform marginals, N_iter alternating scalings from v = 1, cost sum(u * ((K*M) v)), gradient
    through the iterations), pinned by the float64 autograd restatement oracle.strotss_oracle.sinkhorn_knopp.
    Every entry of `dist_metrics` is provided (round 4: 'l2' / 'both' through strotss_sinkhorn_metric_fwd_bwd)."""
    if not l > 0:
        raise ValueError("l must be greater than 0")
    if distance not in dist_metrics:
        raise KeyError(distance)
    _no_grad_side(x, "sinkhorn_knopp(x, y)")
    bx = _buf(x)
    ns, _ = reshape_2d(x).shape
    if distance != 'cosine':            # 'l2' / 'both' (round 4): cost matrix with the distance in its epilogue, same scalings
        return _FusedLoss.apply(y, lambda bp, n, dd, g, loss: _ops.sinkhorn_metric_fwd_bwd(bx, ns, bp, n, dd, distance, l, N_iter,
                                                                                            1.0, g, loss))
    rs = _ops.row_inv_norm(bx, ns)
    return _FusedLoss.apply(y, lambda bp, n, dd, g, loss: _ops.sinkhorn_cos_fwd_bwd(bx, rs, ns, bp, n, dd, l, N_iter, 1.0,
                                                                                     g, loss))
