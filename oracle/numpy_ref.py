"""Independent NumPy float64 restatement of the STROTSS losses WITH hand-derived gradients.

TEST INFRASTRUCTURE (see oracle/strotss_oracle.py header: parity unpinned).  The torch
oracle differentiates with autograd; this file carries the closed-form backward formulas the
HIP kernels implement.  tests/test_oracle_losses.py checks the two against each other and
against finite differences, so a wrong derivation is caught on the CPU before any kernel is
written.  Reference lines: nn/losses.py:12-80, run_strotss.py:27-40, strotss_utils.py:166-167.
"""
from __future__ import annotations

import numpy as np

RGB2YUV = np.array([[0.299, -0.14714119, 0.61497538],
                    [0.587, -0.28886916, -0.51496512],
                    [0.114, 0.43601035, -0.10001026]], dtype=np.float64)


def inv_norm(x):
    """r_i = rsqrt(max(sum x_i^2, 1e-12))  (tf.nn.l2_normalize, losses.py:13-14)."""
    return 1.0 / np.sqrt(np.maximum((x * x).sum(1), 1e-12))


def cosine_distance(x, y):
    return 1.0 - (x * inv_norm(x)[:, None]) @ (y * inv_norm(y)[:, None]).T


def l2_distance(x, y):
    m = (x * x).sum(1)[:, None] + (y * y).sum(1)[None, :] - 2.0 * x @ y.T
    return np.sqrt(np.maximum(m, 1e-6) / x.shape[1])


def unnormalise_grad(x, r, g_hat, q=None):
    """dL/dx from g_hat = dL/dx_hat, x_hat = x*r.   dx = r*(g - x_hat*(x_hat.g)) where the
    row norm is not clamped; q may carry the pre-computed x_hat.g row dots."""
    xh = x * r[:, None]
    if q is None:
        q = (xh * g_hat).sum(1)
    live = ((x * x).sum(1) >= 1e-12)
    return r[:, None] * (g_hat - xh * (q * live)[:, None])


# ---------------------------------------------------------------- self similarity
def self_similarity_fwd_bwd(x, y):
    """loss = self_similarity(x, y) (losses.py:55-66) and dloss/dx."""
    n = y.shape[0]
    rx = inv_norm(x)
    Dx = cosine_distance(x, x)
    Dy = cosine_distance(y, y)
    sx_raw = Dx.sum(0); sy_raw = Dy.sum(0)
    sx = np.maximum(sx_raw, 1e-12); sy = np.maximum(sy_raw, 1e-12)
    A = Dx / sx[None, :]
    B = Dy / sy[None, :]
    loss = np.abs(A - B).mean() * n
    S = np.sign(A - B) * (n / A.size)                # dL/dA
    t = (S * A).sum(0) * (sx_raw >= 1e-12)           # column term of the normalisation
    Gd = (S - t[None, :]) / sx[None, :]              # dL/dDx
    M = -(Gd + Gd.T)                                 # dL/dG symmetrised: dXhat = M @ Xhat
    q = (M * (1.0 - Dx)).sum(1)                      # xhat_i . g_i without a D-length dot
    g_hat = (M * rx[None, :]) @ x                    # = M @ xhat
    dx = unnormalise_grad(x, rx, g_hat, q)
    return loss, dx


# ---------------------------------------------------------------- relaxed EMD
def _remd_weights(C):
    """W = dL/dC for L = max(mean rowmin, mean colmin) with TF's tie rules."""
    ns, n = C.shape
    rmin = C.min(1); cmin = C.min(0)
    r_x = rmin.mean(); r_y = cmin.mean()
    if r_x >= r_y:
        E = (C == rmin[:, None]).astype(np.float64)
        W = E / E.sum(1, keepdims=True) / ns
        return r_x, W
    E = (C == cmin[None, :]).astype(np.float64)
    W = E / E.sum(0, keepdims=True) / n
    return r_y, W


def relaxed_emd_cos_fwd_bwd(x, y):
    """relaxed_emd(x, y, 'cosine') (losses.py:69-80) and dloss/dy."""
    rx = inv_norm(x); ry = inv_norm(y)
    xh = x * rx[:, None]
    C = cosine_distance(x, y)
    loss, W = _remd_weights(C)
    g_hat = -(W.T @ xh)                               # dL/dyhat
    q = -(W * (1.0 - C)).sum(0)                       # yhat_j . g_j
    dy = unnormalise_grad(y, ry, g_hat, q)
    return loss, dy


def palette_remd_fwd_bwd(x_rgb, y_rgb):
    """relaxed_emd(yuv(x), yuv(y), 'both') (run_strotss.py:37-39) and dloss/dy_rgb."""
    x = x_rgb @ RGB2YUV
    y = y_rgb @ RGB2YUV
    rx = inv_norm(x); ry = inv_norm(y)
    xh = x * rx[:, None]
    Cc = cosine_distance(x, y)
    m = (x * x).sum(1)[:, None] + (y * y).sum(1)[None, :] - 2.0 * x @ y.T
    L2 = np.sqrt(np.maximum(m, 1e-6) / 3.0)
    C = Cc + L2
    loss, W = _remd_weights(C)
    # cosine part
    g_hat = -(W.T @ xh)
    q = -(W * (1.0 - Cc)).sum(0)
    dy = unnormalise_grad(y, ry, g_hat, q)
    # l2 part: d l2 / d y_j = (y_j - x_i) / (3 l2)   where m >= 1e-6
    K = W * (m >= 1e-6) / (3.0 * L2)
    dy = dy + K.sum(0)[:, None] * y - K.T @ x
    return loss, dy @ RGB2YUV.T


# ---------------------------------------------------------------- moment matching
def moment_matching_fwd_bwd(x, y):
    """moment_matching(x, y) (losses.py:39-52) and dloss/dy."""
    n, d = y.shape
    mx = x.mean(0); my = y.mean(0)
    cx = x - mx; cy = y - my
    Sx = cx.T @ cx / x.shape[0]
    Sy = cy.T @ cy / n
    loss = np.abs(Sx - Sy).mean() + np.abs(mx - my).mean()
    T = np.sign(Sy - Sx) / (d * d)
    dcy = cy @ (T + T.T) / n
    dy = dcy - dcy.mean(0, keepdims=True)
    dy = dy + np.sign(my - mx)[None, :] / (d * n)
    return loss, dy


def style_loss_fwd_bwd(target, pred, alpha):
    """StyleLoss.__call__ (run_strotss.py:33-40) and d/dpred."""
    inv_alpha = 1.0 / max(alpha, 1.0)
    l_m, g_m = moment_matching_fwd_bwd(target, pred)
    l_r, g_r = relaxed_emd_cos_fwd_bwd(target, pred)
    l_p, g_p = palette_remd_fwd_bwd(target[:, :3], pred[:, :3])
    g = g_m + g_r
    g[:, :3] += inv_alpha * g_p
    return l_m + l_r + inv_alpha * l_p, g
