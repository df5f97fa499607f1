"""Pins the oracle (parity unpinned against TF -- see oracle/strotss_oracle.py): torch-autograd
restatement vs the independent NumPy closed-form gradients vs finite differences vs known answers."""
import numpy as np
import pytest
import torch

from oracle import numpy_ref as R
from oracle import strotss_oracle as O

torch.set_num_threads(4)


def _feat(n, d, seed, relu=True):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((n, d))
    if relu:
        x = np.maximum(x, 0) + 0.01 * rng.random((n, d))
    x[:, :3] = rng.random((n, 3))        # RGB-like leading channels
    return x


def _autograd(fn, y):
    yt = torch.from_numpy(y).clone().requires_grad_(True)
    l = fn(yt)
    g, = torch.autograd.grad(l, yt)
    return float(l), g.numpy()


@pytest.mark.parametrize("n,ns,d", [(48, 48, 35), (64, 40, 19)])
def test_closed_form_gradients_match_autograd(n, ns, d):
    x = _feat(ns, d, 1); y = _feat(n, d, 2); c = _feat(n, d, 3)
    xt = torch.from_numpy(x); ct = torch.from_numpy(c)

    l, g = R.self_similarity_fwd_bwd(y, c)
    lo, go = _autograd(lambda t: O.self_similarity(t, ct), y)
    assert abs(l - lo) < 1e-12 and np.abs(g - go).max() < 1e-12 * max(1, np.abs(go).max())

    l, g = R.relaxed_emd_cos_fwd_bwd(x, y)
    lo, go = _autograd(lambda t: O.relaxed_emd(xt, t), y)
    assert abs(l - lo) < 1e-12 and np.abs(g - go).max() < 1e-13

    l, g = R.palette_remd_fwd_bwd(x[:, :3], y[:, :3])
    lo, go = _autograd(lambda t: O.relaxed_emd(O.convert_rgb_to_yuv(xt), O.convert_rgb_to_yuv(t), "both"), y)
    assert abs(l - lo) < 1e-12 and np.abs(g - go[:, :3]).max() < 1e-12

    l, g = R.moment_matching_fwd_bwd(x, y)
    lo, go = _autograd(lambda t: O.moment_matching(xt, t), y)
    assert abs(l - lo) < 1e-12 and np.abs(g - go).max() < 1e-14

    for alpha in (16.0, 0.5):
        l, g = R.style_loss_fwd_bwd(x, y, alpha)
        lo, go = _autograd(lambda t: O.style_loss(xt, t, alpha), y)
        assert abs(l - lo) < 1e-12 and np.abs(g - go).max() < 1e-12


def test_finite_difference_gradients():
    n, d = 24, 11
    x = _feat(n, d, 4); y = _feat(n, d, 5); c = _feat(n, d, 6)
    xt = torch.from_numpy(x); ct = torch.from_numpy(c)

    def total(t):
        return 3.0 * O.content_loss(ct, t) + O.style_loss(xt, t, 2.0)

    _, g = _autograd(total, y)
    rng = np.random.default_rng(0)
    eps = 1e-6
    for _ in range(25):
        i, j = rng.integers(n), rng.integers(d)
        yp = y.copy(); yp[i, j] += eps
        ym = y.copy(); ym[i, j] -= eps
        fd = (float(total(torch.from_numpy(yp))) - float(total(torch.from_numpy(ym)))) / (2 * eps)
        assert abs(fd - g[i, j]) < 1e-5 * max(1.0, abs(g[i, j])), (i, j, fd, g[i, j])


def test_known_answers():
    x = torch.from_numpy(_feat(32, 9, 7))
    assert float(O.self_similarity(x, x)) == 0.0
    assert float(O.moment_matching(x, x)) == 0.0
    assert abs(float(O.relaxed_emd(x, x))) < 1e-15
    perm = torch.randperm(32, generator=torch.Generator().manual_seed(0))
    y = torch.from_numpy(_feat(32, 9, 8))
    assert abs(float(O.relaxed_emd(x, y)) - float(O.relaxed_emd(x[perm], y))) < 1e-15
    # cosine distance of orthonormal rows = 1 - I
    e = torch.eye(5, dtype=torch.float64)
    assert torch.allclose(O.cosine_distance(e, e), 1 - e, atol=1e-15)
    # l2_distance(0-row, e_k-row) = sqrt(1/D); clamp: l2_distance(x,x) = sqrt(1e-6/D)
    z = torch.zeros(1, 5, dtype=torch.float64)
    assert torch.allclose(O.l2_distance(z, e), torch.full((1, 5), (1 / 5) ** 0.5, dtype=torch.float64))
    assert torch.allclose(torch.diagonal(O.l2_distance(e, e)), torch.full((5,), (1e-6 / 5) ** 0.5, dtype=torch.float64))
    # rgb_to_yuv(white) = (1, ~0, ~0)
    yuv = O.convert_rgb_to_yuv(torch.ones(1, 7, dtype=torch.float64))
    assert abs(float(yuv[0, 0]) - 1) < 1e-12 and abs(float(yuv[0, 1])) < 1e-7 and abs(float(yuv[0, 2])) < 1e-7
    # l2_normalize clamp: an all-zero row stays zero, no NaN
    zz = torch.zeros(2, 4, dtype=torch.float64)
    assert torch.equal(O.cosine_distance(zz, zz), torch.ones(2, 2, dtype=torch.float64))


def test_remd_tie_rules():
    # two identical style rows: the column-min gradient is split between them (tf.reduce_min),
    # and the max() tie goes to the first argument (tf.maximum: x >= y).
    x = np.array([[1.0, 0.0], [1.0, 0.0], [0.0, 1.0]])
    y = np.array([[0.8, 0.6], [0.1, 0.9]])
    l, g = R.relaxed_emd_cos_fwd_bwd(x, y)
    lo, go = _autograd(lambda t: O.relaxed_emd(torch.from_numpy(x), t), y)
    assert abs(l - lo) < 1e-15 and np.abs(g - go).max() < 1e-15
    # symmetric problem -> R_X == R_Y exactly -> gradient follows the row-min branch
    x = np.array([[1.0, 0.0], [0.0, 1.0]]); y = np.array([[0.6, 0.8], [0.8, 0.6]])
    C = R.cosine_distance(x, y)
    assert C.min(1).mean() == C.min(0).mean()
    l, g = R.relaxed_emd_cos_fwd_bwd(x, y)
    lo, go = _autograd(lambda t: O.relaxed_emd(torch.from_numpy(x), t), y)
    assert np.abs(g - go).max() < 1e-15


def test_remd_exact_tie_between_branches_is_asymmetric():
    """R_X == R_Y exactly, yet the two branches weigh the entries differently (row 0 ties over three columns): the
    gradient must be the ROW branch's (tf.maximum -> first argument), in both restatements."""
    d = 8
    x = np.zeros((2, d)); x[0, 0] = 1.0; x[1, 4] = 1.0
    y = np.zeros((4, d)); y[:3, 0:4] = 1.0; y[3, 4:8] = 1.0
    C = R.cosine_distance(x, y)
    assert C.min(1).mean() == C.min(0).mean() == 0.5
    l, g = R.relaxed_emd_cos_fwd_bwd(x, y)
    lo, go = _autograd(lambda t: O.relaxed_emd(torch.from_numpy(x), t), y)
    assert l == lo == 0.5 and np.abs(g - go).max() < 1e-15
    # hand-computed row-branch weights: W = [[1/6,1/6,1/6,0],[0,0,0,1/2]]; dL/dyhat_j = -sum_i W_ij xhat_i
    W = np.array([[1 / 6, 1 / 6, 1 / 6, 0], [0, 0, 0, 1 / 2]])
    ghat = -(W.T @ x)
    ry = R.inv_norm(y)
    yh = y * ry[:, None]
    want = ry[:, None] * (ghat - yh * (yh * ghat).sum(1, keepdims=True))
    assert np.abs(g - want).max() < 1e-15
    Wc = (C == C.min(0, keepdims=True)) / 4.0                       # the column branch would give this instead
    assert np.abs(W - Wc).max() > 0.08


def test_rmsprop_first_step():
    v = torch.zeros(5, dtype=torch.float64); r = torch.zeros(5, dtype=torch.float64)
    g = torch.tensor([1.0, -2.0, 0.5, 1e-3, -7.0], dtype=torch.float64)
    O.rmsprop_update(v, r, g, 2e-3)
    assert torch.allclose(v, -2e-3 * g / (0.1 * g.abs() + 1e-8), rtol=1e-12)
    # identical to torch.optim.RMSprop(alpha=.99, eps=1e-8)
    p = torch.zeros(5, dtype=torch.float64, requires_grad=True)
    opt = torch.optim.RMSprop([p], lr=2e-3, alpha=0.99, eps=1e-8)
    v2 = torch.zeros(5, dtype=torch.float64); r2 = torch.zeros(5, dtype=torch.float64)
    for k in range(3):
        gk = g * (k + 1)
        p.grad = gk.clone(); opt.step()
        O.rmsprop_update(v2, r2, gk, 2e-3)
    assert torch.allclose(p.detach(), v2, rtol=1e-12, atol=0)


def test_sinkhorn_oracle_properties():
    """The build-defined Sinkhorn restatement: marginals of the plan after the last v-update, permutation
    invariance, and the small-regularisation limit approaching the relaxed EMD lower bound from above."""
    g = torch.Generator().manual_seed(0)
    x = torch.rand(40, 16, generator=g, dtype=torch.float64) + 0.05
    y = torch.rand(48, 16, generator=g, dtype=torch.float64) + 0.05
    c = O.sinkhorn_knopp(x, y, 'cosine', 10.0, 200)
    perm = torch.randperm(48, generator=g)
    assert abs(float(O.sinkhorn_knopp(x, y[perm], 'cosine', 10.0, 200)) - float(c)) < 1e-12
    M = O.cosine_distance(x, y)
    K = torch.exp(-10.0 * M)
    u = torch.ones(40, 1, dtype=torch.float64); v = torch.ones(48, 1, dtype=torch.float64)
    for _ in range(200):
        u = (1 / 40) / (K @ v); v = (1 / 48) / (K.t() @ u)
    P = u * K * v.t()
    assert (P.sum(0) - 1 / 48).abs().max() < 1e-12 and (P.sum(1) - 1 / 40).abs().max() < 1e-6
    assert abs(float((P * M).sum()) - float(c)) < 1e-12
    # a transport plan with uniform marginals can never cost less than either relaxed bound
    assert float(c) >= float(torch.amin(M, 1).mean()) - 1e-12 and float(c) >= float(torch.amin(M, 0).mean()) - 1e-12
