"""GPU: size-independent properties of the HIP path at BASELINE.json's FULL sizes (1024x1024 images,
N = 1024 samples, D = 2179), where the float64 oracle is too slow to be the checker:
adjointness (<A x, y> == <x, A^T y>) of every linear operator pair, exact identities
(self_similarity(x,x) = 0, moment_matching(x,x) = 0, fold(make_pyramid(x)) = x), invariances
(permutation of the samples), idempotence / sanity of RMSprop and postprocess, and the
direct-vs-Winograd agreement of the conv kernels."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
N, D, S = 1024, 2179, 1024


def _rand(*shape, seed=0, relu=False):
    g = torch.Generator(device="cpu").manual_seed(seed)
    x = torch.randn(*shape, generator=g)
    return (torch.relu(x) if relu else x).to(DEV)


def _dot(a, b):
    return float((a.double() * b.double()).sum())


def _feat(seed):
    from nn import _ops
    x = torch.zeros(_ops.pad32(N), _ops.pad32(D), device=DEV)
    x[:N, :D] = _rand(N, D, seed=seed, relu=True) + 0.01
    x[:N, :3] = torch.rand(N, 3, generator=torch.Generator().manual_seed(seed)).to(DEV)
    return x


def test_pyramid_roundtrip_and_fold_adjoint_1024():
    from nn import _ops, strotss_utils as SU
    x = torch.rand(1, S, S, 3, generator=torch.Generator().manual_seed(1)).to(DEV)
    pyr = SU.make_laplacian_pyramid(x)
    assert [tuple(p.shape[1:3]) for p in pyr] == [(1024, 1024), (512, 512), (256, 256), (128, 128), (64, 64), (32, 32)]
    assert float((SU.fold_laplacian_pyramid(pyr) - x).abs().max()) < 2e-6
    # resize and its adjoint at full size: <U a, b> == <a, U^T b>
    a = _rand(1, 512, 512, 3, seed=2); b = _rand(1, S, S, 3, seed=3)
    lhs = _dot(_ops.resize_bilinear(a, S, S), b)
    rhs = _dot(a, _ops.resize_bilinear_adjoint(b, 512, 512))
    assert abs(lhs - rhs) < 1e-5 * max(1.0, abs(lhs))


@pytest.mark.parametrize("cfg", [(1024, 64, 64), (512, 64, 128), (256, 256, 256), (128, 512, 512)])
def test_conv_adjoint_and_winograd_agreement_fullsize(cfg):
    """<conv(x), y> == <x, conv^T(y)> for the kernels at the VGG16 shapes of a 1024-px image; the
    Winograd entry points agree with the direct ones."""
    from nn import _ops
    hw, cin, cout = cfg
    g = torch.Generator().manual_seed(hw + cin)
    wt = torch.randn(3, 3, cin, cout, generator=g) * (2.0 / (9 * cin)) ** 0.5
    x = _rand(1, hw, hw, cin, seed=5); y = _rand(1, hw, hw, cout, seed=6)
    w_f = wt.permute(0, 1, 3, 2).reshape(9, cout, cin).contiguous().to(DEV)
    w_b = wt.flip(0, 1).reshape(9, cin, cout).contiguous().to(DEV)
    zero_b = torch.full((cout,), -1e30, device=DEV)        # not used: the adjoint check needs the LINEAR map
    # linear part through the dgrad kernel in both directions (no bias / ReLU there)
    cx = _ops.conv3x3_dgrad(x, w_f, cout)                   # = conv(x) as a "dgrad" with the forward weights
    ct = _ops.conv3x3_dgrad(y, w_b, cin)                    # = conv^T(y)
    lhs, rhs = _dot(cx, y), _dot(x, ct)
    assert abs(lhs - rhs) < 2e-5 * max(1.0, abs(lhs)), (lhs, rhs)
    if cin >= 128:
        u_f = _ops.winograd_weights(wt.permute(3, 2, 0, 1)).to(DEV)
        u_b = _ops.winograd_weights(wt.flip(0, 1).permute(2, 3, 0, 1)).to(DEV)
        wx = _ops.conv3x3_winograd_dgrad(x, u_f, cout)
        assert float((wx - cx).abs().max()) < 2e-5 * float(cx.abs().max())
        wty = _ops.conv3x3_winograd_dgrad(y, u_b, cin)
        assert float((wty - ct).abs().max()) < 2e-5 * float(ct.abs().max())
    # forward entry == linear part + bias, ReLU
    b = (torch.randn(cout, generator=g) * 0.1).to(DEV)
    fwd = _ops.conv3x3_relu_fwd(x, w_f, b)
    assert float((fwd - torch.relu(cx + b)).abs().max()) < 1e-5 * float(cx.abs().max())
    del zero_b


def test_first_layer_and_pool_adjoint_1024():
    from nn import _ops
    g = torch.Generator().manual_seed(9)
    wt = torch.randn(3, 3, 3, 64, generator=g) * 0.3
    img = torch.rand(1, S, S, 3, generator=g).to(DEV)
    y = _rand(1, S, S, 64, seed=10)
    b0 = torch.zeros(64, device=DEV)
    # pre-ReLU linear map L(img) = conv((img - mean)/std); check <L(a) - L(b), y> == <a - b, L^T y>
    a2 = torch.rand(1, S, S, 3, generator=g).to(DEV)
    big = torch.full((64,), 1e3, device=DEV)                # bias large enough that ReLU is the identity
    La = _ops.conv3x3_c3_fwd(img, wt.reshape(27, 64).to(DEV), big) - 1e3
    Lb = _ops.conv3x3_c3_fwd(a2, wt.reshape(27, 64).to(DEV), big) - 1e3
    gt = _ops.conv3x3_c3_dgrad(y, wt.flip(0, 1).reshape(9, 3, 64).contiguous().to(DEV))
    lhs, rhs = _dot(La - Lb, y), _dot(img - a2, gt)
    assert abs(lhs - rhs) < 5e-3 * max(1.0, abs(lhs)), (lhs, rhs)      # 1e3 offset costs ~4 digits of the f32 sums
    del b0
    # max-pool: routed gradient sums to the pooled gradient wherever the window max is positive
    act = _rand(1, S, S, 64, seed=11, relu=True)
    gp = _rand(1, S // 2, S // 2, 64, seed=12)
    gin = _ops.maxpool2_bwd(act, gp)
    pooled = _ops.maxpool2_fwd(act)
    s4 = gin.view(1, S // 2, 2, S // 2, 2, 64).sum(dim=(2, 4))
    assert float((s4 - gp * (pooled > 0)).abs().max()) < 1e-6


def test_gather_scatter_adjoint_fullsize():
    from nn import _ops
    shapes = [(S, S, 3), (S, S, 64), (S, S, 64), (S // 2, S // 2, 128), (S // 2, S // 2, 128), (S // 4, S // 4, 256),
              (S // 4, S // 4, 256), (S // 4, S // 4, 256), (S // 8, S // 8, 512), (S // 16, S // 16, 512)]
    maps = [_rand(1, *s, seed=20 + i) for i, s in enumerate(shapes)]
    rng = np.random.default_rng(0)
    from nn import strotss_utils as SU
    idx = torch.from_numpy(SU.make_indices_np(S, S, True, N, rng)).to(DEV)
    f = _ops.hypercol_gather(maps, idx, True)
    assert tuple(f.shape) == (1024, 2208) and float(f[:, D:].abs().sum()) == 0
    gf = torch.zeros_like(f); gf[:N, :D] = _rand(N, D, seed=40)
    gm = [torch.zeros_like(m) for m in maps]
    _ops.hypercol_scatter(maps, gm, idx, gf, relu_mask_from=len(maps))
    lhs = _dot(f, gf)
    rhs = sum(_dot(m, g) for m, g in zip(maps, gm))
    assert abs(lhs - rhs) < 1e-4 * max(1.0, abs(lhs)), (lhs, rhs)


def test_loss_identities_fullsize():
    from nn import _ops
    x, y = _feat(1), _feat(2)
    loss = torch.zeros(8, device=DEV)
    g = torch.zeros_like(x)
    # self_similarity(x, x) == 0 exactly (identical cost matrices), gradient entries tiny
    _ops.selfsim_fwd_bwd(x, x, N, D, 1.0, g, loss[0:])
    assert float(loss[0]) == 0.0
    # moment_matching(x, x) == 0 exactly
    mean, cov = _ops.moment_stats(x, N, D)
    g.zero_(); _ops.moment_fwd_bwd(mean, cov, x, N, D, 1.0, g, loss[1:])
    assert float(loss[1]) == 0.0 and float(g.abs().max()) == 0.0
    # relaxed_emd(x, x) ~ 0: |1 - <xhat_i, xhat_j>| at f32 rounding of a 2179-term all-positive dot product whose
    # norm comes from a differently ordered sum (K*eps/2 = 1.3e-4 worst case, ~1e-5 observed)
    rs = _ops.row_inv_norm(x, N)
    g.zero_(); _ops.remd_cos_fwd_bwd(x, rs, N, x, N, D, 1.0, g, loss[2:])
    assert abs(float(loss[2])) < 5e-5
    # permutation invariance of the three losses in the sample order of the prediction
    perm = torch.randperm(N, generator=torch.Generator().manual_seed(3)).to(DEV)
    yp = torch.zeros_like(y); yp[:N] = y[:N][perm]
    vals = []
    for pred in (y, yp):
        l = torch.zeros(8, device=DEV); gg = torch.zeros_like(x)
        _ops.moment_fwd_bwd(mean, cov, pred, N, D, 1.0, gg, l[0:])
        _ops.remd_cos_fwd_bwd(x, rs, N, pred, N, D, 1.0, gg, l[1:])
        _ops.palette_remd_fwd_bwd(x, N, pred, N, 1.0, gg, l[2:])
        vals.append(l[:3].cpu().numpy().astype(np.float64))
    assert np.abs(vals[0] - vals[1]).max() < 2e-6 * max(1.0, np.abs(vals[0]).max())
    # cosine cost matrix at full size: bitwise symmetric, zero diagonal to rounding, range [0, 2]
    Dm = _ops.cosine_distance(x, rs, N, x, rs, N)[:, :N]
    assert torch.equal(Dm, Dm.T) and float(Dm.diagonal().abs().max()) < 5e-5
    assert float(Dm.min()) > -5e-5 and float(Dm.max()) < 2.0


def test_loss_gradient_is_directional_derivative_fullsize():
    """d/dt L(y + t v) at t = 0 by central differences in float64-accumulated f32 losses vs <grad, v>
    for the smooth moment term (the L1 / hard-min terms are only piecewise smooth)."""
    from nn import _ops
    x, y = _feat(5), _feat(6)
    mean, cov = _ops.moment_stats(x, N, D)
    g = torch.zeros_like(y); l = torch.zeros(8, device=DEV)
    _ops.moment_fwd_bwd(mean, cov, y, N, D, 1.0, g, l)
    v = torch.zeros_like(y); v[:N, :D] = _rand(N, D, seed=7)
    t = 1e-2
    lp = torch.zeros(8, device=DEV); lm = torch.zeros(8, device=DEV); tmp = torch.zeros_like(y)
    _ops.moment_fwd_bwd(mean, cov, y + t * v, N, D, 1.0, tmp, lp)
    _ops.moment_fwd_bwd(mean, cov, y - t * v, N, D, 1.0, tmp, lm)
    fd = (float(lp[0]) - float(lm[0])) / (2 * t)
    an = _dot(g, v)
    assert abs(fd - an) < 5e-2 * max(abs(an), 1e-6), (fd, an)


def test_rmsprop_and_postprocess_properties_1024():
    from nn import _ops
    shapes = [(1, 1024, 1024, 3), (1, 512, 512, 3), (1, 256, 256, 3), (1, 128, 128, 3), (1, 64, 64, 3), (1, 32, 32, 3)]
    v = [_rand(*s, seed=50 + i) for i, s in enumerate(shapes)]
    v0 = [t.clone() for t in v]
    r = [torch.zeros_like(t) for t in v]
    gr = [_rand(*s, seed=60 + i) for i, s in enumerate(shapes)]
    _ops.rmsprop_step(v, r, gr, 2e-3)
    for a, a0, g_, rr in zip(v, v0, gr, r):
        # first step from zero slots: -lr * g / (0.1 |g| + 1e-8)
        ref = a0 - 2e-3 * g_ / (0.1 * g_.abs() + 1e-8)
        assert float((a - ref).abs().max()) < 2e-6
        assert float((rr - 0.01 * g_ * g_).abs().max()) < 1e-6 * float((g_ * g_).max())
    img = torch.rand(1, S, S, 3, generator=torch.Generator().manual_seed(1)).to(DEV) * 1.5 - 0.25
    u8 = _ops.postprocess(img)
    assert u8.dtype == torch.uint8 and int(u8.min()) == 0 and int(u8.max()) == 255
    # idempotent up to the truncating cast: postprocess(u8/255) == u8
    again = _ops.postprocess(u8.float() / 255.0)
    assert int((again.int() - u8.int()).abs().max()) <= 1


@pytest.mark.parametrize("scale,regions", [(1024, 1), (512, 1), (1024, 4)])
def test_engine_step_invariants_fullsize(scale, regions):
    """One full step at the bench configurations -- BASELINE config 3 (1024 px), config 2's last scale (512 px) and
    config 4's masked step (1024 px, 4 mask regions: one trunk pass, four loss groups): finite losses, the level-k
    gradient is the bilinear adjoint of the level-(k-1) gradient, the update moves every variable by at most 10*lr,
    and the pixel gradient is the directional derivative of the logged loss (central difference along a random
    direction of the finest variable: checks fold + trunk + gather + losses + their backward as a whole)."""
    import bench
    from nn import _ops
    from nn.model import VGGParams, synthetic_weights
    params = VGGParams(synthetic_weights('16', 0), '16', None, DEV)
    eng, rng = bench.build_engine(params, scale, torch.device(DEV), seed=0, regions=regions)
    assert eng.R == regions
    idx = bench.index_stream(scale, 2, rng, torch.device(DEV), regions=regions)
    before = [t.clone() for t in eng.variables]
    # directional derivative first (forward_backward does not move the variables)
    eng.forward_backward(list(idx[0]))
    g0 = eng.gvars[0].clone()
    # along the gradient itself (scaled to unit max): a random direction's derivative drowns in the f32 rounding of the
    # logged loss at this size
    dirn = g0 / g0.abs().max()
    want = _dot(g0, dirn)
    eps = 2e-3
    vals = []
    for sgn in (1.0, -1.0):
        eng.variables[0].copy_(before[0] + sgn * eps * dirn)
        eng.forward_backward(list(idx[0]))
        vals.append(eng.losses()["loss"])
    eng.variables[0].copy_(before[0])
    fd = (vals[0] - vals[1]) / (2 * eps)
    assert want > 0 and abs(fd - want) < 0.1 * want, (fd, want)       # L1 / hard-min terms are piecewise smooth
    eng.step(list(idx[0]))
    torch.cuda.synchronize()
    ls = eng.losses()
    assert all(np.isfinite(v) and v >= 0 for v in ls.values()), ls
    for k in range(1, 6):
        hk, wk = eng.sizes[k]
        ref = _ops.resize_bilinear_adjoint(eng.gvars[k - 1], hk, wk)
        assert torch.equal(ref, eng.gvars[k])
    for a, b in zip(eng.variables, before):
        assert float((a - b).abs().max()) <= 10 * eng.lr * 1.0001


def test_pyramid_config2_runs_all_four_scales():
    """BASELINE config 2: a 512-px pair through the full 4-scale pyramid (64 -> 512) of the CLI driver, with the
    reference's default 200 steps per scale; checks the per-scale sizes and alpha schedule, that every scale lowers
    its loss, and the output."""
    import os, tempfile
    import run_strotss as RS
    from nn import utils
    with tempfile.TemporaryDirectory() as tmp:
        paths = []
        for name, seed in (("c.jpg", 100), ("s.jpg", 200)):
            paths.append(os.path.join(tmp, name))
            utils.write_image(bench_image(512, seed) * 255.0, paths[-1])
        args = RS.build_parser().parse_args(paths + ["-o", os.path.join(tmp, "o.jpg"), "--max_size", "512", "--level", "4",
                                                     "--max_iter", "200", "--log_every", "200"])
        tr = []
        final = RS.run(args, trace=tr)
    assert [t["scl"] for t in tr] == [64, 128, 256, 512] and [t["hw"] for t in tr] == [(64, 64), (128, 128), (256, 256), (512, 512)]
    assert [t["alpha"] for t in tr] == [16.0, 8.0, 4.0, 2.0] and [t["lr"] for t in tr] == [2e-3, 2e-3, 2e-3, 1e-3]
    for t in tr:
        assert len(t["steps"]) == 200
        head = np.mean([s_["loss"] for s_ in t["steps"][:5]]); tail = np.mean([s_["loss"] for s_ in t["steps"][-20:]])
        assert np.isfinite(tail) and tail < head, (t["scl"], head, tail)
    assert tuple(final.shape) == (512, 512, 3) and final.dtype == torch.uint8


def bench_image(size, seed):
    import bench
    return bench.synth_image(size, size, seed)


def test_sinkhorn_fullsize_permutation_invariance_and_bounds():
    """N = 1024, D = 2179: the Sinkhorn cost does not depend on the order of the prediction rows, its gradient
    permutes with them, and a plan with uniform marginals costs at least the relaxed-EMD bound."""
    from nn import losses as L
    x, y = _feat(11)[:N, :D].contiguous(), _feat(12)[:N, :D].contiguous()
    perm = torch.randperm(N, generator=torch.Generator().manual_seed(3)).to(DEV)
    ya = y.clone().requires_grad_(True)
    yb = y[perm].clone().requires_grad_(True)
    ca, cb = L.sinkhorn_knopp(x, ya), L.sinkhorn_knopp(x, yb)
    ca.backward(); cb.backward()
    assert abs(float(ca) - float(cb)) < 1e-4 * abs(float(ca))
    rel = float((ya.grad[perm] - yb.grad).norm() / yb.grad.norm())
    assert rel < 1e-3, rel
    assert float(ca) >= float(L.relaxed_emd(x, y)) - 1e-5


@pytest.mark.parametrize("cfg", [(1024, 64, 64), (512, 128, 128), (256, 256, 256), (128, 512, 512)])
def test_relu_sign_words_fullsize(cfg):
    """Sign words at the VGG16 shapes of a 1024-px image, whichever kernel the route ends in: the words a forward pass writes
    from its registers == the words derived from the activation it wrote (every tile, every channel), and the data-gradient
    masked by them == the data-gradient masked by the activation, bit for bit (include/strotss_hip.h: strotss_relu_bits)."""
    from nn import _ops
    hw, cin, cout = cfg
    g = torch.Generator().manual_seed(hw * 3 + cin)
    wt = torch.randn(3, 3, cin, cout, generator=g) * (2.0 / (9 * cin)) ** 0.5
    x = _rand(1, hw, hw, cin, seed=7, relu=True)
    bias = (torch.randn(cout, generator=g) * 0.1).to(DEV)
    u_f = _ops.winograd_weights(wt.permute(3, 2, 0, 1).contiguous().to(DEV), 4, DEV)
    bits = _ops.relu_bits_buffer(hw, hw, cout, DEV)
    act = _ops.conv3x3_winograd_fwd(x, u_f, bias, relu_bits_out=bits)
    assert torch.equal(bits, _ops.relu_bits(act))                      # hw % 4 == 0: no unspecified bits
    frac = float((act > 0).float().mean())
    assert 0.2 < frac < 0.8, frac                                       # (the pattern is not trivial)
    # the NEXT layer's data-gradient: gradient (hw, hw, cout2) back to this activation, masked by its sign
    u_b = _ops.winograd_weights(wt.flip(0, 1).permute(2, 3, 0, 1).contiguous().to(DEV), 4, DEV)   # (cin, cout) kernel: cout -> cin
    gy = _rand(1, hw, hw, cout, seed=8)
    xbits = _ops.relu_bits(x)
    a = _ops.conv3x3_winograd_dgrad(gy, u_b, cin, act_in=x)
    b = _ops.conv3x3_winograd_dgrad(gy, u_b, cin, relu_bits=xbits)
    assert torch.equal(a, b)
    # accumulate (ABI 7, the pre-scatter backward): gin += the same masked gradient -- ONE f32 addition per element
    pre = _rand(1, hw, hw, cin, seed=9)
    for kw in ({"act_in": x}, {"relu_bits": xbits}):
        c = pre.clone()
        _ops.conv3x3_winograd_dgrad(gy, u_b, cin, out=c, accumulate=True, **kw)
        assert torch.equal(c, pre + a), kw.keys()


@pytest.mark.parametrize("cfg", [(64, 256, 256), (32, 512, 512), (62, 128, 256)])
def test_winograd_dgrad_accumulate_small_maps(cfg):
    """The same on the maps of the 256 / 512-px scales, where the F(4x4,3x3) data-gradient runs as three kernels with f32 GEMMs
    (and a ragged 62 x 62 map): out += the masked gradient, bit for bit one addition."""
    from nn import _ops
    hw, cin, cout = cfg
    g = torch.Generator().manual_seed(hw + cin)
    wt = torch.randn(3, 3, cin, cout, generator=g) * (2.0 / (9 * cin)) ** 0.5
    x = _rand(1, hw, hw, cin, seed=3, relu=True)
    u_b = _ops.winograd_weights(wt.flip(0, 1).permute(2, 3, 0, 1).contiguous().to(DEV), 4, DEV)
    gy = _rand(1, hw, hw, cout, seed=4)
    a = _ops.conv3x3_winograd_dgrad(gy, u_b, cin, act_in=x)
    pre = _rand(1, hw, hw, cin, seed=5)
    c = pre.clone()
    _ops.conv3x3_winograd_dgrad(gy, u_b, cin, act_in=x, out=c, accumulate=True)
    assert torch.equal(c, pre + a)
    with pytest.raises(Exception):                       # no mask source: only the masked (tapped-layer) form accumulates
        _ops.conv3x3_winograd_dgrad(gy, u_b, cin, out=c, accumulate=True)
