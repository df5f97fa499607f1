"""Known answers for the image-side oracle: bilinear resize, Laplacian pyramid, sampling,
VGG shapes, schedule arithmetic (SURVEY.md section 8c list)."""
import math

import numpy as np
import torch
import torch.nn.functional as F

from oracle import strotss_oracle as O


def _img(h, w, c=3, seed=0, dtype=torch.float64):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(1, h, w, c, generator=g, dtype=dtype)


def test_resize_matches_independent_bilinear():
    # half-pixel-centre bilinear without antialias == F.interpolate(align_corners=False)
    for (h, w, oh, ow) in [(64, 64, 32, 32), (32, 32, 64, 64), (85, 128, 42, 64), (42, 64, 85, 128),
                           (21, 32, 341, 512), (7, 5, 1, 1), (1, 1, 4, 3), (321, 481, 42, 64)]:
        x = _img(h, w)
        ours = O.resize_bilinear(x, oh, ow)
        ref = F.interpolate(x.permute(0, 3, 1, 2), size=(oh, ow), mode="bilinear",
                            align_corners=False, antialias=False).permute(0, 2, 3, 1)
        assert ours.shape == (1, oh, ow, 3)
        assert (ours - ref).abs().max() < 5e-5      # fp32 coordinate arithmetic (TF) vs fp64 coords


def test_resize_linear_ramp_and_identity():
    # a linear ramp is reproduced exactly in the interior when upsampling x2
    w = 16
    ramp = torch.arange(w, dtype=torch.float64).view(1, 1, w, 1).expand(1, 4, w, 1).contiguous()
    up = O.resize_bilinear(ramp, 8, 2 * w)
    xs = (torch.arange(2 * w, dtype=torch.float64) + 0.5) / 2 - 0.5
    assert torch.allclose(up[0, 0, 1:-1, 0], xs[1:-1], atol=1e-6)
    assert up[0, 0, 0, 0] == 0 and up[0, 0, -1, 0] == w - 1           # edge clamp
    x = _img(9, 13)
    assert torch.equal(O.resize_bilinear(x, 9, 13), x)


def test_pyramid_roundtrip_and_sizes():
    for (h, w) in [(64, 64), (42, 64), (341, 512), (33, 70)]:
        x = _img(h, w, seed=3)
        pyr = O.make_laplacian_pyramid(x)
        assert len(pyr) == 6
        hh, ww = h, w
        for p in pyr[:-1]:
            assert p.shape[1:3] == (hh, ww)
            hh, ww = max(hh // 2, 1), max(ww // 2, 1)
        assert pyr[-1].shape[1:3] == (hh, ww)
        assert (O.fold_laplacian_pyramid(pyr) - x).abs().max() < 1e-12


def test_scale_size_table_content_im():
    # content_im.jpg is 321x481 (HxW): 42x64, 85x128, 170x256, 341x512, 683x1024
    x = torch.zeros(1, 321, 481, 3)
    sizes = [tuple(O.resize(x, s).shape[1:3]) for s in O.scale_schedule(5)]
    assert sizes == [(42, 64), (85, 128), (170, 256), (341, 512), (683, 1024)]
    assert O.scale_schedule(4) == [64, 128, 256, 512]


def test_sampling_step_table():
    # square S: steps (1,1),(1,1),(2,2),(4,4),(8,8); content_im: (1,1),(1,1),(1,2),(3,4),(6,7)
    assert [O.sampling_steps(s, s) for s in (64, 128, 256, 512, 1024)] == [(1, 1), (1, 1), (2, 2), (4, 4), (8, 8)]
    assert [O.sampling_steps(h, w) for h, w in ((42, 64), (85, 128), (170, 256), (341, 512), (683, 1024))] == \
        [(1, 1), (1, 1), (1, 2), (3, 4), (6, 7)]


def test_make_indices_properties():
    rng = np.random.default_rng(0)
    idx = O.make_indices(256, 256, True, 1024, rng)
    assert idx.shape == (1024, 2) and idx.dtype == np.float32
    assert len({(a, b) for a, b in idx.tolist()}) == 1024            # no repeats
    sx, sy = O.sampling_steps(256, 256)
    assert len(set((idx[:, 0] % sx).tolist())) == 1 and len(set((idx[:, 1] % sy).tolist())) == 1
    idx = O.make_indices(20, 30, False, 1024, rng)
    assert idx.shape == (600, 2)
    # mask: only pixels inside the (resized, >0.5) mask survive
    m = np.zeros((64, 64, 1), np.float32); m[:, :32] = 1
    idx = O.make_indices(128, 128, True, 1024, rng, mask=m)
    assert idx[:, 1].max() < 64 and idx.shape[0] == 1024
    # all-zero mask -> treated as all-true (strotss_utils.py:107-108)
    idx = O.make_indices(64, 64, True, 1024, rng, mask=np.zeros((64, 64, 1), np.float32))
    assert idx.shape[0] == 1024


def test_map_divisors_axis_choice():
    # square power-of-two: H axis; 341x512 (H not a power of two at the first shrink): W axis
    sq = [(64, 64)] * 3 + [(32, 32)] * 2 + [(16, 16)] * 3 + [(8, 8), (4, 4)]
    ch = O.map_divisors(sq)
    assert ch[0] == [] and ch[3] == [2.0] and ch[5] == [2.0, 2.0] and ch[9] == [2.0] * 4
    ns = [(341, 512)] * 3 + [(170, 256)] * 2 + [(85, 128)] * 3 + [(42, 64), (21, 32)]
    ch = O.map_divisors(ns)
    assert ch[3] == [512 / 256] and ch[9] == [2.0, 2.0, 2.0, 2.0]
    # portrait 512x341: first shrunk H=256 is a power of two -> H axis: 512/256
    pt = [(512, 341)] * 3 + [(256, 170)] * 2
    assert O.map_divisors(pt)[3] == [2.0]


def test_sample_features_known_patterns():
    x0 = _img(8, 8, 2, seed=1)
    x1 = O.resize_bilinear(x0, 4, 4)
    idx = np.array([[2, 3], [5, 6], [7, 7], [0, 0]], np.float32)
    f = O.sample_features([x0, x1], idx, True)
    assert f.shape == (4, 4)
    # integer coordinates on the full-res map == plain gather
    for k, (r, c) in enumerate(idx.astype(int)):
        assert torch.equal(f[k, :2], x0[0, r, c])
    # on the /2 map: (2,3)/2 = (1, 1.5) -> rows 1, cols 1|2 weights 1/2,1/2
    assert torch.allclose(f[0, 2:], 0.5 * x1[0, 1, 1] + 0.5 * x1[0, 1, 2])
    # (5,6)/2 = (2.5, 3): rows 2|3 at col 3, weight 1/2 each (col+1 clipped to 3, weight 0)
    assert torch.allclose(f[1, 2:], 0.5 * x1[0, 2, 3] + 0.5 * x1[0, 3, 3])
    # (7,7)/2 = (3.5,3.5): x0 = 3, x1 = clip(4) = 3 -> all four taps hit pixel (3,3)
    assert torch.allclose(f[2, 2:], x1[0, 3, 3])
    # nearest: trunc
    fn = O.sample_features([x0, x1], idx, False)
    assert torch.equal(fn[1, 2:], x1[0, 2, 3]) and torch.equal(fn[1, :2], x0[0, 5, 6])


def test_vgg_shapes_and_manual_conv():
    w = O.make_synthetic_vgg16_weights(0)
    vgg = O.VGG(w, dtype=torch.float64)
    x = _img(32, 48, seed=5)
    outs = vgg(x)
    exp = [(32, 48, 64)] * 2 + [(16, 24, 128)] * 2 + [(8, 12, 256)] * 3 + [(4, 6, 512), (2, 3, 512)]
    assert [tuple(o.shape[1:]) for o in outs] == exp
    assert sum(o.shape[-1] for o in outs) + 3 == 2179
    # first layer by hand at one interior and one corner pixel (3x3 SAME zero pad, HWIO kernel)
    xp = vgg.preprocess(x)[0]
    wk, b = vgg.weights[0]
    for (r, c) in [(5, 7), (0, 0), (31, 47)]:
        acc = b.clone()
        for dy in range(3):
            for dx in range(3):
                rr, cc = r + dy - 1, c + dx - 1
                if 0 <= rr < 32 and 0 <= cc < 48:
                    acc = acc + xp[rr, cc] @ wk[dy, dx]
        assert torch.allclose(outs[0][0, r, c], torch.relu(acc), atol=1e-12)


def test_postprocess():
    x = torch.tensor([[[[-0.5, 0.2, 0.4]], [[0.8, 1.5, 0.6]]]], dtype=torch.float64)
    out = O.postprocess(x)
    # clip -> [0,.2,.4,.8,1,.6]; min 0; max 1; *255 truncated
    assert out.dtype == np.uint8 and out.shape == (2, 1, 3)
    assert out.reshape(-1).tolist() == [0, 51, 102, 204, 255, 153]


def test_run_scales_schedule_init_lr_alpha():
    """The oracle's coarse-to-fine driver against run_strotss.py:65-96,154-155 read as text: alpha 16, 8, 4 ...,
    loss_denom = 2 + alpha + 1/max(alpha, 1) (92), the three initialisation branches (82 / 84-85 / 87-88: the last
    scale gets lr/2 and NO Laplacian), fresh RMSprop slots per scale, `--start_level`."""
    w = O.make_synthetic_vgg16_weights(0)
    c = _img(40, 64, seed=1).float(); s = _img(48, 48, seed=2).float()
    tr, steps = [], []
    out = O.run_scales(c, s, w, level=3, max_iter=1, lr=2e-3, sample_size=64, scale_trace=tr, trace=steps)
    assert [t["scl"] for t in tr] == [64, 128, 256] and [t["alpha"] for t in tr] == [16.0, 8.0, 4.0]
    assert [t["lr"] for t in tr] == [2e-3, 2e-3, 1e-3]
    for t in tr:
        assert math.isclose(t["loss_denom"], 2.0 + t["alpha"] + 1.0 / max(t["alpha"], 1.0))
    assert math.isclose(tr[0]["loss_denom"], 18.0625)
    # sizes: long side = scl, int(h / factor)   (utils.py:32-37)
    assert [tuple(t["init"].shape[1:3]) for t in tr] == [(40, 64), (80, 128), (160, 256)]
    c0, s0 = O.resize(c, 64), O.resize(s, 64)
    assert torch.equal(tr[0]["init"], O.make_laplacian(c0) + s0.mean(dim=(1, 2), keepdim=True))
    c1 = O.resize(c, 128)
    assert torch.equal(tr[1]["init"], O.resize_like(tr[0]["final"], c1) + O.make_laplacian(c1))
    c2 = O.resize(c, 256)
    assert torch.equal(tr[2]["init"], O.resize_like(tr[1]["final"], c2))
    assert torch.equal(out, tr[2]["final"]) and len(steps) == 3
    # one RMSprop step from zero slots moves every pixel by <= 10 * lr per level (6 levels)
    assert float((tr[0]["final"] - tr[0]["init"]).abs().max()) <= 6 * 10 * 2e-3 * 1.01
    # a single executed scale keeps the full lr (level = 1 in the reference never reaches the lr/2 branch)
    tr1 = []
    O.run_scales(c, s, w, level=1, max_iter=0, sample_size=64, scale_trace=tr1)
    assert [t["lr"] for t in tr1] == [2e-3] and tr1[0]["alpha"] == 16.0
    # --start_level 1: first executed scale initialised like the reference's first one
    tr2 = []
    O.run_scales(c, s, w, level=3, start_level=1, max_iter=0, sample_size=64, scale_trace=tr2)
    assert [t["scl"] for t in tr2] == [128, 256] and [t["lr"] for t in tr2] == [2e-3, 1e-3]
    assert [t["alpha"] for t in tr2] == [8.0, 4.0]            # alpha as the full schedule would have it at these scales
    s1 = O.resize(s, 128)
    assert torch.equal(tr2[0]["init"], O.make_laplacian(c1) + s1.mean(dim=(1, 2), keepdim=True))


def test_train_step_runs_and_fold_adjoint():
    torch.manual_seed(0)
    w = O.make_synthetic_vgg16_weights(0)
    vgg = O.VGG(w, dtype=torch.float64)
    c = _img(32, 32, seed=1); s = _img(32, 32, seed=2)
    rng = np.random.default_rng(0)
    with torch.no_grad():
        cf = [c] + vgg(c); sf = [s] + vgg(s)
        ss = O.sample_features(sf, O.make_indices(32, 32, False, 256, rng), False)
    variables = [v.clone().requires_grad_(True) for v in O.make_laplacian_pyramid(c)]
    idx = O.make_indices(32, 32, True, 256, rng)
    res = O.train_step(variables, vgg, cf, ss, idx, 16.0, 18.0625)
    assert len(res["grads"]) == 6 and all(g.shape == v.shape for g, v in zip(res["grads"], variables))
    assert torch.isfinite(res["loss"]) and res["loss"] > 0
    # grads[k+1] is the bilinear-transpose of grads[k] (fold is linear): <U g_small', g0> identity
    g0, g1 = res["grads"][0], res["grads"][1]
    probe = _img(16, 16, seed=9)
    lhs = (O.resize_bilinear(probe, 32, 32) * g0).sum()
    rhs = (probe * g1).sum()
    assert abs(float(lhs - rhs)) < 1e-12 * max(1.0, abs(float(lhs)))
