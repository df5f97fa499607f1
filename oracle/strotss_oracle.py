"""CPU oracle for the STROTSS hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module.  The product path (``strotss-tensorflow_amd/``) never does.

PARITY UNPINNED: the reference (interaction-lab-uh/STROTSS-tensorflow) has no tests, no
golden vectors, and every one of its operations is a ``tensorflow`` call; TensorFlow is not
installed in the build container (ordinary missing dependency, nothing was denied) and the
VGG weights are a network fetch.  This file is therefore a *restatement* of the reference's
arithmetic written from its source read as text, with TF-internal semantics (bilinear
resize, l2_normalize, reduce_min / maximum gradients, Keras RMSprop ...) taken from
TensorFlow's documented behaviour.  It is pinned only by analytic known answers,
finite-difference gradient checks and an independent NumPy restatement
(``oracle/numpy_ref.py``) -- see ``tests/test_oracle_*.py``.

Everything is written with torch CPU ops so that the same code yields
  * the float64 oracle (``dtype=torch.float64``) the HIP kernels are compared against, and
  * the float32 single-/multi-thread CPU baseline (``bench.py`` ``cpu_baseline``).
Gradients come from ``torch.autograd`` (the HIP side uses hand-derived backward kernels, so
agreement between the two is an independent check of both).

Each function cites the reference lines it follows (paths relative to /root/reference).
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------------------
# constants of the reference
# --------------------------------------------------------------------------------------
# nn/model.py:7-15 -- tapped layers of Keras VGG16 (post-ReLU conv outputs)
VGG16_CFG: Tuple = (
    ("block1_conv1", 3, 64), ("block1_conv2", 64, 64), "pool",
    ("block2_conv1", 64, 128), ("block2_conv2", 128, 128), "pool",
    ("block3_conv1", 128, 256), ("block3_conv2", 256, 256), ("block3_conv3", 256, 256), "pool",
    ("block4_conv1", 256, 512), ("block4_conv2", 512, 512), ("block4_conv3", 512, 512), "pool",
    ("block5_conv1", 512, 512), ("block5_conv2", 512, 512), ("block5_conv3", 512, 512),
)
VGG16_TAPS = ("block1_conv1", "block1_conv2", "block2_conv1", "block2_conv2", "block3_conv1",
              "block3_conv2", "block3_conv3", "block4_conv3", "block5_conv3")
IMAGENET_MEAN = (0.485, 0.456, 0.406)      # nn/model.py:34
IMAGENET_STD = (0.229, 0.224, 0.225)       # nn/model.py:35
# tf.image.rgb_to_yuv kernel, applied as rgb @ M  (nn/strotss_utils.py:166-167)
RGB2YUV = ((0.299, -0.14714119, 0.61497538),
           (0.587, -0.28886916, -0.51496512),
           (0.114, 0.43601035, -0.10001026))


# --------------------------------------------------------------------------------------
# tf.image.resize(method='bilinear')   (half-pixel centres, antialias=False)
# --------------------------------------------------------------------------------------
def _resize_axis_table(in_size: int, out_size: int):
    """Interpolation table of TF2's bilinear resize along one axis.

    TF computes the source coordinate in float32:
        scale = float(in)/float(out);  src = (float(i)+0.5f)*scale-0.5f
        lower = max(floor(src),0); upper = min(ceil(src), in-1); lerp = src-floor(src)
    """
    scale = np.float32(in_size) / np.float32(out_size)
    i = np.arange(out_size, dtype=np.float32)
    src = (i + np.float32(0.5)) * scale - np.float32(0.5)
    fl = np.floor(src)
    lower = np.maximum(fl, 0).astype(np.int64)
    upper = np.minimum(np.ceil(src), in_size - 1).astype(np.int64)
    lerp = (src - fl).astype(np.float32)
    return lower, upper, lerp


def resize_bilinear(x: torch.Tensor, out_h: int, out_w: int) -> torch.Tensor:
    """tf.image.resize(x, (out_h, out_w)) for x of shape (1,H,W,C) or (H,W,C)."""
    squeeze = x.dim() == 3
    if squeeze:
        x = x[None]
    _, h, w, _ = x.shape
    y0, y1, ly = _resize_axis_table(h, out_h)
    x0, x1, lx = _resize_axis_table(w, out_w)
    ly_t = torch.from_numpy(ly).to(x.dtype).view(1, -1, 1, 1)
    lx_t = torch.from_numpy(lx).to(x.dtype).view(1, 1, -1, 1)
    y0 = torch.from_numpy(y0); y1 = torch.from_numpy(y1)
    x0 = torch.from_numpy(x0); x1 = torch.from_numpy(x1)
    top = x[:, y0]
    bot = x[:, y1]
    tl, tr = top[:, :, x0], top[:, :, x1]
    bl, br = bot[:, :, x0], bot[:, :, x1]
    t = tl + (tr - tl) * lx_t
    b = bl + (br - bl) * lx_t
    out = t + (b - t) * ly_t
    return out[0] if squeeze else out


def resize(image: torch.Tensor, max_size: Optional[int]) -> torch.Tensor:
    """nn/utils.py:32-37 -- resize by long side; sizes via Python doubles + int()."""
    if max_size is None:
        return image
    if image.dim() == 3:
        h, w, _ = image.shape
    elif image.dim() == 4:
        _, h, w, _ = image.shape
    else:
        raise ValueError(f"Invalid rank: {image.dim()}")
    factor = max(h / max_size, w / max_size)
    return resize_bilinear(image, int(h / factor), int(w / factor))


def resize_like(image: torch.Tensor, base: torch.Tensor) -> torch.Tensor:
    """nn/utils.py:40-41"""
    h, w = (base.shape[0], base.shape[1]) if base.dim() == 3 else (base.shape[1], base.shape[2])
    return resize_bilinear(image, h, w)


# --------------------------------------------------------------------------------------
# Laplacian pyramid      nn/strotss_utils.py:139-163
# --------------------------------------------------------------------------------------
def make_laplacian(x: torch.Tensor, return_downscale: bool = False):
    h, w = x.shape[1], x.shape[2]
    hd, wd = max(h // 2, 1), max(w // 2, 1)
    temp = resize_bilinear(x, hd, wd)
    pyr = x - resize_bilinear(temp, h, w)
    if return_downscale:
        return pyr, temp
    return pyr


def make_laplacian_pyramid(x: torch.Tensor, levels: int = 5) -> List[torch.Tensor]:
    xs = []
    cur = x
    for _ in range(levels):
        pyr, cur = make_laplacian(cur, return_downscale=True)
        xs.append(pyr)
    xs.append(cur)
    return xs


def fold_laplacian_pyramid(xs: Sequence[torch.Tensor]) -> torch.Tensor:
    ret = xs[-1]
    for x in reversed(xs[:-1]):
        ret = x + resize_bilinear(ret, x.shape[1], x.shape[2])
    return ret


# --------------------------------------------------------------------------------------
# VGG16 feature extractor     nn/model.py:17-55
# --------------------------------------------------------------------------------------
def make_synthetic_vgg16_weights(seed: int = 0, dtype=torch.float32):
    """Seeded He-normal weights (HWIO) + small seeded biases; there is no network for the
    real ones (nn/model.py:31-33 fetches them).  Returns [(w(3,3,Cin,Cout), b(Cout)), ...]
    in layer order.  Biases are non-zero so the bias path is exercised."""
    g = torch.Generator().manual_seed(seed)
    out = []
    for item in VGG16_CFG:
        if item == "pool":
            continue
        _, cin, cout = item
        std = math.sqrt(2.0 / (9 * cin))
        w = torch.randn(3, 3, cin, cout, generator=g, dtype=torch.float32) * std
        b = torch.randn(cout, generator=g, dtype=torch.float32) * 0.05
        out.append((w.to(dtype), b.to(dtype)))
    return out


class VGG:
    """[9 post-ReLU taps] = VGG(weights)(image in [0,1], NHWC)."""

    def __init__(self, weights, taps: Sequence[str] = VGG16_TAPS, dtype=torch.float64):
        self.dtype = dtype
        self.weights = [(w.to(dtype), b.to(dtype)) for w, b in weights]
        self.taps = tuple(taps)
        self.mean = torch.tensor(IMAGENET_MEAN, dtype=dtype).view(1, 1, 1, 3)
        self.std = torch.tensor(IMAGENET_STD, dtype=dtype).view(1, 1, 1, 3)
        # OIHW copies for F.conv2d
        self._oihw = [(w.permute(3, 2, 0, 1).contiguous(), b) for w, b in self.weights]

    def preprocess(self, x: torch.Tensor) -> torch.Tensor:
        return (x - self.mean) / self.std            # nn/model.py:50-51

    def __call__(self, x: torch.Tensor) -> List[torch.Tensor]:
        h = self.preprocess(x.to(self.dtype)).permute(0, 3, 1, 2)   # NCHW internally
        outs = []
        li = 0
        for item in VGG16_CFG:
            if item == "pool":
                # Keras MaxPooling2D((2,2), strides=(2,2)), padding='valid' -> floor
                h = F.max_pool2d(h, 2, 2)
                continue
            name = item[0]
            w, b = self._oihw[li]
            li += 1
            h = F.relu(F.conv2d(h, w, b, padding=1))  # Conv2D 3x3 'same' + bias + relu
            if name in self.taps:
                outs.append(h.permute(0, 2, 3, 1))
        return outs


# --------------------------------------------------------------------------------------
# Sampling      nn/strotss_utils.py:20-136
# --------------------------------------------------------------------------------------
def sampling_steps(h: int, w: int) -> Tuple[int, int]:
    """nn/strotss_utils.py:89-90"""
    area = math.sqrt((h * w) // (128 ** 2))
    return max(1, math.floor(area)), max(1, math.ceil(area))


def make_indices(h: int, w: int, bilinear_sampling: bool, sample_size: int,
                 rng: np.random.Generator, mask: Optional[np.ndarray] = None) -> np.ndarray:
    """nn/strotss_utils.py:83-121.  The TF Philox stream cannot be reproduced without TF;
    the draw is restated with a NumPy Generator: 'identical seeds' == identical injected
    index sequences.  Returns float32 (n,2) of (row, col)."""
    if bilinear_sampling:
        step_x, step_y = sampling_steps(h, w)
        off_x = int(rng.integers(0, step_x))
        off_y = int(rng.integers(0, step_y))
        X = np.arange(h)[off_x::step_x]
        Y = np.arange(w)[off_y::step_y]
    else:
        X = np.arange(h)
        Y = np.arange(w)
    # tf.meshgrid(X, Y) default indexing='xy' -> shapes (len(Y), len(X)); then flattened
    XX, YY = np.meshgrid(X, Y)
    ret = np.stack([XX.reshape(-1), YY.reshape(-1)], axis=1)
    if mask is not None:
        m = np.asarray(mask, dtype=np.float32)
        if m.ndim == 2:
            m = m[..., None]
        m = resize_bilinear(torch.from_numpy(m), h, w).numpy()[..., 0]
        if m.max() < 0.1:
            keep = (m + 1) > 0.5
        else:
            keep = m > 0.5
        ret = ret[keep[ret[:, 0], ret[:, 1]]]
    perm = rng.permutation(ret.shape[0])     # tf.random.shuffle of the PAIRS (115-119)
    ret = ret[perm][:sample_size]
    return ret.astype(np.float32)


def map_divisors(shapes: Sequence[Tuple[int, int]]) -> List[List[float]]:
    """For each map in the list, the chain of float divisors applied so far to the indices.
    nn/strotss_utils.py:31-37: `indices /= y` is cumulative, the axis is chosen ONCE."""
    chains, cur, index = [], [], None
    for i, (h, w) in enumerate(shapes):
        if i > 0 and h < shapes[i - 1][0]:
            if index is None:
                index = 0 if not (math.log2(h) % 1) else 1     # 0 -> H axis, 1 -> W axis
            y = shapes[i - 1][index] / shapes[i][index]
            cur = cur + [y]
        chains.append(list(cur))
    return chains


def sample_features(xs: Sequence[torch.Tensor], indices: np.ndarray,
                    bilinear_sampling: bool) -> torch.Tensor:
    """nn/strotss_utils.py:25-81.  xs: list of (1,h,w,c) maps; indices (n,2) float32.
    Index arithmetic is done in float32 exactly as the reference's float32 tensors do."""
    shapes = [(int(x.shape[1]), int(x.shape[2])) for x in xs]
    chains = map_divisors(shapes)
    feats = []
    for x, chain in zip(xs, chains):
        idx = indices.astype(np.float32).copy()
        for y in chain:
            idx = (idx / np.float32(y)).astype(np.float32)
        cur = x[0]
        h, w, c = cur.shape
        gx, gy = idx[:, 0], idx[:, 1]
        flat = cur.reshape(h * w, c)
        if bilinear_sampling:
            gxf = np.floor(gx); dx = gx - gxf
            gyf = np.floor(gy); dy = gy - gyf
            wa = (1 - dx) * (1 - dy); wb = (1 - dx) * dy
            wc = dx * (1 - dy); wd = dx * dy
            x0 = np.clip(gxf, 0, h - 1).astype(np.int64)
            y0 = np.clip(gyf, 0, w - 1).astype(np.int64)
            x1 = np.clip(x0 + 1, 0, h - 1)
            y1 = np.clip(y0 + 1, 0, w - 1)
            tw = lambda a: torch.from_numpy(a.astype(np.float32)).to(x.dtype).view(-1, 1)
            ti = lambda a: torch.from_numpy(a)
            g = (flat[ti(x0 * w + y0)] * tw(wa) + flat[ti(x0 * w + y1)] * tw(wb)
                 + flat[ti(x1 * w + y0)] * tw(wc) + flat[ti(x1 * w + y1)] * tw(wd))
        else:
            xi = np.clip(gx, 0, h - 1).astype(np.int32).astype(np.int64)   # trunc cast
            yi = np.clip(gy, 0, w - 1).astype(np.int32).astype(np.int64)
            g = flat[torch.from_numpy(xi * w + yi)]
        feats.append(g)
    return torch.cat(feats, dim=1)


# --------------------------------------------------------------------------------------
# losses      nn/losses.py:4-80
# --------------------------------------------------------------------------------------
def l2_normalize_rows(x: torch.Tensor) -> torch.Tensor:
    """tf.nn.l2_normalize(x, axis=1): x * rsqrt(max(sum(x^2), 1e-12))"""
    ss = (x * x).sum(dim=1, keepdim=True)
    return x * torch.rsqrt(torch.clamp(ss, min=1e-12))


def cosine_distance(x, y):                      # losses.py:12-15
    return 1 - l2_normalize_rows(x) @ l2_normalize_rows(y).T


def l2_distance(x, y):                          # losses.py:18-24
    x_sq = (x ** 2).sum(dim=1).view(-1, 1)
    y_sq = (y ** 2).sum(dim=1).view(1, -1)
    m = x_sq + y_sq - 2.0 * (x @ y.T)
    m = torch.clamp(m, min=1e-6) / x.shape[1]
    return torch.sqrt(m)


dist_metrics = {"cosine": cosine_distance, "l2": l2_distance,
                "both": lambda x, y: cosine_distance(x, y) + l2_distance(x, y)}


def moment_matching(x, y):                      # losses.py:39-52
    xm = x.mean(dim=0, keepdim=True)
    ym = y.mean(dim=0, keepdim=True)
    cx, cy = x - xm, y - ym
    xv = cx.T @ cx / x.shape[0]
    yv = cy.T @ cy / y.shape[0]
    return (xv - yv).abs().mean() + (xm - ym).abs().mean()


def self_similarity(x, y):                      # losses.py:55-66
    xd = cosine_distance(x, x)
    xd = xd / torch.clamp(xd.sum(dim=0), min=1e-12)
    yd = cosine_distance(y, y)
    yd = yd / torch.clamp(yd.sum(dim=0), min=1e-12)
    return (xd - yd).abs().mean() * y.shape[0]


def relaxed_emd(x, y, distance: str = "cosine"):   # losses.py:69-80
    C = dist_metrics[distance](x, y)
    # tf.reduce_min's gradient is split equally among ties -> torch.amin (same rule)
    r_x = torch.amin(C, dim=1).mean()
    r_y = torch.amin(C, dim=0).mean()
    # tf.maximum sends the gradient to the FIRST argument on ties (x >= y)
    return torch.where(r_x >= r_y, r_x, r_y)


def sinkhorn_knopp(x, y, distance: str = "cosine", l: float = 10.0, N_iter: int = 30):
    """BUILD-DEFINED (no reference behaviour exists: losses.py:83-105 is marked untested, never called, and raises on
    `tf.ones_like(shape)` of a tuple).  Its evident intent, with the second marginal 1/len(y):
    K = exp(-l M); u = p / max(K v, 1e-12); v = q / max(K^T u, 1e-12), N_iter times from v = 1; sum(u * ((K*M) v)).
    Differentiated through the iterations by autograd, as TF would."""
    M = dist_metrics[distance](x, y)
    K = torch.exp(-l * M)
    p, q = 1.0 / x.shape[0], 1.0 / y.shape[0]
    v = torch.ones(y.shape[0], 1, dtype=M.dtype)
    u = torch.ones(x.shape[0], 1, dtype=M.dtype)
    for _ in range(N_iter):
        u = p / torch.clamp(K @ v, min=1e-12)
        v = q / torch.clamp(K.t() @ u, min=1e-12)
    return (u * ((K * M) @ v)).sum()


def convert_rgb_to_yuv(x):                      # strotss_utils.py:166-167
    m = torch.tensor(RGB2YUV, dtype=x.dtype)
    return x[:, :3] @ m


def content_loss(target, prediction):           # run_strotss.py:21-24
    return self_similarity(prediction, target)


def style_loss(target, prediction, alpha: float):   # run_strotss.py:27-40
    inv_alpha = 1 / max(alpha, 1)
    l_m = moment_matching(target, prediction)
    l_remd = relaxed_emd(target, prediction)
    l_pal = relaxed_emd(convert_rgb_to_yuv(target), convert_rgb_to_yuv(prediction), "both")
    return l_m + l_remd + inv_alpha * l_pal


# --------------------------------------------------------------------------------------
# optimiser / postprocess
# --------------------------------------------------------------------------------------
def rmsprop_update(var, rms, grad, lr: float, rho: float = 0.99, eps: float = 1e-8):
    """Keras OptimizerV2 RMSprop dense path, momentum 0, not centred (run_strotss.py:63,148):
       rms <- rho*rms + (1-rho)*g^2 ;  var <- var - lr*g/(sqrt(rms)+eps).   In place."""
    rms.mul_(rho).add_((1 - rho) * grad * grad)
    var.sub_(lr * grad / (rms.sqrt() + eps))


def postprocess(final: torch.Tensor) -> np.ndarray:     # strotss_utils.py:170-175
    f = final.to(torch.float32).clamp(0, 1)
    f = f - f.min()
    f = f / f.max()
    return (f * 255).to(torch.uint8)[0].numpy()          # truncating cast


# --------------------------------------------------------------------------------------
# the hot path: one optimisation step    run_strotss.py:131-148
# --------------------------------------------------------------------------------------
def train_step(variables: List[torch.Tensor], vgg: VGG, content_feat, style_samples,
               indices: np.ndarray, alpha: float, loss_denom: float):
    """Forward + backward of run_strotss.py:131-142 (no masks).  `variables` are the 6
    pyramid tensors (leaf, requires_grad).  Returns dict(loss, loss_c, loss_s, grads)."""
    for v in variables:
        v.grad = None
    img = fold_laplacian_pyramid(variables)
    pred = [img] + vgg(img)
    c_feat = sample_features(content_feat, indices, True)
    p_feat = sample_features(pred, indices, True)
    loss_c = content_loss(c_feat, p_feat)
    loss_s = style_loss(style_samples, p_feat, alpha)
    loss = (alpha * loss_c + loss_s) / loss_denom
    grads = torch.autograd.grad(loss, variables)
    return {"loss": loss.detach(), "loss_c": loss_c.detach(), "loss_s": loss_s.detach(),
            "grads": list(grads), "p_feat": p_feat.detach(), "img": img.detach()}


def train_step_masked(variables, vgg: VGG, content_feat, style_samples_per_region,
                      indices_per_region, alpha: float, loss_denom: float):
    """run_strotss.py:104-125: one VGG forward, per-region samples/losses, mean over regions."""
    img = fold_laplacian_pyramid(variables)
    pred = [img] + vgg(img)
    loss = 0.0
    lc_a = 0.0
    ls_a = 0.0
    r = len(indices_per_region)
    for idx, s_samp in zip(indices_per_region, style_samples_per_region):
        c_feat = sample_features(content_feat, idx, True)
        p_feat = sample_features(pred, idx, True)
        lc = content_loss(c_feat, p_feat)
        ls = style_loss(s_samp, p_feat, alpha)
        loss = loss + (alpha * lc + ls) / loss_denom
        lc_a = lc_a + lc
        ls_a = ls_a + ls
    loss = loss / r
    grads = torch.autograd.grad(loss, variables)
    return {"loss": loss.detach(), "loss_c": (lc_a / r).detach(), "loss_s": (ls_a / r).detach(),
            "grads": list(grads), "img": img.detach()}


def scale_schedule(level: int, start_level: int = 0):
    """run_strotss.py:70-71: scl = 2 << (5+i)."""
    return [2 << (5 + i) for i in range(start_level, level)]


def run_scales(content: torch.Tensor, style: torch.Tensor, weights, *, level: int = 4,
               start_level: int = 0, max_iter: int = 200, lr: float = 2e-3,
               alpha: float = 1.0, seed: int = 0, sample_size: int = 1024,
               dtype=torch.float32, index_stream=None, trace=None, scale_trace=None, previous_override=None, rng=None):
    """The coarse-to-fine driver of run_strotss.py:43-161 (no masks) on CPU tensors.
    `index_stream(scale_i, it, h, w)` may inject the (n,2) indices; by default they come
    from make_indices with `rng` (default np.random.default_rng(seed)).  Returns the float stylised image.
    `scale_trace` (a list) receives per executed scale dict(i, scl, lr, alpha, loss_denom, init, final);
    `previous_override(i)` may return the image to take as the previous scale's result at scale i (tests
    re-synchronise free-running trajectories with it), or None to keep the oracle's own."""
    rng = np.random.default_rng(seed) if rng is None else rng     # (any object with integers / permutation: make_indices)
    vgg = VGG(weights, dtype=dtype)
    content = content.to(dtype); style = style.to(dtype)
    a = alpha * 16.0 / 2.0 ** start_level      # alpha is halved after every scale of the schedule, skipped ones included
    stylized = None
    executed = list(range(start_level, level))
    for n_exec, i in enumerate(executed):
        scl = 2 << (5 + i)
        c = resize(content, scl); s = resize(style, scl)
        lap = make_laplacian(c)
        cur_lr = lr
        if previous_override is not None and n_exec > 0:
            over = previous_override(i)
            if over is not None:
                stylized = over.to(dtype)
        if n_exec == 0:
            stylized = lap + s.mean(dim=(1, 2), keepdim=True)
        elif i < level - 1:
            stylized = resize_like(stylized, c) + lap
        else:
            stylized = resize_like(stylized, c)
            cur_lr = lr / 2
        variables = [v.clone().requires_grad_(True) for v in make_laplacian_pyramid(stylized)]
        rms = [torch.zeros_like(v) for v in variables]
        denom = 2.0 + a + 1.0 / max(a, 1.0)
        if scale_trace is not None:
            scale_trace.append(dict(i=i, scl=scl, lr=cur_lr, alpha=a, loss_denom=denom, init=stylized.clone()))
        with torch.no_grad():
            c_feat = [c] + vgg(c)
            s_feat = [s] + vgg(s)
            s_idx = (index_stream(i, -1, s.shape[1], s.shape[2]) if index_stream else
                     make_indices(s.shape[1], s.shape[2], False, sample_size, rng))
            s_samp = sample_features(s_feat, s_idx, False)
        for it in range(max_iter):
            idx = (index_stream(i, it, c.shape[1], c.shape[2]) if index_stream else
                   make_indices(c.shape[1], c.shape[2], True, sample_size, rng))
            res = train_step(variables, vgg, c_feat, s_samp, idx, a, denom)
            with torch.no_grad():
                for v, r, g in zip(variables, rms, res["grads"]):
                    rmsprop_update(v, r, g, cur_lr)
            if trace is not None:
                trace.append((i, it, float(res["loss"]), float(res["loss_c"]), float(res["loss_s"])))
        with torch.no_grad():
            stylized = fold_laplacian_pyramid([v.detach() for v in variables])
        if scale_trace is not None:
            scale_trace[-1]["final"] = stylized.clone()
        a /= 2.0
    return stylized
