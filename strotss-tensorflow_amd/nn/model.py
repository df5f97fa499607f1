"""VGG feature extractor on the HIP conv kernels -- mirrors the reference's nn/model.py:17-55.

`VGG(layers=None, vgg_type='16', use_keras_weight=False, name=None)` keeps the reference
signature and returns the same 9 post-ReLU taps from `__call__(inputs)` for an NHWC [0,1] RGB
image.  Differences forced by the environment (documented in DESIGN.md):
  * the reference downloads `vgg16_norm.h5` at construction (model.py:31-33); there is no network
    here, so weights come from `weights=` (an .npz with HWIO kernels, see `load_weights`) or are
    seeded He-normal synthetic ones (`seed=`);
  * tensors are torch HIP tensors.
The trunk is frozen (model.py:45): only the data gradient exists.
"""
from __future__ import annotations

import math
import os
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _ops

_STROTSS_DEFAULTS = ['block1_conv1', 'block1_conv2', 'block2_conv1', 'block2_conv2', 'block3_conv1',
                     'block3_conv2', 'block3_conv3', 'block4_conv3', 'block5_conv3']

_BLOCKS = {'16': (2, 2, 3, 3, 3), '19': (2, 2, 4, 4, 4)}
_WIDTHS = (64, 128, 256, 512, 512)


def vgg_config(vgg_type: str = '16') -> List:
    """[('block1_conv1', 3, 64), ..., 'pool', ...] in Keras layer order (include_top=False)."""
    cfg, cin = [], 3
    for b, (reps, width) in enumerate(zip(_BLOCKS[str(vgg_type)], _WIDTHS), start=1):
        for r in range(1, reps + 1):
            cfg.append((f'block{b}_conv{r}', cin, width))
            cin = width
        cfg.append('pool')
    return cfg


def synthetic_weights(vgg_type: str = '16', seed: int = 0) -> List[Tuple[torch.Tensor, torch.Tensor]]:
    """Seeded He-normal HWIO kernels + small biases (same generator order as the oracle's
    make_synthetic_vgg16_weights so both sides see identical numbers)."""
    g = torch.Generator().manual_seed(seed)
    out = []
    for item in vgg_config(vgg_type):
        if item == 'pool':
            continue
        _, cin, cout = item
        # in place: the same numbers as `randn(...) * scale` (one f32 product each) without a second 59-MB pass of
        # allocations -- this is a third of the "model build" the wall-clock metric starts with
        w = torch.randn(3, 3, cin, cout, generator=g, dtype=torch.float32).mul_(math.sqrt(2.0 / (9 * cin)))
        b = torch.randn(cout, generator=g, dtype=torch.float32).mul_(0.05)
        out.append((w, b))
    return out


def load_weights(path: str, vgg_type: str = '16') -> List[Tuple[torch.Tensor, torch.Tensor]]:
    """.npz with arrays `<layer>/kernel` (3,3,Cin,Cout HWIO, the Keras layout) and `<layer>/bias`,
    or a torchvision-style state dict saved as .npz: `features.<i>.weight` (Cout,Cin,3,3) / `features.<i>.bias` in
    layer order (the batch-norm-free vgg16 / vgg19; `classifier.*` and any other keys are ignored).
    Returns [(HWIO kernel, bias)] in Keras layer order (the reference fetches vgg16_norm.h5 instead, model.py:31-33)."""
    z = np.load(path)
    cfg = [it for it in vgg_config(vgg_type) if it != 'pool']
    names = [it[0] for it in cfg]
    out = []
    if f'{names[0]}/kernel' in z:
        for n in names:
            out.append((torch.from_numpy(z[f'{n}/kernel']).float(), torch.from_numpy(z[f'{n}/bias']).float()))
    else:
        keys = sorted((k for k in z.files if k.startswith('features.') and k.endswith('.weight') and z[k].ndim == 4),
                      key=lambda k: int(k.split('.')[1]))
        if len(keys) != len(names):
            raise ValueError(f"{path}: {len(keys)} convolution kernels `features.<i>.weight`, VGG{vgg_type} has {len(names)}")
        for k in keys:
            w = torch.from_numpy(z[k]).float().permute(2, 3, 1, 0).contiguous()   # OIHW -> HWIO
            out.append((w, torch.from_numpy(z[k[:-len('.weight')] + '.bias']).float()))
    for (name, cin, cout), (w, b) in zip(cfg, out):
        if tuple(w.shape) != (3, 3, cin, cout) or tuple(b.shape) != (cout,):
            raise ValueError(f"{path}: {name} has kernel {tuple(w.shape)} / bias {tuple(b.shape)}, expected "
                             f"(3, 3, {cin}, {cout}) / ({cout},)")
    return out


def use_winograd(cin: int, cout: int) -> bool:
    """Layers that MAY run in Winograd form (their transformed weights are prepared): every generic layer.
    STROTSS_WINOGRAD=0 disables Winograd altogether (direct implicit GEMM everywhere)."""
    if os.environ.get("STROTSS_WINOGRAD", "auto") == "0":
        return False
    return cin % 32 == 0 and cout % 64 == 0


def direct_splitk(h: int, w: int, cin: int, cout: int) -> bool:
    """Small maps: the direct kernel with K split over up to 256 workgroups + a fixed-order finish kernel (csrc/conv.hip)
    instead of the three dependent, latency-bound launches of the Winograd form.  Measured by scale (bench.py --scale,
    ms/step with / without): 64 px 0.87 / 0.97, 128 px 0.96 / 1.07, 256 px 1.27 / 1.28; beyond these borders the
    direct form's 4x MACs cost more than the launches save (256 px with every layer up to 256 tiles: 1.40).
    A layer takes it with at most STROTSS_DIRECT_MAX_TILES (64) output tiles of 64 x 64, or at most 4x that many when it
    has at most 128 channels on both sides (short K); 0 switches it off."""
    tiles = -(-(h * w) // 64) * (cout // 64)
    mt = int(os.environ.get("STROTSS_DIRECT_MAX_TILES", "64"))
    return 0 < tiles <= mt or (max(cin, cout) <= 128 and 0 < tiles <= 4 * mt)


def winograd_tile(h: int, w: int, cin: int = 128, cout: int = 0) -> int:
    """0 = direct kernel, 2 = F(2x2,3x3), 4 = F(4x4,3x3) for a layer with `cin` input channels at (h, w).
    Measured on MI355X (tools/conv_bench.py): F(4x4,3x3) -- 2.25 MACs per output, transform tensors 2.25x the
    activations -- wins from 32x32 pixels up for every channel count (1.2x at 64->64 ... 2.9x at 512->512 over
    the direct kernel); below that F(2x2,3x3) (4 MACs, 4x tensors) wins for Cin >= 128 and the direct kernel
    for Cin = 64.  f32 rounding of the convolution: direct ~2e-7, F(2x2,3x3) ~5e-7, F(4x4,3x3) ~1e-5 of the
    output range; every step-level parity test holds at unchanged tolerances.
    STROTSS_WINOGRAD_TILE=2 | 4 forces one tiling for the Cin >= 128 layers."""
    mode = os.environ.get("STROTSS_WINOGRAD_TILE", "auto")
    if mode in ("2", "4"):
        return int(mode) if cin >= 128 else 0
    if cout and direct_splitk(h, w, cin, cout):
        return 0
    if h * w >= 1024:
        return 4
    return 2 if cin >= 128 else 0


ROUTE_NAMES = {0: "F2_gemm_f32", 1: "F4_fused_f32", 2: "F4_gemm_f32", 3: "F4_x3_gemm_128", 4: "F4_x3_gemm_64"}


def conv_route(h: int, w: int, cin: int, cout: int, dgrad: bool = False) -> str:
    """Which kernels the generic 3x3 layer `cin` -> `cout` runs at (h, w), forward or data-gradient: the host policy above
    (`winograd_tile`, one decision per layer for both directions, as VGGTrunk takes it) followed by the library's
    (`strotss_conv3x3_winograd_route` / `strotss_conv3x3_workspace_bytes`, include/strotss_hip.h; the data-gradient is the
    same kernel with the channel roles swapped).  No GPU needed: tests/test_route_table.py pins the default table of the
    five BASELINE scales so that a policy regression cannot pass unnoticed."""
    from . import _hip
    t = winograd_tile(h, w, cin, cout) if use_winograd(cin, cout) else 0
    ci, co = (cout, cin) if dgrad else (cin, cout)
    if t == 0:
        return "direct_splitk" if _ops.conv3x3_direct_splits(h, w, ci, co) else "direct"
    p = 16 if t == 2 else 36
    r = int(_hip.load_library().strotss_conv3x3_winograd_route(h, w, ci, co, t, int(_ops.winograd_packed_wanted(p, co, ci)),
                                                               int(_ops.winograd_x3_wanted(p, co, ci, h, w))))
    return ROUTE_NAMES[r]


class _LazyWinograd:
    """u[m] = Winograd-domain weights (P, rows, k) of one layer and direction for tile size m, made on first use."""

    def __init__(self, w_hwio: torch.Tensor, forward: bool, device):
        self.w, self.forward, self.u, self.device = w_hwio, forward, {}, torch.device(device)

    def __getitem__(self, m: int) -> torch.Tensor:
        if m not in self.u:
            g = self.w.permute(3, 2, 0, 1) if self.forward else self.w.flip(0, 1).permute(2, 3, 0, 1)
            self.u[m] = _ops.winograd_weights(g, m, self.device)        # on the MODEL's device, whichever is current
            assert not torch.cuda.is_available() or self.u[m].device == _ops.canonical_device(self.device)
        return self.u[m]


class VGGParams:
    """Frozen weights on the device in the layouts the kernels want (built once)."""

    def __init__(self, weights: Sequence[Tuple[torch.Tensor, torch.Tensor]], vgg_type: str = '16',
                 layers: Optional[Sequence[str]] = None, device="cuda", use_keras_weight: bool = False):
        self.cfg = vgg_config(vgg_type)
        self.tap_names = list(layers or _STROTSS_DEFAULTS)
        names = [it[0] for it in self.cfg if it != 'pool']
        for t in self.tap_names:
            assert t in names, f"unknown layer {t}"
        # drop everything after the last tapped layer: it can never influence a tap
        last = max(names.index(t) for t in self.tap_names)
        cut, seen = [], 0
        for it in self.cfg:
            if it != 'pool':
                if seen > last:
                    break
                seen += 1
            cut.append(it)
        while cut and cut[-1] == 'pool':
            cut.pop()
        self.cfg = cut
        if use_keras_weight:
            # keras.applications.vgg16.preprocess_input(x*255): RGB->BGR, minus the BGR means, no
            # std (model.py:38).  Equivalent: (x - m/255) / (1/255) with the kernel's ci flipped.
            self.mean = tuple(v / 255.0 for v in (123.68, 116.779, 103.939))
            self.std = (1 / 255.0,) * 3
        else:
            self.mean, self.std = _ops.IMAGENET_MEAN, _ops.IMAGENET_STD
        self.layers = []
        li = 0
        for it in self.cfg:
            if it == 'pool':
                continue
            name, cin, cout = it
            w, b = weights[li]
            li += 1
            assert tuple(w.shape) == (3, 3, cin, cout), (name, tuple(w.shape))
            w = w.float().to(device)          # re-layouts below run on the device (once per model)
            if use_keras_weight and cin == 3:
                w = w.flip(2)
            L = {"name": name, "cin": cin, "cout": cout, "bias": b.float().contiguous().to(device)}
            if cin == 3:
                L["w_fwd"] = w.reshape(27, cout).contiguous().to(device)                  # (27, cout)
                L["w_bwd"] = w.flip(0, 1).reshape(9, 3, cout).contiguous().to(device)      # (9, 3, cout)
            else:
                L["w_fwd"] = w.permute(0, 1, 3, 2).reshape(9, cout, cin).contiguous().to(device)   # (9,cout,cin)
                L["w_bwd"] = w.flip(0, 1).reshape(9, cin, cout).contiguous().to(device)           # (9,cin,cout)
                if use_winograd(cin, cout):
                    # forward: g[co][ci][r][q] = W[r,q,ci,co]; dgrad: g'[ci][co][r][q] = W[2-r,2-q,ci,co].
                    # The float64 transform G g G^T runs on the device, and only for the (direction, tiling) pairs a
                    # trunk actually asks for: all four for all 12 layers are 104x the weights = 6 GB and a third of
                    # the model's build time, and F(2x2,3x3) is hardly ever chosen since the small maps take the
                    # direct split-K kernel.
                    L["u_fwd"] = _LazyWinograd(w, True, device)
                    L["u_bwd"] = _LazyWinograd(w, False, device)
            self.layers.append(L)
        self.device = device

    @property
    def tap_layer_indices(self) -> List[int]:
        names = [L["name"] for L in self.layers]
        return [names.index(t) for t in self.tap_names]


class VGGTrunk:
    """Pre-allocated forward/backward of the trunk for one image size.

    forward(img) fills `acts` and returns the tapped maps (views, no copies).
    backward(scatter) walks the layers in reverse: `scatter(layer_index)` is called right after
    the gradient buffer of a tapped layer is complete from above, and must ADD the hypercolumn
    gradient of that tap into `grads[layer_index]` (ReLU-masked); returns the pixel gradient buffer
    into which `scatter(-1)` has added the image-channel part."""

    def __init__(self, params: VGGParams, h: int, w: int, with_grad: bool = True, halo=None):
        """halo: a parallel.HaloExchange when (h, w) is the window of a halo-exchange strip: after every layer, forward
        and backward, the window's outermost rows are refreshed from the neighbouring ranks."""
        self.p = params
        self.halo = halo
        dev = params.device
        self.h, self.w = h, w
        self.acts: List[torch.Tensor] = []
        self.pools: List[torch.Tensor] = []
        self.plan = []     # ('conv', layer_idx, src) | ('pool', pool_idx, src_layer)
        ch, cw = h, w
        li = pi = 0
        src = ('img', 0)
        for it in params.cfg:
            if it == 'pool':
                self.pools.append(torch.empty((1, ch // 2, cw // 2, self.acts[-1].shape[-1]), dtype=torch.float32, device=dev))
                self.plan.append(('pool', pi, li - 1))
                src = ('pool', pi)
                pi += 1
                ch, cw = ch // 2, cw // 2
                if ch < 1 or cw < 1:
                    raise ValueError(f"image {h}x{w} too small for this VGG depth")
            else:
                self.acts.append(torch.empty((1, ch, cw, it[2]), dtype=torch.float32, device=dev))
                self.plan.append(('conv', li, src))
                src = ('conv', li)
                li += 1
        # pooling level (number of 2x2 pools above) of every layer's output and of every pooled map
        self.layer_level, self.pool_level, lvl = {}, {}, 0
        for step in self.plan:
            if step[0] == 'pool':
                lvl += 1
                self.pool_level[step[1]] = lvl
            else:
                self.layer_level[step[1]] = lvl
        self.taps = params.tap_layer_indices
        self.with_grad = with_grad
        # Winograd tile per layer for this image size
        self.wtile = [winograd_tile(int(a.shape[1]), int(a.shape[2]), L["cin"], L["cout"]) if "u_fwd" in L else 0
                      for L, a in zip(params.layers, self.acts)]
        self.relu_bits = [None] * len(self.acts)
        if with_grad:
            # argmax codes of the pools (1 byte per pooled element): the backward pass reads them, not the activations
            self.pool_codes = [torch.empty(p.shape, dtype=torch.uint8, device=dev) for p in self.pools]
            # sign words of the activations whose ReLU mask an F(4x4,3x3) data-gradient applies: 4 bytes per 4x4 tile and
            # channel, written by the producing forward kernel from registers, instead of re-reading the f32 activation
            # (268 MB for block1_conv1 at 1024^2).  Not with a halo exchange: it rewrites border rows of the activations.
            if halo is None and os.environ.get("STROTSS_RELU_BITS", "1") != "0":
                for step in self.plan:
                    if step[0] == 'conv' and step[2][0] == 'conv' and self.wtile[step[1]] == 4:
                        si = step[2][1]
                        if self.wtile[si] == 4 or params.layers[si]["cin"] == 3:
                            a = self.acts[si]
                            self.relu_bits[si] = _ops.relu_bits_buffer(int(a.shape[1]), int(a.shape[2]), int(a.shape[3]), dev)
            # "Pre-scatter" backward: when EVERY tapped layer's gradient is produced by a kernel that can add to its
            # output (the split-K direct data-gradient, the F(4x4,3x3) data-gradients since ABI 7, the pooling backward,
            # the first layer's pixel gradient), the taps of all maps are scattered in ONE launch into zeroed buffers
            # before the backward pass and the producers accumulate -- 9 launches less per step.  It costs ONE fill of
            # every tapped gradient buffer plus the producers' extra read of their output, so it pays only where the maps
            # are tiny: measured (alternating runs on one box) 256 px 1.188 against 1.185 ms per step interleaved, 512 px
            # 1.904 against 1.860 (65 / 262 MB of fill) -- hence up to 160 x 160 pixels, the 64 / 128-px scales, by default
            # (STROTSS_PRESCATTER_MAX_PIXELS).  The tapped layers' gradient buffers are then slices of one allocation.
            max_px = int(os.environ.get("STROTSS_PRESCATTER_MAX_PIXELS", str(160 * 160)))
            self.prescatter = (halo is None and h * w <= max_px and self._can_prescatter()
                               and os.environ.get("STROTSS_PRESCATTER", "1") != "0")
            if self.prescatter:
                sizes = [a.numel() if i in set(self.taps) else 0 for i, a in enumerate(self.acts)]
                # (+ the pixel gradient at the end: ONE fill clears everything the single scatter launch adds into)
                self._tap_flat = torch.zeros(sum(sizes) + 3 * h * w, dtype=torch.float32, device=dev)
                offs = np.cumsum([0] + sizes)
                self.grads = [self._tap_flat[offs[i]:offs[i + 1]].view(a.shape) if sizes[i] else torch.empty_like(a)
                              for i, a in enumerate(self.acts)]
                self._gimg_in_flat = self._tap_flat[sum(sizes):].view(1, h, w, 3)
            else:
                self.grads = [torch.empty_like(a) for a in self.acts]
            self.gpools = [torch.empty_like(p) for p in self.pools]
            self.gimg = self._gimg_in_flat if self.prescatter else torch.empty((1, h, w, 3), dtype=torch.float32, device=dev)
        self.img = None
        # block ends on split-K layers: the max-pool (and its adjoint) inside the layer's finish kernel (ABI 8)
        self._pool_in_finish = halo is None and os.environ.get("STROTSS_POOL_IN_FINISH", "1") != "0"

    def _can_prescatter(self) -> bool:
        tapped = set(self.taps)
        n = len(self.acts)
        for step in self.plan:
            if step[0] != 'conv':
                continue
            _, li, (kind, si) = step
            if kind == 'conv' and si in tapped:           # layer li's data-gradient writes the tapped grads[si]
                L = self.p.layers[li]
                a = self.acts[li]
                if self.wtile[li] == 4:
                    continue                              # winograd_dgrad(accumulate=1): any F(4x4,3x3) route
                if self.wtile[li] != 0:                   # (F(2x2,3x3) overwrites; the direct kernels add, split-K or not)
                    return False
        return n - 1 in tapped                            # (the deepest layer is scattered into a zeroed buffer anyway)

    def _src(self, src):
        kind, i = src
        return self.img if kind == 'img' else (self.acts[i] if kind == 'conv' else self.pools[i])

    def forward(self, img: torch.Tensor) -> List[torch.Tensor]:
        self.img = img
        P = self.p
        pooled = set()                  # pools already written by the conv before them
        for si, step in enumerate(self.plan):
            if step[0] == 'pool':
                if step[1] not in pooled:
                    _ops.maxpool2_fwd(self.acts[step[2]], out=self.pools[step[1]],
                                      code=self.pool_codes[step[1]] if self.with_grad else None)
            else:
                _, li, src = step
                L = P.layers[li]
                x = self._src(src)
                if L["cin"] == 3:
                    _ops.conv3x3_c3_fwd(x, L["w_fwd"], L["bias"], out=self.acts[li], mean=P.mean, std=P.std,
                                        relu_bits_out=self.relu_bits[li])
                elif self.wtile[li]:
                    nxt = self.plan[si + 1] if si + 1 < len(self.plan) else None
                    pool_out = pool_code = None
                    if self.halo is None and nxt is not None and nxt[0] == 'pool' and nxt[2] == li:   # its pool rides along
                        pool_out = self.pools[nxt[1]]
                        pool_code = self.pool_codes[nxt[1]] if self.with_grad else None
                        pooled.add(nxt[1])
                    _ops.conv3x3_winograd_fwd(x, L["u_fwd"][self.wtile[li]], L["bias"], out=self.acts[li],
                                              pool_out=pool_out, pool_code=pool_code, relu_bits_out=self.relu_bits[li])
                else:
                    nxt = self.plan[si + 1] if si + 1 < len(self.plan) else None
                    a = self.acts[li]
                    if (self._pool_in_finish and nxt is not None and nxt[0] == 'pool' and nxt[2] == li
                            and _ops.conv3x3_direct_splits(int(a.shape[1]), int(a.shape[2]), L["cin"], L["cout"])):
                        # split-K layer at a block end: its finish kernel pools as well
                        _ops.conv3x3_relu_fwd(x, L["w_fwd"], L["bias"], out=a, pool_out=self.pools[nxt[1]],
                                              pool_code=self.pool_codes[nxt[1]] if self.with_grad else None)
                        pooled.add(nxt[1])
                    else:
                        _ops.conv3x3_relu_fwd(x, L["w_fwd"], L["bias"], out=a)
                if self.halo is not None:     # (the pooling launch that may follow then reads right rows only)
                    self.halo.refresh(self.acts[li], self.layer_level[li])
        return [self.acts[i] for i in self.taps]

    def backward(self, scatter: Callable[[int], None], scatter_all: Optional[Callable[[], None]] = None) -> torch.Tensor:
        """scatter_all (pre-scatter mode, see __init__): adds the taps of EVERY map (incl. the image's, into gimg) in
        one go; given and usable -> the per-layer `scatter` is not called."""
        assert self.with_grad
        P = self.p
        n_layers = len(self.acts)
        tapped = set(self.taps)
        pre = self.prescatter and scatter_all is not None
        # the deepest layer receives gradient from its tap only
        last = n_layers - 1
        if pre:
            self._tap_flat.zero_()
            if self.gimg.data_ptr() != self._gimg_in_flat.data_ptr():      # (a sharded engine points gimg into its reduce buffer)
                self.gimg.zero_()
            scatter_all()
            scatter = lambda li: None
        else:
            self.grads[last].zero_()
            scatter(last)
        # walk the plan backwards; grads[li] always holds the ReLU-masked gradient of layer li's output
        unpooled = set()                # pools whose adjoint the finish kernel of the layer behind them has already applied
        pool_src = {s_[1]: s_[2] for s_ in self.plan if s_[0] == 'pool'}
        for step in reversed(self.plan):
            if step[0] == 'pool':
                _, pi, src_layer = step
                if pi not in unpooled:
                    _ops.maxpool2_bwd(self.acts[src_layer], self.gpools[pi], out=self.grads[src_layer],
                                      code=self.pool_codes[pi], accumulate=pre and src_layer in tapped)
                if src_layer in tapped:
                    scatter(src_layer)
            else:
                _, li, src = step
                L = P.layers[li]
                kind, si = src
                if kind == 'img':
                    _ops.conv3x3_c3_dgrad(self.grads[li], L["w_bwd"], self.gimg, accumulate=pre, std=P.std)
                    scatter(-1)
                else:
                    wino = self.wtile[li] != 0
                    dgrad = _ops.conv3x3_winograd_dgrad if wino else _ops.conv3x3_dgrad
                    wts = L["u_bwd"][self.wtile[li]] if wino else L["w_bwd"]
                    if kind == 'conv':
                        if pre and si in tapped and wino:
                            dgrad(self.grads[li], wts, L["cin"], act_in=self.acts[si], out=self.grads[si],
                                  relu_bits=self.relu_bits[si], accumulate=True)
                        elif pre and si in tapped:
                            dgrad(self.grads[li], wts, L["cin"], act_in=self.acts[si], out=self.grads[si], accumulate=True)
                        elif wino and self.relu_bits[si] is not None:
                            dgrad(self.grads[li], wts, L["cin"], act_in=self.acts[si], out=self.grads[si],
                                  relu_bits=self.relu_bits[si])
                        else:
                            dgrad(self.grads[li], wts, L["cin"], act_in=self.acts[si], out=self.grads[si])
                        if si in tapped:
                            scatter(si)
                        if self.halo is not None:
                            self.halo.refresh(self.grads[si], self.layer_level[si])
                    else:
                        g = self.grads[li]
                        if (self._pool_in_finish and not wino
                                and _ops.conv3x3_direct_splits(int(g.shape[1]), int(g.shape[2]), L["cout"], L["cin"])):
                            src_layer = pool_src[si]
                            _ops.conv3x3_dgrad_unpool(g, wts, L["cin"], self.pool_codes[si], self.grads[src_layer],
                                                      accumulate=pre and src_layer in tapped)
                            unpooled.add(si)
                        else:
                            dgrad(g, wts, L["cin"], act_in=None, out=self.gpools[si])
                        if self.halo is not None:
                            self.halo.refresh(self.gpools[si], self.pool_level[si])
        return self.gimg


class _VGGFn(torch.autograd.Function):
    """autograd bridge for the public `VGG.__call__`: taps = f(img), d(img) from d(taps)."""

    @staticmethod
    def forward(ctx, img, vgg):
        h, w = int(img.shape[1]), int(img.shape[2])
        trunk = VGGTrunk(vgg.params, h, w, with_grad=img.requires_grad)
        taps = trunk.forward(img.detach().contiguous())
        ctx.trunk = trunk
        return tuple(taps)

    @staticmethod
    def backward(ctx, *gtaps):
        trunk = ctx.trunk

        def scatter(li):
            if li < 0:
                return
            k = trunk.taps.index(li)
            g = gtaps[k]
            if g is not None:
                # incoming gradient is w.r.t. the post-ReLU tap: apply the mask here
                trunk.grads[li].add_(g.contiguous() * (trunk.acts[li] > 0))
        gimg = trunk.backward(scatter)
        return gimg.clone(), None


class VGG:
    """feature extractor.  (reference: nn/model.py:17-55)"""

    def __init__(self, layers: Optional[list] = None, vgg_type: str = '16', use_keras_weight: bool = False,
                 name: Optional[str] = None, weights=None, seed: int = 0, device: str = "cuda"):
        vgg_type = str(vgg_type)
        assert vgg_type in ['16', '19']
        self.name = name
        if weights is None:
            w = synthetic_weights(vgg_type, seed)
        elif isinstance(weights, str):
            w = load_weights(weights, vgg_type)
        else:
            w = list(weights)
        self.params = VGGParams(w, vgg_type, layers or _STROTSS_DEFAULTS, device, use_keras_weight)
        self.mean = torch.tensor(self.params.mean, dtype=torch.float32, device=device).view(1, 1, 1, -1)
        self.std = torch.tensor(self.params.std, dtype=torch.float32, device=device).view(1, 1, 1, -1)

    def preprocess(self, inputs: torch.Tensor) -> torch.Tensor:
        # exposed for API parity; __call__ fuses it into the first conv kernel
        return (inputs - self.mean) / self.std

    def __call__(self, inputs: torch.Tensor) -> List[torch.Tensor]:
        if inputs.dim() != 4 or inputs.shape[0] != 1 or inputs.shape[-1] != 3:
            raise ValueError(f"expected a (1,H,W,3) image, got {tuple(inputs.shape)}")
        if inputs.requires_grad:
            return list(_VGGFn.apply(inputs, self))
        trunk = VGGTrunk(self.params, int(inputs.shape[1]), int(inputs.shape[2]), with_grad=False)
        return [t for t in trunk.forward(inputs.contiguous())]
