"""The optimisation inner loop of run_strotss.py:104-148 as an explicit, pre-allocated kernel
sequence on one HIP stream (no tracing compiler, no autograd tape):

    fold (5 resize+add)  ->  VGG trunk forward  ->  hypercolumn gather (content, prediction)
    ->  self-similarity / moment / REMD / palette fused forward+backward into one (N, D) gradient
    ->  trunk data-gradient with the taps' scatter-adds interleaved  ->  pixel gradient
    ->  [RCCL all-reduce of the pixel gradient when mask regions are sharded over GPUs]
    ->  fold adjoint (5 transposed resizes)  ->  one multi-tensor RMSprop launch.

Nothing in `step()` synchronises with the host; the three logged scalars stay on the device until
`losses()` is called.  Index sets are inputs (reference: drawn inside the traced train_step,
strotss_utils.py:83-121), so identical index streams give comparable trajectories.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import _hip, _ops, parallel
from .model import VGGParams, VGGTrunk


@dataclass
class StyleTarget:
    """Style side of StyleLoss (run_strotss.py:27-31): sampled once per scale, then constant."""
    feats: torch.Tensor      # (pad32(ns), ld)
    ns: int
    inv_norm: torch.Tensor   # row_inv_norm(feats)
    mean: torch.Tensor       # (ld,)
    cov: torch.Tensor        # (ld, ld)
    panels: Optional[torch.Tensor] = None   # the rows as x3 panels (bf16x3 GEMM operand of the relaxed EMD), made once

    @staticmethod
    def build(feats: torch.Tensor, ns: int, d: int) -> "StyleTarget":
        rs = _ops.row_inv_norm(feats, ns)
        mean, cov = _ops.moment_stats(feats, ns, d)
        panels = _ops.row_inv_norm_x3(feats, ns)[1] if int(feats.shape[1]) % 32 == 0 else None
        return StyleTarget(feats, ns, rs, mean, cov, panels)


def extract_features(params: VGGParams, image: torch.Tensor) -> List[torch.Tensor]:
    """[image] + vgg(image)  (run_strotss.py:95-96), taps cloned out of a temporary trunk."""
    h, w = int(image.shape[1]), int(image.shape[2])
    trunk = VGGTrunk(params, h, w, with_grad=False)
    img = image.contiguous()
    return [img] + [t.clone() for t in trunk.forward(img)]


class _Capture:
    """`torch.cuda.graph(g)` without its gc.collect() + torch.cuda.empty_cache() (12 ms per capture, five captures per run:
    the step allocates nothing, so there is no allocator state to tidy up before a capture)."""

    def __init__(self, graph: "torch.cuda.CUDAGraph", stream: "torch.cuda.Stream"):
        self.graph, self.stream = graph, stream

    def __enter__(self):
        torch.cuda.synchronize()
        self._ctx = torch.cuda.stream(self.stream)
        self._ctx.__enter__()
        self.graph.capture_begin()
        return self

    def __exit__(self, *exc):
        if exc[0] is None:
            try:
                self.graph.capture_end()
            finally:
                self._ctx.__exit__(None, None, None)
            return False
        # the body raised: end the (now invalid) capture so the stream is usable again, but never let that second error
        # replace the one the caller has to see
        try:
            self.graph.capture_end()
        except Exception:
            pass
        self._ctx.__exit__(*exc)
        return False


class StepEngine:
    """One scale of the coarse-to-fine loop: owns the 6 pyramid variables and their RMSprop slots.

    `regions` > 1 is the masked path (run_strotss.py:104-125): one trunk forward/backward, one
    index set + loss group per region, loss = mean over regions.  With `dist_group` (a ProcessGroup, or
    `parallel.WORLD` for torch.distributed's default group), regions are dealt round-robin to the ranks and the
    pixel gradient is all-reduced (sum) before the fold adjoint, so every rank applies the identical update.
    The logged scalars ride in the tail of the same buffer: ONE all-reduce per step."""

    N_SCALARS = 4   # loss_c, l_moment, l_remd, l_palette per region

    def __init__(self, params: VGGParams, content_feat: Sequence[torch.Tensor],
                 style_targets: Sequence[StyleTarget], stylized: torch.Tensor, alpha: float,
                 loss_denom: float, lr: float, sample_size: int = 1024, levels: int = 5,
                 dist_group=None, rho: float = 0.99, eps: float = 1e-8, strips: Optional["parallel.StripPlan"] = None,
                 deterministic: Optional[bool] = None):
        dev = stylized.device
        self.params = params
        self.alpha, self.loss_denom, self.lr, self.rho, self.eps = float(alpha), float(loss_denom), float(lr), rho, eps
        self.inv_alpha = 1.0 / max(self.alpha, 1.0)
        self.content_feat = [c.contiguous() for c in content_feat]
        self.style_targets = list(style_targets)
        self.R = len(self.style_targets)
        h, w = int(stylized.shape[1]), int(stylized.shape[2])
        self.h, self.w = h, w
        # --- variables = make_laplacian_pyramid(stylized) (run_strotss.py:89), rms slots start at 0
        from .strotss_utils import make_laplacian_pyramid
        self.variables = [v.contiguous() for v in make_laplacian_pyramid(stylized.contiguous(), levels)]
        self.rms = [torch.zeros_like(v) for v in self.variables]
        self.sizes = [(int(v.shape[1]), int(v.shape[2])) for v in self.variables]
        # fold temporaries: f[k] = v[k] + up(f[k+1]); f[0] is the image
        self.fold = [torch.empty_like(v) for v in self.variables[:-1]]
        import os as _os
        self._fold_one_launch = None if _os.environ.get("STROTSS_FOLD_ONE_LAUNCH", "1") != "0" else False
        # image strips (nn/parallel.py): the trunk covers rows [win0, win1) only; everything else is full-size
        self.strips = strips
        win0, win1 = (strips.win0, strips.win1) if strips is not None else (0, h)
        if strips is not None:
            assert strips.h == h, "image strips: plan made for this scale"
        # halo-exchange strips: the trunk refreshes its window's outermost rows from the neighbours after every layer
        self._halo = parallel.HaloExchange(strips, dist_group) if (strips is not None and strips.halo) else None
        self.trunk = VGGTrunk(params, win1 - win0, w, with_grad=True, halo=self._halo)
        # region sharding (masked runs): pixel gradient + the regions' scalars in ONE buffer = one all-reduce
        self.group = dist_group
        self.rank, self.world = parallel.world_info(dist_group) if dist_group is not None else (0, 1)
        self.my_regions = parallel.regions_for_rank(self.R, self.rank, self.world)
        # the sharded structure ([pixel gradient | scalars] in one buffer, graph | all-reduce | graph) also in a world of one
        # when STROTSS_DIST_FORCE=1 asks for it (the RCCL rehearsal on one GPU)
        self.sharded = dist_group is not None and (self.world > 1 or parallel.force_collectives())
        self._reduce_buf = None
        if self.sharded and strips is None:
            self._reduce_buf = torch.zeros(3 * h * w + self.R * 8, dtype=torch.float32, device=dev)
            self.trunk.gimg = self._reduce_buf[:3 * h * w].view(1, h, w, 3)
        self.d = 3 + sum(int(a.shape[-1]) for a in (self.trunk.acts[i] for i in self.trunk.taps))
        self.ld = _ops.pad32(self.d)
        rows = _ops.pad32(sample_size)
        self.sample_size = sample_size
        self.cf = [torch.zeros((rows, self.ld), dtype=torch.float32, device=dev) for _ in range(self.R)]
        # the regions' prediction rows are slices of ONE buffer: image strips all-reduce them in one collective
        self._pf_all = torch.zeros((self.R, rows, self.ld), dtype=torch.float32, device=dev)
        self.pf = [self._pf_all[r] for r in range(self.R)]
        self.gp = [torch.zeros((rows, self.ld), dtype=torch.float32, device=dev) for _ in range(self.R)]
        self.scalars = (torch.zeros((self.R, 8), dtype=torch.float32, device=dev) if self._reduce_buf is None
                        else self._reduce_buf[3 * h * w:].view(self.R, 8))
        # gradient of the variables: level 0 aliases the pixel gradient
        if strips is not None:
            # full-size pixel gradient, zero outside the window; the trunk writes its window rows in place
            self.gimg_full = torch.zeros((1, h, w, 3), dtype=torch.float32, device=dev)
            self.trunk.gimg = self.gimg_full[:, win0:win1]
            assert self.trunk.gimg.is_contiguous()
            self.gvars = [self.gimg_full] + [torch.empty_like(v) for v in self.variables[1:]]
        else:
            self.gvars = [self.trunk.gimg] + [torch.empty_like(v) for v in self.variables[1:]]
        # hypercolumn descriptors (pointers are static for the life of the engine)
        self._img_window = self.fold[0][:, win0:win1]
        self.pred_maps = [self._img_window] + [self.trunk.acts[i] for i in self.trunk.taps]
        gmaps = [self.trunk.gimg] + [self.trunk.grads[i] for i in self.trunk.taps]
        # divisors and clipping are those of the FULL maps (content_feat holds them), also for a window
        shapes = [_ops.hwc(m)[:2] for m in self.content_feat]
        self.divs = _ops.map_divisors(shapes)
        windows = None
        if strips is not None:
            windows = []
            for k, (m, (fh, _)) in enumerate(zip(self.pred_maps, shapes)):
                shift = (h // fh).bit_length() - 1 if fh < h else 0      # number of 2x2 pools above map k
                assert win0 % (1 << shift) == 0 and (win0 >> shift) + int(m.shape[1]) <= fh
                windows.append((win0 >> shift, fh))
        # (halo-exchange strips: the adjoint scatters ALL samples and drops the taps outside the window)
        self._mt_pred = _hip.make_maps(self.pred_maps, self.divs, gmaps, windows, window_drop=self._halo is not None)
        self._mt_content = _hip.make_maps(self.content_feat, self.divs)
        self._layer_to_map = {li: k + 1 for k, li in enumerate(self.trunk.taps)}
        self._layer_to_map[-1] = 0
        if strips is not None:            # strips shard the IMAGE: every rank runs every region's (replicated) losses
            self.my_regions, self.world = list(range(self.R)), 1
        self._ns = [0] * self.R
        # strips: the block [begin, end) of each region's (owner-ordered) samples this rank gathers and scatters lives in
        # DEVICE memory (strotss_maps_t.sample_range): it changes every step, a captured graph reads it at replay
        self._mt_pred_own = None
        if strips is not None:
            self._range_dev = torch.zeros((self.R, 2), dtype=torch.int32, device=dev)
            self._mt_pred_own = []
            for r in range(self.R):
                m = _hip.MapsT.from_buffer_copy(self._mt_pred)
                m.sample_range = self._range_dev[r].data_ptr()
                self._mt_pred_own.append(m)
        self._idx: List[Optional[torch.Tensor]] = [None] * self.R
        # deterministic mode (STROTSS_DETERMINISTIC=1): the tap adjoint as a sorted scatter, one plan per region and step
        # (no float atomics -> bitwise reproducible steps; the reference asks TF for the same: nn/rand.py:4-8)
        import os
        # default: off with a GPU to itself; ON when two ranks of the group sit on one card (parallel.ranks_share_a_gpu)
        env = os.environ.get("STROTSS_DETERMINISTIC")
        self.deterministic = bool(deterministic) if deterministic is not None else \
            (env == "1" if env is not None else parallel.ranks_share_a_gpu(dist_group))
        self._plans = None
        if self.deterministic:
            nb = _hip.lib().strotss_hypercol_scatter_plan_bytes(len(self.pred_maps))
            self._plans = [torch.empty(nb, dtype=torch.uint8, device=dev) for _ in range(self.R)]
        # bench.py's measurement hook (strotss_debug_winograd_stages) is process-wide: an engine must never be built, captured
        # or stepped under a partial stage mask, the convolutions would silently skip kernels
        if _hip.lib().strotss_debug_winograd_stages(7) != 7:
            raise _hip.StrotssHipError("strotss_debug_winograd_stages was left at a partial mask (a timing pass did not "
                                       "restore it): results since then are meaningless")
        self.steps_done = 0
        self._draw = None                 # device-side index draw (enable_device_draw): counters, masks, output buffers
        self._graph = None
        self._graph_post = None           # sharded regions: the part of the step after the all-reduce
        self._strip_graphs = None         # image strips: the three stages between the two all-reduces
        self._graph_drawn = False         # the captured graph starts with the draw kernel (no per-step upload)
        self._graph_idx: List[torch.Tensor] = []
        self._graph_n: List[int] = []

    # ------------------------------------------------------------------ pieces of the step
    def fold_forward(self) -> torch.Tensor:
        """img = fold_laplacian_pyramid(variables)   (strotss_utils.py:159-163)"""
        if self._fold_one_launch is not False:            # one launch for the whole fold (csrc/image.hip: fold_pyramid_kernel)
            if _ops.fold_pyramid(self.variables, self.fold[0]) is not None:
                self._fold_one_launch = True
                return self.fold[0]
            self._fold_one_launch = False                 # a pyramid shape the kernel does not take: level by level
        t = self.variables[-1]
        for k in range(len(self.variables) - 2, -1, -1):
            hk, wk = self.sizes[k]
            _ops.resize_bilinear(t, hk, wk, 1.0, self.variables[k], out=self.fold[k])
            t = self.fold[k]
        return self.fold[0]

    def _gather(self, maps_t, idx, out):
        n = idx.shape[0]
        _hip.check(_hip.lib().strotss_hypercol_gather(_hip.C.byref(maps_t), idx.data_ptr(), n, 1, out.data_ptr(),
                                                      self.ld, _hip.stream_ptr()), "hypercol_gather")

    def _gather_both(self, maps_pred, idx, r: int) -> None:
        """content rows, prediction rows and the zero fill of the gradient rows of region r in ONE launch"""
        n = idx.shape[0]
        _hip.check(_hip.lib().strotss_hypercol_gather2(_hip.C.byref(self._mt_content), _hip.C.byref(maps_pred), idx.data_ptr(), n,
                                                       1, self.cf[r].data_ptr(), self.pf[r].data_ptr(), self.ld,
                                                       self.gp[r].data_ptr(), int(self.gp[r].shape[0]), _hip.stream_ptr()),
                   "hypercol_gather2")

    def _losses(self, r: int, n: int, zeroed: bool = False):
        """(alpha*loss_c + loss_s)/loss_denom/R and its gradient w.r.t. the sampled prediction.  zeroed: the gradient rows
        were cleared by _gather_both already."""
        st = self.style_targets[r]
        pf, cf, gp, sc = self.pf[r], self.cf[r], self.gp[r], self.scalars[r]
        if n < pf.shape[0]:
            pf[n:].zero_(); cf[n:].zero_()
        if not zeroed:
            gp.zero_()
        base = 1.0 / (self.loss_denom * self.R)
        if st.panels is not None and _ops.step_losses_available():
            # one call: 13 launches instead of 21, the three forward GEMMs in one of them (bit for bit the four calls below)
            _ops.step_losses_fwd_bwd(pf, cf, n, self.d, st.feats, st.inv_norm, st.panels, st.ns, st.mean, st.cov,
                                     self.alpha * base, base, base, self.inv_alpha * base, gp, sc[0:], sc[1:], sc[2:], sc[3:])
            return
        _ops.selfsim_fwd_bwd(pf, cf, n, self.d, self.alpha * base, gp, sc[0:])
        _ops.moment_fwd_bwd(st.mean, st.cov, pf, n, self.d, base, gp, sc[1:])
        # the relaxed EMD borrows the prediction rows' norms and x3 panels from the content loss's workspace (the moment
        # term has its own) and the style rows' panels from the StyleTarget
        _ops.remd_cos_fwd_bwd_after_selfsim(st.feats, st.inv_norm, st.panels, st.ns, pf, n, self.d, base, gp, sc[2:])
        _ops.palette_remd_fwd_bwd(st.feats, st.ns, pf, n, self.inv_alpha * base, gp, sc[3:])

    def _scatter_maps(self, r: int):
        """descriptor + sample count of region r's tap adjoint: strips (recompute margin) = this rank's block of the samples
        (device-side range); halo-exchange strips = every sample, taps outside the window dropped; otherwise all samples."""
        if self.strips is not None and self._halo is None:
            return self._mt_pred_own[r], self._ns[r]
        if self.strips is not None:
            return self._mt_pred, self._ns[r]
        return self._mt_pred, int(self._idx[r].shape[0])

    def _scatter(self, layer_index: int, k_end: Optional[int] = None):
        k = self._layer_to_map[layer_index]
        k_end = k + 1 if k_end is None else k_end
        for r in self.my_regions:
            mt, n = self._scatter_maps(r)
            idx, gp = self._idx[r][:n], self.gp[r][:n]
            if self.deterministic:
                _ops.hypercol_scatter_sorted(mt, self._plans[r], n, gp, relu_mask_from=1, map_begin=k, map_end=k_end)
            else:
                _ops.hypercol_scatter(self.pred_maps, None, idx, gp, relu_mask_from=1, map_begin=k, map_end=k_end, maps_t=mt)

    def _scatter_all(self):
        """every map's taps in one launch per region (pre-scatter backward of the small scales, nn/model.py)"""
        self._scatter(-1, len(self.pred_maps))

    def forward_backward(self, indices: Sequence[torch.Tensor], strip_offsets: Optional[Sequence[int]] = None) -> None:
        """train_step (run_strotss.py:131-142 / 104-125): fills self.gvars and self.scalars.
        With image strips every region's `indices[r]` must be ordered by owning rank and `strip_offsets[r]` be its world + 1
        block offsets (parallel.sort_indices_by_strip); one region: the offsets list itself is accepted too."""
        assert len(indices) == self.R
        if self.strips is not None:
            self._strip_inputs(indices, strip_offsets)
            self._strip_stage_a()
            parallel.allreduce_sum_(self._pf_all, self.group)     # rows of the other ranks' samples arrive here: ONE collective
            self._strip_stage_b()
            parallel.allreduce_sum_(self.gimg_full, self.group)   # windows overlap by the margins: sum
            self._fold_adjoint()
            return
        self._pixel_gradient(indices)
        self._reduce()
        self._fold_adjoint()

    def _pixel_gradient(self, indices: Sequence[torch.Tensor]) -> None:
        """fold, trunk forward, this rank's regions' samples + losses, trunk data-gradient -> trunk.gimg, scalars"""
        if self.sharded:
            self.scalars.zero_()          # regions owned by other ranks arrive through the all-reduce
        img = self.fold_forward()
        self.trunk.forward(img)
        for r in self.my_regions:
            idx = _hip.require(indices[r], "indices")
            n = int(idx.shape[0])
            assert 0 < n <= self.sample_size and idx.shape[1] == 2
            self._idx[r] = idx
            self._gather_both(self._mt_pred, idx, r)
            if self.deterministic:
                _ops.hypercol_scatter_plan(self._mt_pred, idx, self._plans[r])
            self._losses(r, n, zeroed=True)
        if self.my_regions:
            self.trunk.backward(self._scatter, self._scatter_all)
        else:
            self.trunk.gimg.zero_()

    def _reduce(self) -> None:
        """sharded regions: ONE all-reduce(sum) over [pixel gradient | scalars] (RCCL over xGMI; 12 MiB at 1024^2)"""
        if self._reduce_buf is not None:
            parallel.allreduce_sum_(self._reduce_buf, self.group)

    def _fold_adjoint(self) -> None:
        """gvars[k] = up^T(gvars[k-1]): adjoint of the fold"""
        if self._fold_one_launch is not False and _ops.fold_pyramid_adjoint(self.gvars):     # two levels per launch
            return
        for k in range(1, len(self.variables)):
            hk, wk = self.sizes[k]
            _ops.resize_bilinear_adjoint(self.gvars[k - 1], hk, wk, out=self.gvars[k])

    # ---- image strips: the step in three stages with an all-reduce between them
    def _strip_inputs(self, indices: Sequence[torch.Tensor], offsets) -> None:
        """Host side of a strips step (never captured): every region's index set (ordered by owning rank) and this rank's
        block of it, the block bounds going to device memory."""
        if self.R == 1 and offsets is not None and not isinstance(offsets[0], (list, tuple)):
            offsets = [offsets]
        assert offsets is not None and len(offsets) == self.R
        bounds = []
        for r in range(self.R):
            idx = _hip.require(indices[r], "indices")
            n = int(idx.shape[0])
            off = offsets[r]
            assert len(off) == self.strips.world + 1 and off[-1] == n
            assert 0 < n <= self.sample_size and idx.shape[1] == 2
            self._idx[r], self._ns[r] = idx, n
            bounds.append((int(off[self.strips.rank]), int(off[self.strips.rank + 1])))
        # a fresh pinned block per step (the caching host allocator hands it out again only after this copy has run: the
        # host may be several steps ahead of the GPU)
        host = torch.tensor(bounds, dtype=torch.int32)
        self._range_dev.copy_(host.pin_memory() if self._range_dev.is_cuda else host, non_blocking=True)

    def _strip_stage_a(self, indices=None, offsets=None) -> None:
        """fold (replicated), trunk forward on the window; per region: content rows (all, from the replicated full maps),
        prediction rows of THIS rank's samples (the others stay zero for the all-reduce).  Capturable: nothing here
        depends on the block bounds on the host.  (indices, offsets given: _strip_inputs first -- the eager form.)"""
        if indices is not None:
            self._strip_inputs(indices, offsets)
        self.fold_forward()
        self.trunk.forward(self._img_window)
        self._pf_all.zero_()
        for r in range(self.R):
            idx = self._idx[r]
            self._gather_both(self._mt_pred_own[r], idx, r)

    def _strip_stage_b(self) -> None:
        """losses of every region on the assembled features (replicated), backward of this rank's rows through its window."""
        for r in range(self.R):
            self._losses(r, self._ns[r], zeroed=True)
        if self.deterministic:
            for r in range(self.R):
                mt, n = self._scatter_maps(r)
                _ops.hypercol_scatter_plan(mt, self._idx[r][:n], self._plans[r])
        if self._halo is not None:
            # every rank back-propagates (gradient crosses the strip borders through the exchanges, and every sample's
            # taps that land in this window are scattered here); only the OWN rows of the pixel gradient are kept
            self.trunk.backward(self._scatter)
            self.gimg_full[:, :self.strips.own0].zero_()
            self.gimg_full[:, self.strips.own1:].zero_()
            return
        self.trunk.backward(self._scatter, self._scatter_all)     # (a rank without samples back-propagates zeros)
        # rows outside the window still hold the previous step's all-reduced sum
        self.gimg_full[:, :self.strips.win0].zero_()
        self.gimg_full[:, self.strips.win1:].zero_()

    # ---- the step's sample coordinates drawn on the device (csrc/draw.hip), inside the step and inside its graph
    def enable_device_draw(self, seed: int, t0: int = 0, masks: Optional[Sequence] = None) -> bool:
        """Draw every step's index sets on the device (reference: Sampling._make_indices inside the traced train_step,
        strotss_utils.py:83-121 / run_strotss.py:136, 115): region r of step s uses draw number t0 + s*R + r of the Philox
        stream with key `seed` -- what `make_indices_np(..., rng=rand.PhiloxStream(seed, t0))` called region by region, step
        by step, returns on the host.  masks: per region a boolean (h, w) host array at this scale, or None.
        Returns False (and changes nothing) where the device draw does not apply: image strips (the host orders every set
        by owning rank), a grid of more than 32768 candidates, or a region that an unlucky offset leaves with fewer
        candidates than samples (the sample count must be the same at every step: it shapes the launches)."""
        if self.strips is not None:
            return False
        masks = list(masks) if masks is not None else [None] * self.R
        assert len(masks) == self.R
        most, least = _ops.index_draw_counts(self.h, self.w, masks)
        if most > 32768 or least < self.sample_size or self.R > _hip.MAX_DRAW_REGIONS or self.sample_size > 1024:
            return False
        dev = self.variables[0].device
        counters = torch.tensor([int(t0) + r for r in range(self.R)], dtype=torch.int64).to(torch.int32).to(dev)
        mask_dev = [None if m is None else torch.from_numpy(np.ascontiguousarray(m, dtype=np.uint8)).to(dev) for m in masks]
        idx = [torch.zeros((self.sample_size, 2), dtype=torch.float32, device=dev) for _ in range(self.R)]
        self._draw = dict(seed=int(seed), t0=int(t0), counters=counters, masks=mask_dev, idx=idx)
        return True

    def draws_done(self) -> int:
        """draws consumed so far by the device stream (to advance the host twin: rand.PhiloxStream.skip)"""
        return 0 if self._draw is None else self.steps_done * self.R

    def _draw_indices(self) -> List[torch.Tensor]:
        dr = self._draw
        _ops.index_draw(self.h, self.w, self.sample_size, dr["seed"], dr["counters"], dr["idx"], dr["masks"])
        return dr["idx"]

    def apply_gradients(self) -> None:
        """opt.apply_gradients (run_strotss.py:148): Keras RMSprop, all 6 tensors in one launch."""
        _ops.rmsprop_step(self.variables, self.rms, self.gvars, self.lr, self.rho, self.eps)

    def step(self, indices: Optional[Sequence[torch.Tensor]] = None, strip_offsets: Optional[Sequence[int]] = None) -> None:
        """One optimisation step.  indices None: the index sets are drawn on the device at the head of the step
        (enable_device_draw first); otherwise they are the caller's (injected: tests, fixtures, strips)."""
        if indices is None:
            assert self._draw is not None, "step() without indices needs enable_device_draw()"
            if self._graph is not None and self._graph_drawn:
                self._graph.replay()
                if self._graph_post is not None:
                    self._reduce()
                    self._graph_post.replay()
            else:
                self.forward_backward(self._draw_indices())
                self.apply_gradients()
            self.steps_done += 1
            return
        if self.strips is not None:
            if self._strip_graphs is not None and all(int(i.shape[0]) == self._graph_n[r] for r, i in enumerate(indices)):
                for dst, src in zip(self._graph_idx, indices):
                    dst.copy_(src, non_blocking=True)
                self._strip_inputs(self._graph_idx, strip_offsets)
                ga, gb, gc = self._strip_graphs                       # graph | all-reduce | graph | all-reduce | graph
                ga.replay()
                parallel.allreduce_sum_(self._pf_all, self.group)
                gb.replay()
                parallel.allreduce_sum_(self.gimg_full, self.group)
                gc.replay()
            else:
                self.forward_backward(indices, strip_offsets)
                self.apply_gradients()
            self.steps_done += 1
            return
        if self._graph is not None and not self._graph_drawn and all(int(i.shape[0]) == self._graph_n[r] for r, i in enumerate(indices)):
            for dst, src in zip(self._graph_idx, indices):
                dst.copy_(src, non_blocking=True)
            self._graph.replay()
            if self._graph_post is not None:       # sharded regions: graph | all-reduce | graph
                self._reduce()
                self._graph_post.replay()
        else:
            self.forward_backward(indices)
            self.apply_gradients()
        self.steps_done += 1

    def capture_graph(self, example_indices: Optional[Sequence[torch.Tensor]] = None, example_offsets=None) -> None:
        """Capture forward_backward + apply_gradients into ONE hipGraph (the ~100 launches of a step
        replay as one submission; the 64-256 px scales are otherwise bound by host launch rate).  Index
        sets are copied into static buffers before each replay; a step whose index counts differ from
        the captured ones runs eagerly.  The captured step does not advance the optimisation: the
        variables / RMSprop slots are snapshotted around the warm-up and capture passes.
        Sharded regions (world > 1): the all-reduce stays outside -- TWO graphs, [fold .. pixel gradient] and
        [fold adjoint + RMSprop], with the collective launched between their replays.  Image strips run eagerly
        (recompute margin; example_offsets = the block offsets of example_indices): THREE graphs, [fold .. gathers],
        [losses .. pixel gradient of the window], [fold adjoint + RMSprop], the two all-reduces between their replays; the
        block of samples a rank owns changes every step and is read from device memory (strotss_maps_t.sample_range).
        Halo-exchange strips run eagerly (their trunk exchanges rows from Python after every layer)."""
        if self.strips is not None:
            if self._halo is None and example_offsets is not None:
                self._capture_strip_graphs(example_indices, example_offsets)
            return
        drawn = example_indices is None        # the draw kernel is the first node of the graph: nothing to upload per step
        state = self.variables + self.rms
        if drawn:
            assert self._draw is not None, "capture_graph() without indices needs enable_device_draw()"
            state = state + [self._draw["counters"]]       # warm-up and capture passes draw too: put the counters back
            self._graph_idx = self._draw["idx"]
            self._graph_n = [self.sample_size] * self.R
        else:
            self._graph_idx = [i.clone() for i in example_indices]
            self._graph_n = [int(i.shape[0]) for i in example_indices]
        snap = [t.clone() for t in state]
        head = (lambda: self._pixel_gradient(self._draw_indices())) if drawn else (lambda: self._pixel_gradient(self._graph_idx))
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):              # warm-up on the side stream (workspace allocation)
            head()
            self._fold_adjoint()
            self.apply_gradients()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g, post = torch.cuda.CUDAGraph(), None
        if self.sharded:
            with _Capture(g, side):
                head()
            post = torch.cuda.CUDAGraph()
            with _Capture(post, side):
                self._fold_adjoint()
                self.apply_gradients()
        else:
            with _Capture(g, side):
                head()
                self._fold_adjoint()
                self.apply_gradients()
        for t, s0 in zip(state, snap):
            t.copy_(s0)
        self._graph, self._graph_post, self._graph_drawn = g, post, drawn

    # ------------------------------------------------------------------ read-outs (host sync)
    def _capture_strip_graphs(self, example_indices, example_offsets) -> None:
        snap = [t.clone() for t in self.variables + self.rms]
        self._graph_idx = [i.clone() for i in example_indices]
        self._graph_n = [int(i.shape[0]) for i in example_indices]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):              # one whole eager step first: workspaces, collectives, library state
            self.forward_backward(self._graph_idx, example_offsets)
            self.apply_gradients()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self._strip_inputs(self._graph_idx, example_offsets)
        torch.cuda.synchronize()
        graphs = [torch.cuda.CUDAGraph() for _ in range(3)]
        with _Capture(graphs[0], side):
            self._strip_stage_a()
        with _Capture(graphs[1], side):
            self._strip_stage_b()
        with _Capture(graphs[2], side):
            self._fold_adjoint()
            self.apply_gradients()
        for t, s0 in zip(self.variables + self.rms, snap):
            t.copy_(s0)
        self._strip_graphs = tuple(graphs)

    def losses(self) -> dict:
        s = self.scalars.detach().cpu().numpy().astype(np.float64)
        lc = s[:, 0]
        ls = s[:, 1] + s[:, 2] + self.inv_alpha * s[:, 3]
        loss = ((self.alpha * lc + ls) / self.loss_denom).mean()
        return {"loss": float(loss), "loss_c": float(lc.mean()), "loss_s": float(ls.mean()),
                "l_moment": float(s[:, 1].mean()), "l_remd": float(s[:, 2].mean()),
                "l_palette": float(s[:, 3].mean())}

    def stylized(self) -> torch.Tensor:
        return self.fold_forward().clone()


# -------------------------------------------------------------------------------------------
def smoke_check(O, np_, torch_) -> None:
    """__graft_entry__.smoke(): one tiny optimisation step on cuda:0 checked against the oracle."""
    from .model import synthetic_weights
    dev = torch_.device("cuda", 0)
    torch_.cuda.set_device(dev)
    g = torch_.Generator().manual_seed(0)
    content = torch_.rand(1, 32, 32, 3, generator=g, dtype=torch_.float32)
    style = torch_.rand(1, 32, 32, 3, generator=g, dtype=torch_.float32)
    weights = synthetic_weights('16', 0)
    rng = np_.random.default_rng(0)
    s_idx = O.make_indices(32, 32, False, 256, rng)
    idx = O.make_indices(32, 32, True, 256, rng)
    alpha, denom, lr = 16.0, 18.0625, 2e-3
    # oracle (float64)
    vgg = O.VGG(weights, dtype=torch_.float64)
    c64, s64 = content.double(), style.double()
    with torch_.no_grad():
        cf = [c64] + vgg(c64); sf = [s64] + vgg(s64)
        ss = O.sample_features(sf, s_idx, False)
    init = O.make_laplacian(c64) + s64.mean(dim=(1, 2), keepdim=True)
    variables = [v.clone().requires_grad_(True) for v in O.make_laplacian_pyramid(init)]
    ref = O.train_step(variables, vgg, cf, ss, idx, alpha, denom)
    # HIP
    params = VGGParams(weights, '16', None, dev)
    cfeat = extract_features(params, content.to(dev))
    sfeat = extract_features(params, style.to(dev))
    sfe = _ops.hypercol_gather(sfeat, torch_.from_numpy(s_idx).to(dev), False)
    st = StyleTarget.build(sfe, s_idx.shape[0], 2179)
    eng = StepEngine(params, cfeat, [st], init.float().to(dev), alpha, denom, lr, sample_size=256)
    eng.step([torch_.from_numpy(idx).to(dev)])
    torch_.cuda.synchronize()
    got = eng.losses()
    assert abs(got["loss"] - float(ref["loss"])) < 1e-4 * max(1.0, abs(float(ref["loss"]))), (got, float(ref["loss"]))
    g0 = eng.gvars[0].cpu().double()
    rel = float((g0 - ref["grads"][0]).norm() / ref["grads"][0].norm())
    assert rel < 2e-2, rel
    print(f"smoke ok: loss {got['loss']:.6f} (oracle {float(ref['loss']):.6f}), pixel-grad rel err {rel:.2e}")
