"""STROTSS on MI355X -- the reference's command line and coarse-to-fine schedule on the HIP step engine.

    python run_strotss.py content_im.jpg style_im.jpg -o output.jpg
    python run_strotss.py c.jpg s.jpg -o out.jpg --content_mask cm.jpg --style_mask sm.jpg

Every reference flag is kept (run_strotss.py:165-178 there).  Additions:
  --start_level K   run only scales K..level-1 (0 = reference behaviour); the first executed scale is
                    initialised like the reference's first one and alpha starts at 16/2^K
  --seed, --weights PATH (.npz; the reference downloads vgg16_norm.h5, impossible offline -- without
                    --weights a seeded synthetic VGG is used)
  --log_every N     read the three scalars back every N steps (the reference formats them every step,
                    i.e. one device->host sync per step; default 10, 1 restores that)
  --no_graph        launch kernels one by one instead of replaying one hipGraph per step
  --strips          under torchrun (one process per GPU): ONE image on all GPUs -- every rank runs the trunk on its strip
                    of the image (+ halo) at the scales where that pays, two all-reduces per step (nn/parallel.py);
                    rank 0 writes the output
Under torchrun WITH masks (region-guided run, BASELINE config 4) the mask regions are dealt round-robin to the ranks:
every rank runs the replicated trunk forward, its own regions' samples + losses and their data-gradient, ONE RCCL
all-reduce sums the pixel gradient, every rank applies the identical update; rank 0 writes the output.
`--level` is coerced to int (the reference declares type=float, which breaks `range(args.level)`).
"""
import argparse
import os

import torch

from nn import engine as strotss_engine
from nn import rand, utils
from nn import strotss_utils as strotss
from nn.losses import moment_matching, relaxed_emd, self_similarity
from nn.model import VGG

utils.make_logger('STROTSS')
tqdm = __import__('tqdm.notebook' if utils.is_jupyter_env() else 'tqdm', fromlist=['tqdm']).tqdm

SAMPLE_SIZE = 1024          # Sampling(1024), run_strotss.py:68 of the reference


class ContentLoss:
    """ContentLoss()(target, prediction) = self_similarity(prediction, target)"""

    def __call__(self, target: torch.Tensor, prediction: torch.Tensor) -> torch.Tensor:
        return self_similarity(prediction, target)


class StyleLoss:
    """StyleLoss(target, alpha)(prediction) = moment + REMD + palette REMD / max(alpha, 1)"""

    def __init__(self, target: torch.Tensor, alpha: float, **kwargs):
        self.target = target
        self.inv_alpha = 1 / max(alpha, 1)

    def __call__(self, prediction: torch.Tensor) -> torch.Tensor:
        yuv = strotss.convert_rgb_to_yuv
        return (moment_matching(self.target, prediction) + relaxed_emd(self.target, prediction)
                + self.inv_alpha * relaxed_emd(yuv(self.target), yuv(prediction), distance='both'))


# --------------------------------------------------------------------------------------------------
def _load_masks(args):
    """(content_masks, style_masks) or ([None], [None]); one of the two flags alone is an error."""
    if bool(args.content_mask) != bool(args.style_mask):
        raise ValueError('Either both content and style masks must be provided or neither.')
    if not args.content_mask:
        return [None], [None]
    c_masks, s_masks = strotss.load_mask(args.content_mask, args.style_mask, max_size=args.max_size)
    utils.logger.info(f'Loaded {len(c_masks)} masks.')
    return c_masks, s_masks


def _initial_image(position: int, is_last: bool, previous, content, style, base_lr: float):
    """The image a scale starts from and its learning rate (reference run_strotss.py:78-88):
    first executed scale: Laplacian of the content + mean style colour; middle scales: upsampled previous
    result + Laplacian; last scale (when more than one runs): upsampled previous result, lr halved."""
    laplacian = strotss.make_laplacian(content)
    if position == 0:
        return laplacian + style.mean(dim=(1, 2), keepdim=True), base_lr
    if not is_last:
        return utils.resize_like(previous, content) + laplacian, base_lr
    return utils.resize_like(previous, content), base_lr / 2


def _style_targets(params, style, style_masks, sampling):
    """One StyleTarget per region: <= 1024 nearest-sampled hypercolumns of the style image (fixed for the
    scale), their inverse norms and first/second moments."""
    feats = strotss_engine.extract_features(params, style)
    width = sum(int(m.shape[-1]) for m in feats)
    targets = []
    for mask in style_masks:
        idx = sampling._make_indices(feats[0], False, mask)
        rows = strotss_engine._ops.hypercol_gather(feats, idx, False)
        targets.append(strotss_engine.StyleTarget.build(rows, int(idx.shape[0]), width))
    return targets


def _optimise_scale(eng, scl: int, content_masks, args, dev, quiet: bool = False, step_trace=None):
    """`max_iter` RMSprop steps; fresh sample coordinates every step (they are drawn inside the reference's
    traced train_step as well).  `step_trace`: a list receiving every step's loss dict (one host sync per step)."""
    from nn import parallel
    masks_here = [None if m is None else strotss.mask_at_scale(m, eng.h, eng.w) for m in content_masks]
    log_every = max(1, int(getattr(args, "log_every", 10)))
    # The index sets are drawn on the host (as make_indices does in the reference's step).  Uploading them from pageable
    # memory would block the host until the previous step has drained, leaving the GPU idle while the next draw is
    # computed (~0.3 ms per step, a quarter of a 64-px step): a small ring of pinned buffers + asynchronous copies lets
    # the host draw step k+1 while the GPU runs step k.
    # Round 4: the draw itself runs on the device, as the first kernel of the captured step (csrc/draw.hip; counter-based
    # Philox stream whose host twin is rand.index_rng, so the sequence is the one the host loop below would draw) -- wherever
    # the sample count cannot vary from step to step.  No upload, no ring, nothing for the host to do per step.
    stream = rand.index_rng
    if (dev.type == "cuda" and not getattr(args, "host_draw", False) and isinstance(stream, rand.PhiloxStream)
            and eng.enable_device_draw(stream.seed, stream.t, masks_here)):
        with tqdm(range(args.max_iter), disable=quiet) as bar:
            for it in bar:
                if it == 0 and not getattr(args, "no_graph", False):
                    eng.capture_graph()
                eng.step()
                if step_trace is not None:
                    step_trace.append(eng.losses())
                if (it + 1) % log_every == 0 or it + 1 == args.max_iter:
                    r = eng.losses()
                    bar.set_description(f"Scale: {scl:4d} - It: {it+1:4d}")
                    bar.set_postfix({k: f'{r[k]:.3f}' for k in ('loss', 'loss_c', 'loss_s')})
        stream.skip(args.max_iter * len(masks_here))          # the host twin moves past the draws the device made
        return
    ring, slots = 8, {}
    with tqdm(range(args.max_iter), disable=quiet) as bar:
        for it in bar:
            idx_np = [strotss.make_indices_np(eng.h, eng.w, True, SAMPLE_SIZE, rand.index_rng, mk) for mk in masks_here]
            offsets = None
            if eng.strips is not None:            # same seed on every rank: identical draws, every region's set ordered by owner
                offsets = []
                for r in range(len(idx_np)):
                    idx_np[r], off = parallel.sort_indices_by_strip(idx_np[r], eng.strips)
                    offsets.append(off)
            idx = []
            for r, a in enumerate(idx_np):
                if dev.type != "cuda":
                    idx.append(torch.from_numpy(a).to(dev))
                    continue
                buf, ev = slots.get((r, it % ring), (None, None))
                if buf is None:
                    buf = torch.empty((SAMPLE_SIZE, 2), dtype=torch.float32).pin_memory()
                if ev is not None:
                    ev.synchronize()              # the copy that last used this slot has run
                host = buf[:a.shape[0]]
                host.copy_(torch.from_numpy(a))
                idx.append(host.to(dev, non_blocking=True))
                ev = torch.cuda.Event()
                ev.record()
                slots[(r, it % ring)] = (buf, ev)
            if it == 0 and not getattr(args, "no_graph", False):
                eng.capture_graph(idx, offsets)
            eng.step(idx, offsets)
            if step_trace is not None:
                step_trace.append(eng.losses())
            if (it + 1) % log_every == 0 or it + 1 == args.max_iter:
                r = eng.losses()
                bar.set_description(f"Scale: {scl:4d} - It: {it+1:4d}")
                bar.set_postfix({k: f'{r[k]:.3f}' for k in ('loss', 'loss_c', 'loss_s')})


def run(args: argparse.Namespace, trace=None):
    """The reference's run(args) (run_strotss.py:43-161).  `trace` (a list) receives one dict per executed scale:
    scale index and size, lr, alpha, loss_denom, the image the scale starts from, every step's losses and the
    result -- what the parity test of the schedule compares with the oracle's run_scales."""
    timer = utils.Timer()
    timer.start()

    seed = int(getattr(args, "seed", 0))
    rand.seed_everything(seed)
    from nn import parallel
    rank, world = 0, 1
    masked = bool(getattr(args, "content_mask", None))
    if (getattr(args, "strips", False) or masked) and int(os.environ.get("WORLD_SIZE", "1")) > 1:
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count()))
        rank, world = parallel.init_from_env(torch.cuda.current_device())
    dev = utils.device()
    level, first = int(args.level), int(getattr(args, "start_level", 0))

    vgg = VGG(use_keras_weight=args.use_keras_weight, weights=getattr(args, "weights", None), seed=seed, device=dev)
    content = utils.load_image(args.content_path, max_size=args.max_size)
    style = utils.load_image(args.style_path, max_size=args.max_size)
    content_masks, style_masks = _load_masks(args)
    sampling = strotss.Sampling(SAMPLE_SIZE)

    # alpha = 16 (x3500 with Keras weights), halved after every scale -- also after skipped ones
    alpha = args.alpha * 16.0 * (3500 if args.use_keras_weight else 1) / 2.0 ** first
    stylized = None
    for position, i in enumerate(range(first, level)):
        scl = 2 << (5 + i)                                   # long side 64, 128, 256, ...
        scl_content, scl_style = utils.resize(content, scl), utils.resize(style, scl)
        stylized, lr = _initial_image(position, position > 0 and i == level - 1, stylized, scl_content, scl_style,
                                      args.lr)
        # --strips: ONE image sharded by rows (the only sharding that cuts trunk work), with or without mask regions; where
        # a scale is too small for strips to pay (strip_plan -> None) a masked run falls back to dealing its regions out
        plan = (parallel.strip_plan(int(scl_content.shape[1]), world, rank, halo=bool(getattr(args, "halo", False)))
                if world > 1 and getattr(args, "strips", False) else None)
        eng = strotss_engine.StepEngine(
            vgg.params, strotss_engine.extract_features(vgg.params, scl_content),
            _style_targets(vgg.params, scl_style, style_masks, sampling), stylized, alpha,
            loss_denom=2. + alpha + 1. / max(alpha, 1.), lr=lr, sample_size=SAMPLE_SIZE, strips=plan,
            dist_group=parallel.WORLD if (world > 1 and masked and plan is None) else None)
        rec = None
        if trace is not None:
            rec = dict(i=i, scl=scl, lr=lr, alpha=alpha, loss_denom=eng.loss_denom, init=stylized.clone(), steps=[],
                       hw=(eng.h, eng.w))
            trace.append(rec)
        _optimise_scale(eng, scl, content_masks, args, dev, quiet=rank != 0, step_trace=None if rec is None else rec["steps"])
        stylized = eng.stylized()
        if rec is not None:
            rec["final"] = stylized.clone()
        del eng
        alpha /= 2.

    final = strotss.postprocess(stylized)
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    timer.stop()
    if rank == 0:
        utils.logger.info(f"Done in {timer.elapsed_time:.2f}s.")
        utils.write_image(final, args.output_path)
    return final


# (flag, kwargs): the reference's flags first, then this build's additions
_FLAGS = (
    (("content_path",), dict(type=str)), (("style_path",), dict(type=str)),
    (("--content_mask",), dict(type=str, default=None)), (("--style_mask",), dict(type=str, default=None)),
    (("--max_size",), dict(type=int, default=None)), (("--lr",), dict(type=float, default=2e-3)),
    (("--level",), dict(type=float, default=4)), (("--max_iter",), dict(type=int, default=200)),
    (("--alpha",), dict(type=float, default=1.0)), (("--use_keras_weight",), dict(action='store_true')),
    (("--gpu_id",), dict(type=int, default=0)), (("--output_path", "-o"), dict(type=str, default="output.jpg")),
    (("--start_level",), dict(type=int, default=0)), (("--seed",), dict(type=int, default=0)),
    (("--weights",), dict(type=str, default=None)), (("--log_every",), dict(type=int, default=10)),
    (("--no_graph",), dict(action='store_true', help="eager kernel launches instead of one hipGraph per step")),
    (("--host_draw",), dict(action='store_true', help="draw the sample coordinates on the host every step (the same sequence; "
                                                      "the default draws them on the device inside the step)")),
    (("--strips",), dict(action='store_true', help="under torchrun: shard ONE image over the GPUs by image strips")),
    (("--halo",), dict(action='store_true', help="with --strips: per-layer halo EXCHANGE with the neighbouring ranks (16-row "
                                                 "windows margins, one row per layer and direction) instead of a 128-row recompute margin")),
)


def build_parser() -> argparse.ArgumentParser:
    parser = argparse.ArgumentParser(description=__doc__.splitlines()[0])
    for names, kw in _FLAGS:
        parser.add_argument(*names, **kw)
    return parser


if __name__ == "__main__":
    cli = build_parser().parse_args()
    utils.set_gpu(cli.gpu_id)
    run(cli)
