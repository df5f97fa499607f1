"""STROTSS on MI355X -- same command line and driver structure as the reference's run_strotss.py.

    python run_strotss.py content_im.jpg style_im.jpg -o output.jpg
    python run_strotss.py c.jpg s.jpg -o out.jpg --content_mask cm.jpg --style_mask sm.jpg

Reference flags are kept (run_strotss.py:165-178).  Additions: --start_level (run only scales
start_level..level-1; 0 = reference behaviour), --seed, --weights PATH (.npz; the reference fetches
vgg16_norm.h5 from the network, which is not possible here -- without --weights a seeded synthetic
VGG is used), --log_every (the reference formats three scalars every iteration, forcing a
device->host sync per step; default 10 here, 1 restores it).  `--level` is coerced to int (the
reference declares type=float, which makes `range(args.level)` raise when the flag is given).
"""
import argparse

import numpy as np
import torch

from nn import strotss_utils as strotss
from nn import engine as strotss_engine
from nn import rand, utils
from nn.losses import moment_matching, relaxed_emd, self_similarity
from nn.model import VGG

utils.make_logger('STROTSS')

if utils.is_jupyter_env():
    from tqdm.notebook import tqdm
else:
    from tqdm import tqdm


class ContentLoss:
    """reference run_strotss.py:21-24"""

    def __call__(self, target: torch.Tensor, prediction: torch.Tensor) -> torch.Tensor:
        return self_similarity(prediction, target)


class StyleLoss:
    """reference run_strotss.py:27-40"""

    def __init__(self, target: torch.Tensor, alpha: float, **kwargs):
        self.target = target
        self.inv_alpha = 1 / max(alpha, 1)

    def __call__(self, prediction: torch.Tensor) -> torch.Tensor:
        l_m = moment_matching(self.target, prediction)
        l_remd = relaxed_emd(self.target, prediction)
        target = strotss.convert_rgb_to_yuv(self.target)
        pred = strotss.convert_rgb_to_yuv(prediction)
        l_palette = relaxed_emd(target, pred, distance='both')
        return l_m + l_remd + (self.inv_alpha * l_palette)


def run(args: argparse.Namespace):
    timer = utils.Timer()
    timer.start()

    level = int(args.level)
    start_level = int(getattr(args, "start_level", 0))
    log_every = max(1, int(getattr(args, "log_every", 10)))
    rand.seed_everything(int(getattr(args, "seed", 0)))
    dev = utils.device()

    vgg = VGG(use_keras_weight=args.use_keras_weight, weights=getattr(args, "weights", None),
              seed=int(getattr(args, "seed", 0)), device=dev)

    content = utils.load_image(args.content_path, max_size=args.max_size)
    style = utils.load_image(args.style_path, max_size=args.max_size)

    if args.content_mask and args.style_mask:
        content_masks, style_masks = strotss.load_mask(args.content_mask, args.style_mask, max_size=args.max_size)
        utils.logger.info(f'Loaded {len(content_masks)} masks.')
        use_mask = True
    elif not args.content_mask and not args.style_mask:
        use_mask = False
        content_masks, style_masks = [None], [None]
    else:
        raise ValueError('Either both content and style masks must be provided or neither.')

    alpha = args.alpha * 16.0 * (3500 if args.use_keras_weight else 1)
    alpha /= 2.0 ** start_level                     # scales skipped by --start_level still halve alpha
    sample_size = 1024
    sampling = strotss.Sampling(sample_size)

    stylized = None
    executed = list(range(start_level, level))
    for n_exec, i in enumerate(executed):
        scl = 2 << (5 + i)

        scl_content = utils.resize(content, scl)
        scl_style = utils.resize(style, scl)

        laplacian = strotss.make_laplacian(scl_content)

        # init variables (run_strotss.py:81-88); the first EXECUTED scale takes the i == 0 branch
        lr = args.lr
        if n_exec == 0:
            stylized = laplacian + scl_style.mean(dim=(1, 2), keepdim=True)
        elif i < level - 1:
            stylized = utils.resize_like(stylized, scl_content) + laplacian
        else:
            stylized = utils.resize_like(stylized, scl_content)
            lr = args.lr / 2

        loss_denom = (2. + alpha + 1. / max(alpha, 1.))

        # content, style features (once per scale)
        content_feat = strotss_engine.extract_features(vgg.params, scl_content)
        style_feat = strotss_engine.extract_features(vgg.params, scl_style)
        targets = []
        for sm in style_masks:
            s_idx = sampling._make_indices(style_feat[0], False, sm)
            feats = strotss_engine._ops.hypercol_gather(style_feat, s_idx, False)
            d = sum(int(m.shape[-1]) for m in style_feat)
            targets.append(strotss_engine.StyleTarget.build(feats, int(s_idx.shape[0]), d))
        del style_feat

        eng = strotss_engine.StepEngine(vgg.params, content_feat, targets, stylized, alpha, loss_denom, lr,
                                        sample_size=sample_size)
        h, w = eng.h, eng.w
        region_masks = [strotss.mask_at_scale(m, h, w) if m is not None else None for m in content_masks]

        use_graph = not getattr(args, "no_graph", False)
        with tqdm(range(args.max_iter)) as pbar:
            for it in pbar:
                idx = [torch.from_numpy(strotss.make_indices_np(h, w, True, sample_size, rand.index_rng, mk)).to(dev)
                       for mk in region_masks]
                if it == 0 and use_graph:
                    eng.capture_graph(idx)
                eng.step(idx)
                if (it + 1) % log_every == 0 or it + 1 == args.max_iter:
                    result = eng.losses()
                    pbar.set_description(f"Scale: {scl:4d} - It: {it+1:4d}")
                    pbar.set_postfix({'loss': f'{result["loss"]:.3f}',
                                      'loss_c': f'{result["loss_c"]:.3f}',
                                      'loss_s': f'{result["loss_s"]:.3f}'})

        stylized = eng.stylized()
        del eng
        alpha /= 2.

    final = strotss.postprocess(stylized)
    if torch.cuda.is_available():
        torch.cuda.synchronize()

    timer.stop()
    utils.logger.info(f"Done in {timer.elapsed_time:.2f}s.")
    utils.write_image(final, args.output_path)
    return final


def build_parser() -> argparse.ArgumentParser:
    parser = argparse.ArgumentParser()
    parser.add_argument("content_path", type=str)
    parser.add_argument("style_path", type=str)
    parser.add_argument("--content_mask", type=str, default=None)
    parser.add_argument("--style_mask", type=str, default=None)
    parser.add_argument("--max_size", type=int, default=None)
    parser.add_argument("--lr", type=float, default=2e-3)
    parser.add_argument('--level', type=float, default=4)
    parser.add_argument("--max_iter", type=int, default=200)
    parser.add_argument("--alpha", type=float, default=1.0)
    parser.add_argument("--use_keras_weight", action='store_true')
    parser.add_argument('--gpu_id', type=int, default=0)
    parser.add_argument("--output_path", "-o", type=str, default="output.jpg")
    # additions (see module docstring)
    parser.add_argument("--start_level", type=int, default=0)
    parser.add_argument("--seed", type=int, default=0)
    parser.add_argument("--weights", type=str, default=None)
    parser.add_argument("--log_every", type=int, default=10)
    parser.add_argument("--no_graph", action="store_true", help="launch kernels eagerly instead of one hipGraph per step")
    return parser


if __name__ == "__main__":
    args = build_parser().parse_args()
    utils.set_gpu(args.gpu_id)
    run(args)
