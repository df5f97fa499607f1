"""Generates tests/golden/*.npz from the float64 oracle (oracle/strotss_oracle.py, numpy_ref.py).

The reference ships no golden vectors and cannot run here (TensorFlow absent), so these fixtures
pin the ORACLE (regression) and give the HIP path data-only expectations that travel to the GPU
box.  Re-run:  python tests/golden/make_golden.py     (deterministic: seeded generators only)
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import numpy_ref as R            # noqa: E402
from oracle import strotss_oracle as O       # noqa: E402


def feat(n, d, seed):
    rng = np.random.default_rng(seed)
    x = np.maximum(rng.standard_normal((n, d)), 0) + 0.01 * rng.random((n, d))
    x[:, :3] = rng.random((n, 3))
    return x


def losses_fixture():
    n, ns, d = 64, 64, 35
    x, y, c = feat(ns, d, 11), feat(n, d, 12), feat(n, d, 13)
    out = {"style": x, "pred": y, "content": c}
    l, g = R.self_similarity_fwd_bwd(y, c); out["selfsim_loss"], out["selfsim_grad"] = l, g
    l, g = R.relaxed_emd_cos_fwd_bwd(x, y); out["remd_loss"], out["remd_grad"] = l, g
    l, g = R.palette_remd_fwd_bwd(x[:, :3], y[:, :3]); out["palette_loss"], out["palette_grad"] = l, g
    l, g = R.moment_matching_fwd_bwd(x, y); out["moment_loss"], out["moment_grad"] = l, g
    for a in (16.0, 1.0):
        l, g = R.style_loss_fwd_bwd(x, y, a); out[f"style_loss_a{int(a)}"], out[f"style_grad_a{int(a)}"] = l, g
    out["cosine_xy"] = R.cosine_distance(x, y)
    np.savez_compressed(os.path.join(HERE, "losses_n64_d35.npz"), **out)


def img(h, w, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(1, h, w, 3, generator=g, dtype=torch.float64)
    return torch.nn.functional.avg_pool2d(x.permute(0, 3, 1, 2), 3, 1, 1).permute(0, 2, 3, 1).contiguous()


def jpeg_fixture():
    """Decoded pixels of the reference's two images as the product's loader sees them (PIL: libjpeg ISLOW DCT + fancy
    upsampling, the algorithm tf.image.decode_jpeg(dct_method='INTEGER_ACCURATE') names, nn/utils.py:44-57 of the
    reference): shape, SHA-256 of the uint8 bytes, channel sums and a 6 x 6 crop.  Pins the host image pipeline."""
    import hashlib
    from PIL import Image
    out = {}
    for name in ("content_im", "style_im"):
        with Image.open(os.path.join(HERE, name + ".jpg")) as im:
            px = np.array(im.convert("RGB"))
        out[name + "_shape"] = np.array(px.shape)
        out[name + "_sha256"] = np.frombuffer(hashlib.sha256(px.tobytes()).digest(), dtype=np.uint8)
        out[name + "_sums"] = px.reshape(-1, 3).astype(np.int64).sum(0)
        out[name + "_crop"] = px[100:106, 200:206].copy()
    np.savez_compressed(os.path.join(HERE, "jpeg_decode.npz"), **out)


def trace_fixture():
    """64-px single-scale trace: seeded synthetic VGG16 (seed 0), injected indices, 4 steps."""
    h = w = 64
    n_samples, steps, alpha, lr = 256, 4, 16.0, 2e-3
    denom = 2.0 + alpha + 1.0 / max(alpha, 1.0)
    content, style = img(h, w, 21), img(h, w, 22)
    weights = O.make_synthetic_vgg16_weights(0)
    vgg = O.VGG(weights, dtype=torch.float64)
    rng = np.random.default_rng(5)
    s_idx = O.make_indices(h, w, False, n_samples, rng)
    idx = np.stack([O.make_indices(h, w, True, n_samples, rng) for _ in range(steps)])
    with torch.no_grad():
        cf = [content] + vgg(content)
        sf = [style] + vgg(style)
        ss = O.sample_features(sf, s_idx, False)
    init = O.make_laplacian(content) + style.mean(dim=(1, 2), keepdim=True)
    variables = [v.clone().requires_grad_(True) for v in O.make_laplacian_pyramid(init)]
    rms = [torch.zeros_like(v) for v in variables]
    trace, grads0 = [], None
    for it in range(steps):
        res = O.train_step(variables, vgg, cf, ss, idx[it], alpha, denom)
        if it == 0:
            grads0 = [g.numpy().copy() for g in res["grads"]]
            pfeat0 = res["p_feat"].numpy().copy()
        with torch.no_grad():
            for v, r, g in zip(variables, rms, res["grads"]):
                O.rmsprop_update(v, r, g, lr)
        trace.append([float(res["loss"]), float(res["loss_c"]), float(res["loss_s"])])
    final = O.fold_laplacian_pyramid([v.detach() for v in variables])
    out = {"content": content.numpy(), "style": style.numpy(), "style_idx": s_idx, "idx": idx,
           "alpha": alpha, "denom": denom, "lr": lr, "init": init.numpy(), "trace": np.array(trace),
           "final": final.numpy(), "final_u8": O.postprocess(final), "pfeat0_sample": pfeat0[:8],
           "weight_checksum": float(sum(float(w.double().sum() + b.double().sum()) for w, b in weights))}
    for k, g in enumerate(grads0):
        out[f"grad0_level{k}"] = g
    np.savez_compressed(os.path.join(HERE, "trace_64px.npz"), **out)


def trajectory_inputs():
    """the seeded inputs of the 200-step trajectory (shared with tests/test_golden_gpu.py, which re-derives them)"""
    h = w = 64
    cfg = dict(h=h, w=w, n_samples=1024, steps=200, alpha=16.0, lr=2e-3, seed=7)
    cfg["denom"] = 2.0 + cfg["alpha"] + 1.0 / max(cfg["alpha"], 1.0)
    return cfg, img(h, w, 41), img(h, w, 42)


def _trajectory(dtype, init_noise=0.0):
    cfg, content, style = trajectory_inputs()
    h, w, n, steps = cfg["h"], cfg["w"], cfg["n_samples"], cfg["steps"]
    weights = O.make_synthetic_vgg16_weights(0)
    vgg = O.VGG(weights, dtype=dtype)
    content, style = content.to(dtype), style.to(dtype)
    rng = np.random.default_rng(cfg["seed"])
    s_idx = O.make_indices(h, w, False, n, rng)
    with torch.no_grad():
        cf = [content] + vgg(content)
        sf = [style] + vgg(style)
        ss = O.sample_features(sf, s_idx, False)
    init = O.make_laplacian(content) + style.mean(dim=(1, 2), keepdim=True)
    start = init
    if init_noise:
        start = init + init_noise * torch.randn(init.shape, generator=torch.Generator().manual_seed(99), dtype=torch.float64).to(dtype)
    variables = [v.clone().requires_grad_(True) for v in O.make_laplacian_pyramid(start)]
    rms = [torch.zeros_like(v) for v in variables]
    trace, idx_sum = [], 0.0
    import time
    t0 = time.time()
    for it in range(steps):
        idx = O.make_indices(h, w, True, n, rng)
        idx_sum += float((idx.astype(np.float64) * (np.arange(idx.size).reshape(idx.shape) % 97 + 1)).sum())
        res = O.train_step(variables, vgg, cf, ss, idx, cfg["alpha"], cfg["denom"])
        with torch.no_grad():
            for v, r, g in zip(variables, rms, res["grads"]):
                O.rmsprop_update(v, r, g, cfg["lr"])
        trace.append([float(res["loss"]), float(res["loss_c"]), float(res["loss_s"])])
        if it % 50 == 0:
            print(f"trajectory {dtype} step {it}: loss {trace[-1][0]:.5f}  ({time.time() - t0:.0f} s)", flush=True)
    final = O.fold_laplacian_pyramid([v.detach() for v in variables])
    return dict(trace=np.array(trace), final=final, s_idx=s_idx, idx_sum=idx_sum, init=init, weights=weights)


def trajectory_fixture():
    """200 free-running RMSprop steps of ONE 64-px scale (run_strotss.py:131-155: train_step + apply_gradients, first-scale
    initialisation and alpha, N = 1024 samples of the 4096 candidates, a fresh index set per step from one seeded stream,
    seeded synthetic VGG16) THREE times: in float64 (the trajectory the HIP engine is compared with,
    tests/test_golden_gpu.py::test_200_step_trajectory_against_the_float64_oracle), in float32 (the same restatement in the
    product's precision) and in float64 from a start image perturbed by 1e-7 -- the last two are the YARDSTICK: how far two
    correct runs of this optimisation drift apart (RMSprop's first update is 10*lr*sign(g); the L1 / hard-min losses flip
    signs and arg-mins), i.e. the tolerance a free-running comparison can state at all.  The index sets are not stored: the
    product's own draw (nn.strotss_utils.make_indices_np) is bit-equal to the oracle's (tests/test_index_parity.py); their
    checksum is."""
    r64 = _trajectory(torch.float64)
    r32 = _trajectory(torch.float32)
    rpt = _trajectory(torch.float64, init_noise=1e-7)
    np.savez_compressed(os.path.join(HERE, "trajectory_64px_200.npz"), trace=r64["trace"],
                        final=r64["final"].numpy().astype(np.float32), final_u8=O.postprocess(r64["final"]),
                        trace_f32=r32["trace"], final_u8_f32=O.postprocess(r32["final"]),
                        trace_perturbed=rpt["trace"], final_u8_perturbed=O.postprocess(rpt["final"]),
                        style_idx_sum=float(r64["s_idx"].astype(np.float64).sum()), idx_checksum=r64["idx_sum"],
                        init_u8=O.postprocess(r64["init"]),
                        weight_checksum=float(sum(float(w_.double().sum() + b.double().sum()) for w_, b in r64["weights"])))


def image_fixture():
    """bilinear resize / pyramid / sampling known outputs on a tiny non-square image."""
    x = img(21, 32, 31)
    out = {"x": x.numpy(), "down": O.resize_bilinear(x, 10, 16).numpy(), "up": O.resize_bilinear(x, 42, 64).numpy()}
    for k, p in enumerate(O.make_laplacian_pyramid(x)):
        out[f"pyr{k}"] = p.numpy()
    maps = [x, O.resize_bilinear(x, 10, 16), O.resize_bilinear(x, 5, 8)]
    idx = np.array([[0, 0], [20, 31], [7, 9], [13, 30], [3, 3]], np.float32)
    out["idx"] = idx
    out["sample_bilinear"] = O.sample_features(maps, idx, True).numpy()
    out["sample_nearest"] = O.sample_features(maps, idx, False).numpy()
    np.savez_compressed(os.path.join(HERE, "image_21x32.npz"), **out)


if __name__ == "__main__" and "--jpeg-only" in sys.argv:
    jpeg_fixture()
    sys.exit(0)

if __name__ == "__main__" and "--trajectory-only" in sys.argv:      # ~ten minutes of float64 CPU work
    torch.set_num_threads(8)
    trajectory_fixture()
    sys.exit(0)

if __name__ == "__main__":
    torch.set_num_threads(4)
    losses_fixture()
    image_fixture()
    trace_fixture()
    jpeg_fixture()
    if "--with-trajectory" in sys.argv:
        trajectory_fixture()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))
