"""GPU: the HIP path against the committed golden fixtures alone (no oracle call): losses and
gradients at N=64, D=35; resize / pyramid / sampling; a 64-px 4-step optimisation trace."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
DEV = "cuda"


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32).to(DEV)


def test_losses_against_fixture():
    from nn import losses as L
    z = np.load(os.path.join(G, "losses_n64_d35.npz"))
    x, c = dev(z["style"]), dev(z["content"])
    checks = (("selfsim", lambda t: L.self_similarity(t, c), 3e-3), ("remd", lambda t: L.relaxed_emd(x, t), 1e-4),
              ("moment", lambda t: L.moment_matching(x, t), 3e-3))
    for name, fn, tol in checks:
        y = dev(z["pred"]).requires_grad_(True)
        l = fn(y)
        l.backward()
        assert abs(float(l) - float(z[f"{name}_loss"])) < 2e-5 * max(1.0, abs(float(z[f"{name}_loss"])))
        ref = z[f"{name}_grad"]
        assert np.linalg.norm(y.grad.cpu().numpy() - ref) / np.linalg.norm(ref) < tol, name
    assert np.abs(L.cosine_distance(x, dev(z["pred"])).cpu().numpy() - z["cosine_xy"]).max() < 2e-6


def test_image_ops_against_fixture():
    from nn import _ops, strotss_utils as SU
    z = np.load(os.path.join(G, "image_21x32.npz"))
    x = dev(z["x"])
    assert np.abs(_ops.resize_bilinear(x, 10, 16).cpu().numpy() - z["down"]).max() < 2e-6
    assert np.abs(_ops.resize_bilinear(x, 42, 64).cpu().numpy() - z["up"]).max() < 2e-6
    for k, p in enumerate(SU.make_laplacian_pyramid(x)):
        assert np.abs(p.cpu().numpy() - z[f"pyr{k}"]).max() < 2e-6
    maps = [x, dev(z["down"]), _ops.resize_bilinear(x, 5, 8)]
    s = SU.Sampling(8)
    idx = dev(z["idx"])
    assert np.abs(s.bilinear(maps, indices=idx).cpu().numpy() - z["sample_bilinear"]).max() < 2e-6
    assert np.abs(s(maps, indices=idx).cpu().numpy() - z["sample_nearest"]).max() < 2e-6


def test_trace_against_fixture():
    from nn import _ops, engine, strotss_utils as SU
    from nn.model import VGGParams, synthetic_weights
    z = np.load(os.path.join(G, "trace_64px.npz"))
    weights = synthetic_weights('16', 0)
    chk = float(sum(float(w.double().sum() + b.double().sum()) for w, b in weights))
    assert abs(chk - float(z["weight_checksum"])) < 1e-9
    params = VGGParams(weights, '16', None, DEV)
    content, style = dev(z["content"]), dev(z["style"])
    cfeat = engine.extract_features(params, content)
    sfeat = engine.extract_features(params, style)
    s_idx = dev(z["style_idx"])
    feats = _ops.hypercol_gather(sfeat, s_idx, False)
    target = engine.StyleTarget.build(feats, int(s_idx.shape[0]), 2179)
    init = SU.make_laplacian(content) + style.mean(dim=(1, 2), keepdim=True)
    assert np.abs(init.cpu().numpy() - z["init"]).max() < 2e-6
    eng = engine.StepEngine(params, cfeat, [target], init, float(z["alpha"]), float(z["denom"]), float(z["lr"]),
                            sample_size=256)
    idx = dev(z["idx"])
    eng.forward_backward([idx[0]])
    got = eng.losses()
    assert np.abs(eng.pf[0][:8, :2179].cpu().numpy() - z["pfeat0_sample"]).max() < 5e-5 * np.abs(z["pfeat0_sample"]).max()
    for j, k in enumerate(("loss", "loss_c", "loss_s")):
        assert abs(got[k] - z["trace"][0, j]) < 5e-5 * max(1.0, abs(z["trace"][0, j])), (k, got[k])
    for k in range(6):
        ref = z[f"grad0_level{k}"]
        rel = np.linalg.norm(eng.gvars[k].cpu().numpy() - ref) / np.linalg.norm(ref)
        assert rel < 5e-3, (k, rel)          # see GRAD_TOL in test_hip_engine.py
    eng.apply_gradients()
    for it in range(1, idx.shape[0]):
        eng.step([idx[it]])
        got = eng.losses()
        # free-running fp32 vs the fp64 fixture: divergence grows with every sign(g) flip
        assert abs(got["loss"] - z["trace"][it, 0]) < 2e-2 * it * abs(z["trace"][it, 0]), (it, got["loss"], z["trace"][it, 0])
    out = eng.stylized()
    assert np.abs(out.cpu().numpy() - z["final"]).mean() < 0.03
    # postprocess kernel on the FIXTURE's final float image == the fixture's uint8, byte for byte (the fixture's bytes are
    # the oracle's float32 arithmetic on the float32 cast of `final`, which is what dev() uploads)
    u8 = SU.postprocess(dev(z["final"])).cpu().numpy()
    assert u8.dtype == np.uint8 and np.array_equal(u8, z["final_u8"]), int((u8 != z["final_u8"]).sum())


def test_200_step_trajectory_against_the_float64_oracle():
    """north_star's "matching output pixels within a stated tolerance": 200 FREE-RUNNING steps of one 64-px scale (no
    re-synchronisation) on the HIP engine -- deterministic tap adjoint, the same seeded index stream, seeded weights --
    against the float64 oracle's trajectory committed in tests/golden/trajectory_64px_200.npz (generator:
    make_golden.py --trajectory-only; reference loop: run_strotss.py:131-155).  Bit equality is out of reach (RMSprop's first
    update is 10*lr*sign(g), the L1 / hard-min losses flip signs and arg-mins on f32 rounding): the fixture therefore also
    holds two YARDSTICK runs of the oracle itself (float32; float64 from a start image perturbed by 1e-7), which drift from the
    float64 run by 4 % of the loss per step on average, up to 13-16 % at single steps, 0.3 % in the mean loss of the last 50
    steps and 26.7-27.5 dB PSNR of the final image.  What is asserted is the tolerance DESIGN.md 6 states: last-50-step mean
    loss within 1.5 % of the oracle's, per-step deviations within 2.5x the yardsticks', final uint8 image no more than 2 dB
    below the yardsticks' PSNR (measured on MI355X: 4.3 % / 21.8 % / 27.5 dB)."""
    from nn import _ops, engine, strotss_utils as SU
    from nn.model import VGGParams, synthetic_weights
    z = np.load(os.path.join(G, "trajectory_64px_200.npz"))
    # the inputs are re-derived from their seeds exactly as the generator does (trajectory_inputs); only torch + numpy
    def img(h, w, seed):
        g = torch.Generator().manual_seed(seed)
        x = torch.rand(1, h, w, 3, generator=g, dtype=torch.float64)
        return torch.nn.functional.avg_pool2d(x.permute(0, 3, 1, 2), 3, 1, 1).permute(0, 2, 3, 1).contiguous()
    h = w = 64
    n, steps, alpha, lr, seed = 1024, 200, 16.0, 2e-3, 7
    denom = 2.0 + alpha + 1.0 / max(alpha, 1.0)
    content, style = img(h, w, 41).float().to(DEV), img(h, w, 42).float().to(DEV)
    weights = synthetic_weights('16', 0)
    assert abs(float(sum(float(w_.double().sum() + b.double().sum()) for w_, b in weights)) - float(z["weight_checksum"])) < 1e-9
    params = VGGParams(weights, '16', None, DEV)
    cfeat = engine.extract_features(params, content)
    sfeat = engine.extract_features(params, style)
    rng = np.random.default_rng(seed)
    s_idx = SU.make_indices_np(h, w, False, n, rng)
    assert float(s_idx.astype(np.float64).sum()) == float(z["style_idx_sum"])
    target = engine.StyleTarget.build(_ops.hypercol_gather(sfeat, dev(s_idx), False), n, 2179)
    init = SU.make_laplacian(content) + style.mean(dim=(1, 2), keepdim=True)
    assert np.abs(SU.postprocess(init).cpu().numpy().astype(int) - z["init_u8"].astype(int)).max() <= 1   # f32 vs f64 start image
    eng = engine.StepEngine(params, cfeat, [target], init, alpha, denom, lr, sample_size=n, deterministic=True)
    trace, idx_sum = [], 0.0
    for it in range(steps):
        idx = SU.make_indices_np(h, w, True, n, rng)
        idx_sum += float((idx.astype(np.float64) * (np.arange(idx.size).reshape(idx.shape) % 97 + 1)).sum())
        eng.step([dev(idx)])
        got = eng.losses()
        trace.append([got["loss"], got["loss_c"], got["loss_s"]])
    assert idx_sum == float(z["idx_checksum"])                     # the same 200 index sets as the oracle's run
    trace, ref = np.array(trace), z["trace"]

    def drift(tr, u8):
        """how far a trajectory is from the float64 one: per-step relative loss error, the mean loss of the last 50 steps
        (every step's loss is evaluated on its own random sample set: the windowed mean is the stable statistic), PSNR of the
        final uint8 image"""
        rel = np.abs(tr[:, 0] - ref[:, 0]) / np.abs(ref[:, 0])
        mse = float(((u8.astype(np.float64) - z["final_u8"].astype(np.float64)) ** 2).mean())
        return dict(max_head=float(rel[:20].max()), max_tail=float(rel[20:].max()), mean=float(rel.mean()),
                    window=float(tr[150:, 0].mean() / ref[150:, 0].mean() - 1.0), psnr=float(10.0 * np.log10(255.0 ** 2 / max(mse, 1e-12))))
    got = drift(trace, SU.postprocess(eng.stylized()).cpu().numpy())
    # the YARDSTICK, from the same fixture: the float32 run of the oracle itself and its float64 run from a start image
    # perturbed by 1e-7 -- two CORRECT runs of this optimisation drift this far apart (sign(g) first updates, L1 / hard-min
    # flips), so no free-running comparison can state a tighter tolerance than a small multiple of these
    y32, ypt = drift(z["trace_f32"], z["final_u8_f32"]), drift(z["trace_perturbed"], z["final_u8_perturbed"])
    print("TRAJECTORY 200 steps @64px  (max |rel loss| steps <20 / >=20, mean |rel|, last-50-step mean loss vs oracle, PSNR of the final uint8 image)")
    for name, d in (("HIP engine", got), ("oracle float32", y32), ("oracle float64, start + 1e-7", ypt)):
        print(f"  {name:30s} {d['max_head']:.3f} / {d['max_tail']:.3f}   {d['mean']:.4f}   {100 * d['window']:+.2f} %   {d['psnr']:.2f} dB")
    worst = {k: max(abs(y32[k]), abs(ypt[k])) for k in ("max_head", "max_tail", "mean", "window")}
    # the stated tolerance: the stable statistics tight, the per-step ones within 2.5x what correct runs show
    assert abs(got["window"]) < 0.015, got                     # yardsticks: 0.24 % and 0.29 %
    assert got["mean"] < 2.5 * worst["mean"] and got["max_tail"] < 2.5 * worst["max_tail"] and got["max_head"] < 2.5 * worst["max_head"], (got, worst)
    assert got["psnr"] >= min(y32["psnr"], ypt["psnr"]) - 2.0, (got["psnr"], y32["psnr"], ypt["psnr"])
