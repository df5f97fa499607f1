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
