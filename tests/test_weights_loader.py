"""`--weights PATH` (nn/model.py: load_weights; the reference fetches vgg16_norm.h5 over the network instead,
model.py:24-48): both .npz layouts round-trip synthetic weights exactly -- Keras names with HWIO kernels, and a
torchvision-style state dict with OIHW kernels, `classifier.*` matrices and other keys interleaved.  CPU: the loader
and the oracle VGG on the loaded weights; GPU: the HIP trunk's taps from a file equal the in-memory weights' taps."""
import os

import numpy as np
import pytest
import torch

from oracle import strotss_oracle as O

# torchvision's vgg16 / vgg19 `features` indices of the conv layers (ReLU and pool modules sit between them)
TV_INDEX = {'16': [0, 2, 5, 7, 10, 12, 14, 17, 19, 21, 24, 26, 28],
            '19': [0, 2, 5, 7, 10, 12, 14, 16, 19, 21, 23, 25, 28, 30, 32, 34]}


def _save_keras(path, weights, names):
    np.savez(path, **{f"{n}/kernel": w.numpy() for n, (w, _) in zip(names, weights)},
             **{f"{n}/bias": b.numpy() for n, (_, b) in zip(names, weights)})


def _save_torchvision(path, weights, vgg_type):
    d = {}
    for i, (w, b) in zip(TV_INDEX[vgg_type], weights):
        d[f"features.{i}.weight"] = w.permute(3, 2, 0, 1).contiguous().numpy()          # HWIO -> OIHW
        d[f"features.{i}.bias"] = b.numpy()
    # what a full state dict also holds: the classifier's 2-D matrices (their indices 0, 3, 6 interleave with features')
    for i, (o, k) in zip((0, 3, 6), ((8, 16), (8, 8), (4, 8))):
        d[f"classifier.{i}.weight"] = np.zeros((o, k), np.float32)
        d[f"classifier.{i}.bias"] = np.zeros((o,), np.float32)
    np.savez(path, **d)


@pytest.mark.parametrize("vgg_type", ['16', '19'])
def test_load_weights_round_trips_both_layouts(tmp_path, vgg_type):
    from nn.model import load_weights, synthetic_weights, vgg_config
    weights = synthetic_weights(vgg_type, 3)
    names = [it[0] for it in vgg_config(vgg_type) if it != 'pool']
    assert len(weights) == len(names) == len(TV_INDEX[vgg_type])
    _save_keras(tmp_path / "k.npz", weights, names)
    _save_torchvision(tmp_path / "t.npz", weights, vgg_type)
    for f in ("k.npz", "t.npz"):
        got = load_weights(str(tmp_path / f), vgg_type)
        assert len(got) == len(weights)
        for (w, b), (gw, gb) in zip(weights, got):
            assert gw.dtype == torch.float32 and torch.equal(gw, w) and torch.equal(gb, b)
    # a wrong architecture is an error, not a silently truncated list
    other = '19' if vgg_type == '16' else '16'
    with pytest.raises((ValueError, KeyError)):
        load_weights(str(tmp_path / "t.npz"), other)
    if vgg_type == '16':                 # Keras names: a VGG16 file lacks block3_conv4 ... of VGG19
        with pytest.raises((ValueError, KeyError)):
            load_weights(str(tmp_path / "k.npz"), '19')


def test_loaded_weights_drive_the_oracle_vgg(tmp_path):
    from nn.model import load_weights, synthetic_weights
    weights = synthetic_weights('16', 1)
    _save_torchvision(tmp_path / "t.npz", weights, '16')
    img = torch.rand(1, 24, 20, 3, generator=torch.Generator().manual_seed(0), dtype=torch.float64)
    a = O.VGG(load_weights(str(tmp_path / "t.npz")), dtype=torch.float64)(img)
    b = O.VGG(O.make_synthetic_vgg16_weights(1), dtype=torch.float64)(img)
    assert len(a) == 9 and all(torch.equal(x, y) for x, y in zip(a, b))


@pytest.mark.gpu
@pytest.mark.parametrize("layout", ["keras", "torchvision"])
def test_vgg_from_weight_file_matches_in_memory_weights(tmp_path, layout):
    from nn.model import VGG, synthetic_weights, vgg_config
    weights = synthetic_weights('16', 4)
    path = str(tmp_path / "w.npz")
    if layout == "keras":
        _save_keras(path, weights, [it[0] for it in vgg_config('16') if it != 'pool'])
    else:
        _save_torchvision(path, weights, '16')
    img = torch.rand(1, 48, 40, 3, generator=torch.Generator().manual_seed(2)).cuda()
    taps_file = VGG(weights=path, device="cuda")(img)
    taps_mem = VGG(weights=weights, device="cuda")(img)
    assert len(taps_file) == 9
    for a, b in zip(taps_file, taps_mem):
        assert torch.equal(a, b)
    ref = O.VGG(weights, dtype=torch.float64)(img.cpu().double())
    for a, b in zip(taps_file, ref):
        assert (a.cpu().double() - b).abs().max() < 2e-5 * max(1.0, float(b.abs().max()))
