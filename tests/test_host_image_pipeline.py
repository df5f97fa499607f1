"""Host image pipeline (reference nn/utils.py:32-70, nn/strotss_utils.py:170-175): JPEG decode pinned
by a committed fixture of the reference's own two images, the always-JPEG writer (quality 100, 4:2:0 chroma, whatever
the file extension says), the float conversion x * (1/255), error behaviour.  No GPU needed: `load_image` without
`max_size` and `write_image` touch no kernel (the resize path is covered by the GPU tests)."""
import hashlib
import os

import numpy as np
import pytest
import torch

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("name", ["content_im", "style_im"])
def test_decode_matches_fixture(name):
    from nn import utils
    z = np.load(os.path.join(G, "jpeg_decode.npz"))
    u8 = utils.load_image(os.path.join(G, name + ".jpg"), dtype=torch.uint8, batch_expand=False).cpu().numpy()
    assert u8.dtype == np.uint8 and tuple(u8.shape) == tuple(z[name + "_shape"])
    assert np.array_equal(u8[100:106, 200:206], z[name + "_crop"])
    assert np.array_equal(u8.reshape(-1, 3).astype(np.int64).sum(0), z[name + "_sums"])
    assert hashlib.sha256(u8.tobytes()).digest() == z[name + "_sha256"].tobytes()
    # float load: (1, H, W, 3) in [0, 1], exactly uint8 * (1/255) in float32 (tf.image.convert_image_dtype)
    f = utils.load_image(os.path.join(G, name + ".jpg")).cpu()
    assert f.dtype == torch.float32 and tuple(f.shape) == (1,) + tuple(u8.shape)
    assert torch.equal(f[0], torch.from_numpy(u8).float() * (1.0 / 255.0))
    assert float(f.min()) >= 0.0 and float(f.max()) <= 1.0


def test_reference_image_sizes():
    """content_im.jpg is 481 x 321 (W x H), style_im.jpg 1200 x 1600."""
    z = np.load(os.path.join(G, "jpeg_decode.npz"))
    assert tuple(z["content_im_shape"]) == (321, 481, 3) and tuple(z["style_im_shape"]) == (1600, 1200, 3)


def test_writer_is_always_a_quality_100_420_jpeg(tmp_path):
    from PIL import Image
    from nn import utils
    yy, xx = np.mgrid[0:96, 0:128]
    img = np.stack([0.5 + 0.4 * np.sin(xx / 17.0), 0.5 + 0.4 * np.cos(yy / 11.0), 0.3 + 0.002 * (xx + yy)], -1)   # smooth
    t = torch.from_numpy(np.floor(img * 255).astype(np.float32))[None]   # the driver hands over uint8-valued floats
    for fname in ("out.jpg", "out.png"):                         # the reference's output.png is a JPEG as well
        path = str(tmp_path / fname)
        utils.write_image(t, path)
        with Image.open(path) as im:
            assert im.format == "JPEG" and im.size == (128, 96) and im.mode == "RGB"
            assert all(set(q) == {1} for q in im.quantization.values())          # quality 100: all-ones tables
            assert im.layer[0][1:3] == (2, 2) and im.layer[1][1:3] == (1, 1)       # 4:2:0: luma sampled 2 x 2
            back = np.asarray(im).astype(np.float64)
        err = back - np.asarray(t[0])
        psnr = 10 * np.log10(255.0 ** 2 / np.mean(err ** 2))
        assert psnr > 35.0, psnr
    with pytest.raises(ValueError):
        utils.write_image(torch.zeros(2, 4, 4, 3), str(tmp_path / "bad.jpg"))
    with pytest.raises(FileNotFoundError):
        utils.load_image(str(tmp_path / "missing.jpg"))


def _literal_load_mask(c_mask, s_mask, pixel_threth, sample_threth):
    """The reference's algorithm (nn/strotss_utils.py:178-201) written out naively: per-colour loops over np.unique rows."""
    c = c_mask // pixel_threth * pixel_threth
    s = s_mask // pixel_threth * pixel_threth
    rows, counts = np.unique(c.reshape(-1, 3), axis=0, return_counts=True)
    pairs = []
    for row, cnt in zip(rows, counts):
        if cnt < sample_threth:
            continue
        cm = np.all(c == row, axis=-1)
        sm = np.all(s == row, axis=-1)
        if cm.any() and sm.any():
            pairs.append((cm.astype(np.float32)[..., None], sm.astype(np.float32)[..., None]))
    return pairs


@pytest.mark.parametrize("pixel_threth", [255, 128])
def test_load_mask_pairs_regions_like_the_reference(tmp_path, pixel_threth):
    """load_mask on lossless colour-coded images: region order (ascending r, g, b), the size threshold, colours missing
    from the style image dropped, float (H, W, 1) 0/1 masks, bare Exception when nothing pairs (a19)."""
    from PIL import Image
    from nn import strotss_utils as SU
    rng = np.random.default_rng(0)
    palette = np.array([[0, 0, 0], [255, 0, 0], [0, 255, 0], [0, 0, 255], [255, 255, 0], [130, 140, 250]], np.uint8)
    c = palette[rng.integers(0, 4, size=(12, 16))].repeat(16, 0).repeat(16, 1)        # 192 x 256, blocks of 16 x 16
    c[:8, :8] = palette[4]                                                            # 64 px: under every threshold used
    c[100:180, 30:200] = palette[5]                                                   # only quantises away at 255
    s = palette[rng.integers(0, 3, size=(10, 10))].repeat(20, 0).repeat(20, 1)        # blue never occurs in the style
    s[50:120, 50:120] = palette[5]
    Image.fromarray(c, "RGB").save(tmp_path / "c.png")
    Image.fromarray(s, "RGB").save(tmp_path / "s.png")
    for thr in (1, 3000, 10000):
        want = _literal_load_mask(c, s, pixel_threth, thr)
        if not want:
            with pytest.raises(Exception, match="No mask found"):
                SU.load_mask(str(tmp_path / "c.png"), str(tmp_path / "s.png"), None, pixel_threth, thr)
            continue
        cm, sm = SU.load_mask(str(tmp_path / "c.png"), str(tmp_path / "s.png"), None, pixel_threth, thr)
        assert len(cm) == len(sm) == len(want)
        for a, b, (wa, wb) in zip(cm, sm, want):
            assert a.dtype == torch.float32 and tuple(a.shape) == (192, 256, 1) and tuple(b.shape) == (200, 200, 1)
            assert np.array_equal(a.numpy(), wa) and np.array_equal(b.numpy(), wb)
    with pytest.raises(Exception, match="No mask found"):
        SU.load_mask(str(tmp_path / "c.png"), str(tmp_path / "s.png"), None, pixel_threth, 10 ** 9)
