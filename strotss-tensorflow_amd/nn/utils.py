"""Image I/O, resize, logger, Timer, device selection -- mirrors the reference's nn/utils.py:9-114
on torch HIP tensors.  JPEG decode/encode use PIL (libjpeg ISLOW + fancy upsampling is PIL's
default, the same algorithm as tf.image.decode_jpeg(dct_method='INTEGER_ACCURATE'))."""
import logging
import os
import sys
import time
from typing import Optional

import numpy as np
import torch

from . import _ops

logger = logging.getLogger(__name__)


def make_logger(name: str):
    global logger
    logger = logging.getLogger(name)
    if not logger.handlers:
        sh = logging.StreamHandler(sys.stdout)
        sh.setFormatter(
            logging.Formatter('%(asctime)s [%(levelname)s] %(name)s: %(message)s', "%Y-%m-%d %H:%M:%S"))
        logger.addHandler(sh)
    logger.setLevel(logging.INFO)


def device() -> torch.device:
    return torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")


def _validate_and_get_shape(base: torch.Tensor):
    if base.dim() == 3:
        h, w, _ = base.shape
    elif base.dim() == 4:
        _, h, w, _ = base.shape
    else:
        raise ValueError(f"Invalid rank: {base.dim()}")
    return int(h), int(w)


def resize(image: torch.Tensor, max_size: Optional[int]) -> torch.Tensor:
    """reference nn/utils.py:32-37: long side -> max_size, sizes by Python doubles + int()."""
    if max_size is None:
        return image
    h, w = _validate_and_get_shape(image)
    factor = max(h / max_size, w / max_size)
    return _ops.resize_bilinear(image.float().contiguous(), int(h / factor), int(w / factor))


def resize_like(image: torch.Tensor, base: torch.Tensor) -> torch.Tensor:
    h, w = _validate_and_get_shape(base)
    return _ops.resize_bilinear(image.float().contiguous(), h, w)


def load_image(path: str, max_size: Optional[int] = None, dtype: torch.dtype = torch.float32,
               batch_expand: bool = True) -> torch.Tensor:
    """reference nn/utils.py:44-57.  uint8 stays uint8 unless `max_size` forces a (float) resize."""
    if not os.path.exists(path):
        raise FileNotFoundError(f"File not found: {path}")
    from PIL import Image
    arr = np.asarray(Image.open(path).convert("RGB"))
    img = torch.from_numpy(arr.copy()).to(device())
    if dtype.is_floating_point:
        img = img.to(dtype) * (1.0 / 255.0)          # tf.image.convert_image_dtype(uint8 -> float)
    img = resize(img, max_size)
    if batch_expand:
        return img[None]
    return img


def write_image(image: torch.Tensor, path: str):
    """reference nn/utils.py:60-70: ALWAYS a JPEG (quality 100, 4:2:0), whatever the extension."""
    rank = image.dim()
    assert rank in [3, 4], f"Invalid rank: {rank}"
    if rank == 4:
        if image.shape[0] != 1:
            raise ValueError(f"Batch size must be 1. Got {image.shape[0]}")
        image = image[0]
    from PIL import Image
    arr = image.detach().to("cpu").numpy().astype(np.uint8)
    Image.fromarray(arr, "RGB").save(path, format="JPEG", quality=100, subsampling=2)
    logger.info(f"Wrote image to {path}")


def set_gpu(index: int = 0):
    """reference nn/utils.py:73-85 (one visible device); here: torch.cuda.set_device."""
    n = torch.cuda.device_count()
    if n:
        if index >= n:
            raise ValueError(f"Invalid GPU ID: {index}")
        torch.cuda.set_device(index)
        logger.debug(f"Set GPU to {index}")
    else:
        logger.info("GPU not found.")


def is_jupyter_env():
    if 'get_ipython' in globals():
        shell = get_ipython().__class__.__name__  # type: ignore  # noqa: F821
        if shell in ('ZMQInteractiveShell', 'Shell'):
            return True
    return False


class Timer:
    def __init__(self):
        self._start = 0.
        self._stop = 0.
        self._elapsed = 0.

    def start(self):
        self._start = time.time()

    def stop(self):
        self._stop = time.time()
        self._elapsed = round(self._stop - self._start, 3)
        self._start = 0.
        self._stop = 0.

    @property
    def elapsed_time(self):
        return self._elapsed
