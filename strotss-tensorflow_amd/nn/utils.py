"""Host-side helpers with the reference's names (nn/utils.py:9-114 there): logger, long-side resize,
JPEG load/store through PIL, device selection, a wall-clock Timer.  Tensors are torch HIP tensors;
the only arithmetic here (resize) runs on the bilinear HIP kernel.

JPEG notes: PIL decodes with libjpeg's ISLOW DCT + fancy upsampling, the algorithm
tf.image.decode_jpeg(dct_method='INTEGER_ACCURATE') names; the writer always emits a JPEG at
quality 100 with 4:2:0 chroma whatever the file extension says, as the reference's
tf.image.encode_jpeg call does (utils.py:68 there)."""
import logging
import os
import sys
import time
from typing import Optional, Tuple

import numpy as np
import torch

from . import _ops

_LOG_FORMAT = '%(asctime)s [%(levelname)s] %(name)s: %(message)s'
logger = logging.getLogger(__name__)


def make_logger(name: str):
    """Install a stdout handler on logger `name` and make it the module-level `logger`."""
    global logger
    log = logging.getLogger(name)
    if not any(isinstance(h, logging.StreamHandler) for h in log.handlers):
        handler = logging.StreamHandler(sys.stdout)
        handler.setFormatter(logging.Formatter(_LOG_FORMAT, "%Y-%m-%d %H:%M:%S"))
        log.addHandler(handler)
    log.setLevel(logging.INFO)
    logger = log


def device() -> torch.device:
    if torch.cuda.is_available():
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def _hw(t: torch.Tensor) -> Tuple[int, int]:
    """(H, W) of an HWC or 1HWC tensor; anything else is an error, as in the reference."""
    if t.dim() not in (3, 4):
        raise ValueError(f"Invalid rank: {t.dim()}")
    return int(t.shape[-3]), int(t.shape[-2])


_validate_and_get_shape = _hw      # the reference's private name


def resize(image: torch.Tensor, max_size: Optional[int]) -> torch.Tensor:
    """Scale so the LONG side becomes `max_size` (up or down).  The target size is computed exactly like
    the reference does -- Python doubles, then int() truncation: 321x481 at 64 -> 42x64."""
    if max_size is None:
        return image
    h, w = _hw(image)
    factor = max(h / max_size, w / max_size)
    return _ops.resize_bilinear(image.float().contiguous(), int(h / factor), int(w / factor))


def resize_like(image: torch.Tensor, base: torch.Tensor) -> torch.Tensor:
    return _ops.resize_bilinear(image.float().contiguous(), *_hw(base))


def load_image(path: str, max_size: Optional[int] = None, dtype: torch.dtype = torch.float32,
               batch_expand: bool = True) -> torch.Tensor:
    """RGB image -> (1,H,W,3) float in [0,1] (or uint8 when asked; a resize makes it float, as in TF)."""
    if not os.path.exists(path):
        raise FileNotFoundError(f"File not found: {path}")
    from PIL import Image
    with Image.open(path) as im:
        pixels = np.array(im.convert("RGB"))
    img = torch.from_numpy(pixels).to(device())
    if dtype.is_floating_point:
        img = img.to(dtype) * (1.0 / 255.0)
    img = resize(img, max_size)
    return img[None] if batch_expand else img


def write_image(image: torch.Tensor, path: str):
    if image.dim() == 4:
        if image.shape[0] != 1:
            raise ValueError(f"Batch size must be 1. Got {image.shape[0]}")
        image = image[0]
    assert image.dim() == 3, f"Invalid rank: {image.dim()}"
    from PIL import Image
    Image.fromarray(image.detach().cpu().numpy().astype(np.uint8), "RGB").save(
        path, format="JPEG", quality=100, subsampling=2)
    logger.info(f"Wrote image to {path}")


def set_gpu(index: int = 0):
    """Select ONE device (the reference hides the others from TF; here: torch.cuda.set_device)."""
    count = torch.cuda.device_count()
    if count == 0:
        logger.info("GPU not found.")
        return
    if index >= count:
        raise ValueError(f"Invalid GPU ID: {index}")
    torch.cuda.set_device(index)
    logger.debug(f"Set GPU to {index}")


def is_jupyter_env() -> bool:
    shell = globals().get('get_ipython')
    return bool(shell) and shell().__class__.__name__ in ('ZMQInteractiveShell', 'Shell')


class Timer:
    """start() / stop() / elapsed_time (seconds, rounded to ms) -- the scope `run()` reports as 'Done in'."""

    def __init__(self):
        self._t0: Optional[float] = None
        self._elapsed = 0.0

    def start(self):
        self._t0 = time.time()

    def stop(self):
        if self._t0 is not None:
            self._elapsed = round(time.time() - self._t0, 3)
        self._t0 = None

    @property
    def elapsed_time(self) -> float:
        return self._elapsed
