"""Reproducibility contract -- mirrors the reference's nn/rand.py:1-21.

The reference seeds python/numpy/TF with 0 and draws the sample coordinates from
`tf.random.Generator.from_seed(0)` + `tf.random.shuffle`.  TF's Philox streams cannot be
reproduced without TF, so "identical seeds" means here: identical *index sequences*, produced by
`index_rng` (a NumPy Generator, seed 0) and injectable everywhere (`Sampling(..., rng=...)`,
`Sampling.__call__(..., indices=...)`, the index sets passed to `engine.StepEngine.step`).  The reference also pins TF to one inter-op and one intra-op
thread; there is no host compute left to pin in this build."""
import os
import random

import numpy as np
import torch

os.environ.setdefault('PYTHONHASHSEED', '0')

SEED = 0
np_rng = np.random.default_rng(SEED)          # reference: np_rng (unused there as well)
index_rng = np.random.default_rng(SEED)       # replaces tf_rng for the sampling coordinates


def seed_everything(seed: int = 0):
    """random.seed / np.random.seed / torch.manual_seed + a fresh index stream."""
    global np_rng, index_rng, SEED
    SEED = seed
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    np_rng = np.random.default_rng(seed)
    index_rng = np.random.default_rng(seed)


seed_everything(0)
