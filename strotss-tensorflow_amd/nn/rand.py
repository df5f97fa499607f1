"""Reproducibility contract -- mirrors the reference's nn/rand.py:1-21.

The reference seeds python/numpy/TF with 0 and draws the sample coordinates from
`tf.random.Generator.from_seed(0)` (Philox) + `tf.random.shuffle`.  TF's streams cannot be
reproduced without TF, so "identical seeds" means here: identical *index sequences*, and every
consumer takes them as an input (`Sampling(..., rng=...)`, `Sampling.__call__(..., indices=...)`, the
index sets passed to `engine.StepEngine.step`).

Round 4: the product's own stream is COUNTER-BASED (`PhiloxStream`, Philox4x32-10) so that the
step's coordinates can be drawn on the device inside the captured step (csrc/draw.hip,
`strotss_index_draw`) and still be reproduced exactly on the host: `PhiloxStream` is a NumPy-Generator
look-alike (`integers`, `permutation`) whose draw number t is the same counter the kernel keeps in device
memory, so `make_indices_np(..., rng=PhiloxStream)` -- and the oracle's `make_indices`, which takes the same
object -- give the kernel's coordinates element for element (tests/test_index_parity.py).  A NumPy Generator is
still accepted everywhere an rng is (the committed fixtures were drawn from PCG64 streams).  The reference also
pins TF to one inter-op and one intra-op thread; there is no host compute left to pin in this build."""
import os
import random

import numpy as np
import torch

os.environ.setdefault('PYTHONHASHSEED', '0')

_M0, _M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_W0, _W1 = 0x9E3779B9, 0xBB67AE85
_MASK32 = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0: int, k1: int):
    """Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11) on arrays of counters:
    returns the four 32-bit outputs as uint64 arrays.  Bit for bit csrc/draw.hip: philox4x32_10."""
    c = [np.asarray(x, dtype=np.uint64) & _MASK32 for x in (c0, c1, c2, c3)]
    shape = np.broadcast(*c).shape
    c = [np.broadcast_to(x, shape).copy() for x in c]
    k0, k1 = int(k0) & 0xFFFFFFFF, int(k1) & 0xFFFFFFFF
    for _ in range(10):
        p0, p1 = _M0 * c[0], _M1 * c[2]                       # < 2^64: both factors are < 2^32
        hi0, lo0, hi1, lo1 = p0 >> np.uint64(32), p0 & _MASK32, p1 >> np.uint64(32), p1 & _MASK32
        c = [hi1 ^ c[1] ^ np.uint64(k0), lo1, hi0 ^ c[3] ^ np.uint64(k1), lo0]
        k0, k1 = (k0 + _W0) & 0xFFFFFFFF, (k1 + _W1) & 0xFFFFFFFF
    return c


class PhiloxStream:
    """The host twin of strotss_index_draw (include/strotss_hip.h): draw number `t` of the stream with key `seed`.

    One *draw* = one call of make_indices (strotss_utils.py:83-121): up to two `integers` calls (the grid offsets) followed by
    exactly one `permutation`, which ends the draw (t += 1).
        integers(0, n), k-th call of the draw   philox(ctr = (0, 1, t, 0))[k] mod n
        permutation(m)                          order of the positions j < m by (philox(ctr = (j >> 2, 0, t, 0))[j & 3], j)
    """

    def __init__(self, seed: int = 0, t: int = 0):
        self.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        self.t = int(t)
        self._k = 0

    @property
    def key(self):
        return self.seed & 0xFFFFFFFF, self.seed >> 32

    def integers(self, low, high=None, **_):
        if high is None:
            low, high = 0, low
        n = int(high) - int(low)
        assert n > 0
        assert self._k < 4, "a draw has at most four offset words (make_indices uses two)"
        v = int(philox4x32_10(0, 1, self.t & 0xFFFFFFFF, 0, *self.key)[self._k])
        self._k += 1
        return int(low) + v % n

    def keys(self, m: int) -> np.ndarray:
        """shuffle keys of the positions 0..m-1 of this draw: one Philox block per four positions"""
        q = np.arange((m + 3) // 4, dtype=np.uint64)
        out = philox4x32_10(q, 0, self.t & 0xFFFFFFFF, 0, *self.key)
        return np.stack(out, axis=1).reshape(-1)[:m]

    def permutation(self, m):
        m = int(m)
        order = np.lexsort((np.arange(m), self.keys(m)))            # by key, ties by position
        self.t += 1
        self._k = 0
        return order

    def permutation_head(self, m, k):
        """permutation(m)[:k] without sorting all m keys (the style draw of a 1024-px scale shuffles a million positions to
        keep 1024: strotss_utils.py:99-120): the k smallest (key, position) pairs, in order.  Ends the draw like permutation."""
        m, k = int(m), int(k)
        if k >= m:
            return self.permutation(m)
        keys = self.keys(m)
        T = np.partition(keys, k - 1)[k - 1]
        below = np.flatnonzero(keys < T)
        ties = np.flatnonzero(keys == T)[:k - below.size]            # flatnonzero is ascending: ties in position order
        sel = np.concatenate([below, ties])
        order = sel[np.lexsort((sel, keys[sel]))]
        self.t += 1
        self._k = 0
        return order

    def skip(self, draws: int) -> None:
        """advance past `draws` draws made elsewhere (on the device)"""
        self.t += int(draws)
        self._k = 0


SEED = 0
np_rng = np.random.default_rng(SEED)          # reference: np_rng (unused there as well)
index_rng = PhiloxStream(SEED)                # replaces tf_rng for the sampling coordinates


def seed_everything(seed: int = 0):
    """random.seed / np.random.seed / torch.manual_seed + a fresh index stream."""
    global np_rng, index_rng, SEED
    SEED = seed
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    np_rng = np.random.default_rng(seed)
    index_rng = PhiloxStream(seed)


seed_everything(0)
