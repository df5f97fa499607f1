#!/usr/bin/env python3
"""Writes two synthetic JPEGs (smooth random fields) for end-to-end CLI timing: python tools/make_synth_jpegs.py DIR SIZE"""
import sys
import numpy as np
from PIL import Image
d, s = sys.argv[1], int(sys.argv[2])
rng = np.random.default_rng(0)
for name in ("content.jpg", "style.jpg"):
    a = (rng.random((s // 16, s // 16, 3)) * 255).astype(np.uint8)
    Image.fromarray(a).resize((s, s), Image.BICUBIC).save(f"{d}/{name}", quality=95)
