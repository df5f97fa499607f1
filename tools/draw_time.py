#!/usr/bin/env python3
"""HIP-event time of strotss_index_draw alone (fast and general selection path) at the candidate counts of the five scales.
usage: python tools/draw_time.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "strotss-tensorflow_amd")):
    sys.path.insert(0, p)
import torch
from nn import _ops
for S in (64, 128, 256, 1024):
    for general in (False, True):
        counters = torch.zeros(1, dtype=torch.int32, device="cuda")
        out = [torch.zeros((1024, 2), device="cuda")]
        for _ in range(5):
            _ops.index_draw(S, S, 1024, 0, counters, out, None, None, general_path=general)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(200):
            _ops.index_draw(S, S, 1024, 0, counters, out, None, None, general_path=general)
        e1.record(); torch.cuda.synchronize()
        most, _ = _ops.index_draw_counts(S, S, None)
        print(f"{S:5d} px  {most:6d} candidates  {'general' if general else 'fast   '} path  {e0.elapsed_time(e1) / 200 * 1e3:7.2f} us")
