#!/usr/bin/env python3
"""Where the CLI's wall clock goes OUTSIDE the optimisation loops (bench.py wall_clock_to_output's run): per scale, the host
wall time from the end of the previous scale's loop to the start of this one's (resize, Laplacian, content / style features,
style draw + statistics, engine construction) and the first step's graph capture; then the tail (postprocess, JPEG).
usage: python tools/setup_profile.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "strotss-tensorflow_amd")):
    sys.path.insert(0, p)
import torch
import bench, run_strotss
from nn import engine
dev = torch.device("cuda:0")
bench.wall_clock_to_output(dev)
marks = []
orig_opt, orig_cap = run_strotss._optimise_scale, engine.StepEngine.capture_graph
def opt(eng, scl, *a, **k):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = orig_opt(eng, scl, *a, **k)
    torch.cuda.synchronize(); marks.append(("loop", scl, t0, time.perf_counter()))
    return r
def cap(self, *a, **k):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = orig_cap(self, *a, **k)
    torch.cuda.synchronize(); marks.append(("capture", self.h, t0, time.perf_counter()))
    return r
run_strotss._optimise_scale, engine.StepEngine.capture_graph = opt, cap
t_start = time.perf_counter()
res = bench.wall_clock_to_output(dev)
t_end = time.perf_counter()
print(res["seconds"], "s wall clock (Timer scope)")
prev = None
for kind, s, a, b in marks:
    if kind == "loop":
        print(f"scale {s:5d}: setup before the loop {1e3 * (a - prev) if prev else float('nan'):8.1f} ms   loop {1e3 * (b - a):8.1f} ms")
        prev = b
    else:
        print(f"           graph capture at {s} px: {1e3 * (b - a):6.1f} ms (inside the loop time)")
