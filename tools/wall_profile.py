#!/usr/bin/env python3
"""cProfile of bench.py's wall-clock-to-output run (the CLI's whole 5-scale schedule): where the time outside the
optimisation steps goes.  usage: wall_profile.py [top-N]"""
import cProfile, io, os, pstats, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "strotss-tensorflow_amd")):
    sys.path.insert(0, p)
import torch
import bench
dev = torch.device("cuda:0")
bench.wall_clock_to_output(dev)                       # warm: library load, allocator, first-use costs
pr = cProfile.Profile()
pr.enable()
r = bench.wall_clock_to_output(dev)
pr.disable()
print(r)
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(int(sys.argv[1]) if len(sys.argv) > 1 else 45)
print(s.getvalue())
