"""Tap adjoint (strotss_hypercol_scatter) timed per map and as one launch on the ten maps of a scale (graph replays of 20 launches):\n   python tools/scatter_by_map.py 64     (STROTSS_SCATTER_DENSE=0 for atomics on every map)"""
import os, sys, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "strotss-tensorflow_amd")); sys.path.insert(0, ROOT)
from nn import _ops
dev = torch.device("cuda", 0)
S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
chans = [3, 64, 64, 128, 128, 256, 256, 256, 512, 512]
sizes = [S, S, S, S // 2, S // 2, S // 4, S // 4, S // 4, S // 8, S // 16]
torch.manual_seed(0)
maps = [torch.rand(1, s, s, c, device=dev) for s, c in zip(sizes, chans)]
gmaps = [torch.zeros_like(m) for m in maps]
n = 1024
idx = (torch.rand(n, 2, device=dev) * (S - 1)).contiguous()
ld = 2208
g = torch.randn(n, ld, device=dev)
def timeit(f, it=20):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for _ in range(3): f()
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=st):
            for _ in range(it): f()
    torch.cuda.synchronize()
    gr.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5): gr.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / (5 * it) * 1e3
print("all maps, one launch: %.1f us" % timeit(lambda: _ops.hypercol_scatter(maps, gmaps, idx, g)))
for k in range(10):
    print("map %d (%3d px^2 x %3d ch): %.1f us" % (k, sizes[k], chans[k], timeit(lambda: _ops.hypercol_scatter(maps, gmaps, idx, g, map_begin=k, map_end=k + 1))))
print("maps 0-7: %.1f us" % timeit(lambda: _ops.hypercol_scatter(maps, gmaps, idx, g, map_begin=0, map_end=8)))
print("maps 8-9: %.1f us" % timeit(lambda: _ops.hypercol_scatter(maps, gmaps, idx, g, map_begin=8, map_end=10)))
