#!/usr/bin/env python3
"""Per-launch HIP-event times of the conv trunk inside real optimisation steps (eager launches), plus the
hipGraph step rate, for the current environment (STROTSS_X3, STROTSS_WINO_FUSED ... are read once per process).
Usage: python tools/trunk_layer_times.py [scale] [steps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "strotss-tensorflow_amd")):
    sys.path.insert(0, p)
import torch
import bench
from nn import _ops, model


def main():
    scale = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    dev = torch.device("cuda:0")
    params = model.VGGParams(model.synthetic_weights('16', 0), '16', None, dev)
    eng, rng = bench.build_engine(params, scale, dev, 0)
    idx = bench.index_stream(scale, 16, rng, dev)
    eng.capture_graph(list(idx[0]))
    bench.run_steps(eng, idx, 0, 5)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    bench.run_steps(eng, idx, 5, steps)
    e1.record()
    torch.cuda.synchronize()
    print(f"scale {scale}: {e0.elapsed_time(e1) / steps:.3f} ms/step over {steps} steps (X3={os.environ.get('STROTSS_X3', '1')})")
    rec = []

    def wrap(name):
        orig = getattr(_ops, name)

        def timed(*a, **k):
            a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a0.record()
            r = orig(*a, **k)
            a1.record()
            x = a[0]
            rec.append((name, tuple(x.shape[1:]), int((k.get("out") if k.get("out") is not None else r).shape[-1]), a0, a1))
            return r
        setattr(_ops, name, timed)
    for n in ("conv3x3_winograd_fwd", "conv3x3_winograd_dgrad", "conv3x3_relu_fwd", "conv3x3_dgrad"):
        wrap(n)
    n_e = 5
    keep = eng._graph                         # destroying the graph stalls the next eager launches: keep it
    eng._graph = None                         # per-launch events need eager launches
    eng.step(list(idx[15]))                   # one untimed eager step
    torch.cuda.synchronize()
    rec.clear()
    for i in range(n_e):
        eng.step(list(idx[i]))
    torch.cuda.synchronize()
    del keep
    agg = {}
    for name, shp, cout, a0, a1 in rec:
        key = (name.replace("conv3x3_", ""), shp, cout)
        t, c = agg.get(key, (0.0, 0))
        agg[key] = (t + a0.elapsed_time(a1), c + 1)
    tot = 0.0
    for key, (t, c) in agg.items():
        per = t / n_e
        tot += per
        print(f"  {key[0]:16s} in {str(key[1]):18s} -> {key[2]:4d}: {per * 1e3:8.1f} us/step ({c // n_e} launches)")
    print(f"  conv total {tot:.3f} ms/step (eager, event-timed)")


if __name__ == "__main__":
    main()
