#!/usr/bin/env python3
"""Per-rank compute time of the image-strip sharding (nn/parallel.py) on ONE GPU: builds the engine of the rank with the
largest window for world = 2, 4, 8 and times its steps with the two all-reduces absent (torch.distributed not
initialised -> no-ops; the feature all-reduce is replaced by a copy of a full feature matrix).  Speed-up bound of one 1024-px image = t(world 1) / t(rank); the all-reduces (8.5 MiB + 12 MiB
per step over xGMI) come on top.  With `halo` as second argument: the per-layer halo-EXCHANGE plan (16-row margins); the row
exchanges are stubbed out as well (torch.distributed is not initialised), i.e. the time is the rank's compute alone and
the ~30 neighbour exchanges per step (one row of each layer, <= 0.26 MB, point to point) come on top.
Usage: python tools/strips_rank_time.py [scale] [halo]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "strotss-tensorflow_amd")):
    sys.path.insert(0, p)
import torch
import bench
from nn import parallel
from nn.model import VGGParams, synthetic_weights


def main():
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    halo = len(sys.argv) > 2 and sys.argv[2] == "halo"
    if halo:                                 # no process group here: the exchange object only has to count
        class _Stub:
            def __init__(self, plan, group=None): self.plan, self.messages = plan, 0
            def refresh(self, t, level):
                parallel.halo_rows(self.plan, int(t.shape[1]), level); self.messages += 2
        parallel.HaloExchange = _Stub
    dev = torch.device("cuda", 0)
    params = VGGParams(synthetic_weights('16', 0), '16', None, dev)
    base = None
    for world in (1, 2, 4, 8):
        plan = None
        if world > 1:
            plans = [parallel.strip_plan(S, world, r, halo=halo) for r in range(world)]
            if plans[0] is None:
                print(f"world {world}: strip_plan declines (windows cover too much of {S} rows)")
                continue
            plan = max(plans, key=lambda p: p.win1 - p.win0)
        eng, rng = bench.build_engine(params, S, dev, seed=0, strips=plan)
        if plan is None:
            idx = bench.index_stream(S, 16, rng, dev)
            step = lambda i: eng.step(list(idx[i % 16]))
            eng.capture_graph(list(idx[0]))
        else:
            idx, offs = bench.strip_index_stream(S, 16, rng, dev, plan)
            step = lambda i: eng.step(idx[i % 16], offs[i % 16])
            if "graph" in sys.argv:            # the three stages as three graphs (eager otherwise)
                eng.capture_graph(idx[0], offs[0])
            # stand-in for the feature all-reduce: the other ranks' rows come from the unsharded engine's matrix (left
            # zero they would tie every relaxed-EMD minimum and distort the loss kernels' time)
            parallel.allreduce_sum_ = (lambda e, ref: (lambda t, group=None: t.copy_(ref) if t is e._pf_all else t))(eng, ref_pf)
        for i in range(3):
            step(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 20
        for i in range(n):
            step(i)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        base = base or ms
        if plan is None:
            ref_pf = eng._pf_all.clone()
        rows = S if plan is None else plan.win1 - plan.win0
        if halo and plan is not None:
            print(f"    ({eng._halo.messages // 2 // 23} neighbour exchanges per step, stubbed) ", end="")
        print(f"world {world}: rank window {rows:4d} of {S} rows  {ms:6.2f} ms/step (no collectives, "
              f"{'hipGraph' if plan is None else ('3 hipGraphs' if eng._strip_graphs is not None else 'eager')})  -> compute-side speed-up bound {base / ms:4.2f}x")
        del eng


if __name__ == "__main__":
    main()
