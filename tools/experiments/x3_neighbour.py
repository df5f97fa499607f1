#!/usr/bin/env python3
"""Aggressor of the shared-GPU experiment: loops ONE part of the step in its own process until STOPFILE
appears.  The part that makes a co-resident process's float-atomic tap adjoint lose contributions is the bf16x3 GEMM
core (LDS-DMA between bf16 MFMAs): `mstats` (strotss_moment_stats: centring + the 128 x 128-tile covariance GEMM).
usage: x3_neighbour.py STOPFILE SECONDS [full|fwd|bwd|loss|mstats|cosx3|normx3|gather|fold|scat]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "strotss-tensorflow_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import _region_worker as W
from nn import _ops
dev = torch.device("cuda", 0)
eng, idx = W.problem(dev, None, deterministic=False)
stop, limit = sys.argv[1], float(sys.argv[2])
mode = sys.argv[3] if len(sys.argv) > 3 else "mstats"
eng.forward_backward(idx[0]); torch.cuda.synchronize()
ns = int(idx[0][0].shape[0])
X3 = _ops.row_inv_norm_x3(eng.pf[0], ns)
def part():
    if mode == "full": eng.forward_backward(idx[0])
    elif mode == "fwd": eng.trunk.forward(eng.fold_forward())
    elif mode == "bwd": eng.trunk.backward(eng._scatter, eng._scatter_all)
    elif mode == "loss":
        for r in eng.my_regions: eng._losses(r, ns)
    elif mode == "mstats": _ops.moment_stats(eng.pf[0], ns, eng.d)
    elif mode == "cosx3": _ops.cosine_distance_x3(X3[1], X3[0], ns, X3[1], X3[0], ns, eng.ld)
    elif mode == "normx3": _ops.row_inv_norm_x3(eng.pf[0], ns)
    elif mode == "gather":
        for r in eng.my_regions: eng._gather(eng._mt_pred, eng._idx[r], eng.pf[r])
    elif mode == "fold": eng.fold_forward(); eng._fold_adjoint(); eng.apply_gradients()
    elif mode == "scat":
        for k in sorted(eng._layer_to_map): eng._scatter(k)
    else: raise SystemExit("mode?")
open(stop + ".ready", "w").write("1")
t0 = time.time(); n = 0
while not os.path.exists(stop) and time.time() - t0 < limit:
    for _ in range(20):
        part(); n += 1
    torch.cuda.synchronize()
print("neighbour", mode, "iterations", n, flush=True)
