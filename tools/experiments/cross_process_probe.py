#!/usr/bin/env python3
"""Victim side of the shared-GPU experiment: repeats several kernels of the step on bitwise-constant inputs
and compares every result with the first one.  Run it next to tools/experiments/x3_neighbour.py (another process on the same GPU).
usage: cross_process_probe.py ITERATIONS"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "strotss-tensorflow_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import _region_worker as W
from nn import _ops, engine as E
dev = torch.device("cuda", 0)
eng, idx = W.problem(dev, None, deterministic=False)
det = E.StepEngine(eng.params, eng.content_feat, eng.style_targets, eng.stylized(), eng.alpha, eng.loss_denom, eng.lr,
                   sample_size=256, deterministic=True)
eng.forward_backward(idx[0]); det.forward_backward(idx[0]); torch.cuda.synchronize()
tr = eng.trunk
maps = sorted(k for k in eng._layer_to_map if k >= 0)
ns = int(idx[0][0].shape[0])

def atomic_scatter():            # every tapped layer's gradient buffer, float atomics
    for li in maps: tr.grads[li].zero_()
    for li in maps: eng._scatter(li)
    return [tr.grads[li].clone() for li in maps]
def sorted_scatter():
    for li in maps: det.trunk.grads[li].zero_()
    for li in maps: det._scatter(li)
    return [det.trunk.grads[li].clone() for li in maps]
def gather():
    for r in eng.my_regions: eng._gather(eng._mt_pred, eng._idx[r], eng.pf[r])
    return [eng.pf[r].clone() for r in eng.my_regions]
def trunk_forward():
    tr.forward(eng.fold_forward())
    return [a.clone() for a in tr.acts]
def losses():
    for r in eng.my_regions: eng._losses(r, ns)
    return [eng.gp[r].clone() for r in eng.my_regions] + [eng.scalars.clone()]
def det_step():
    det.forward_backward(idx[0])
    return [g.clone() for g in det.gvars]
checks = {"atomic_scatter": (atomic_scatter, 1e-5), "sorted_scatter": (sorted_scatter, 0.0), "gather": (gather, 0.0),
          "trunk_forward": (trunk_forward, 0.0), "losses": (losses, 0.0), "deterministic_step": (det_step, 0.0)}
ref = {}; bad = {k: 0 for k in checks}; worst = {k: 0.0 for k in checks}
for it in range(int(sys.argv[1]) + 1):
    for name, (fn, tol) in checks.items():
        torch.cuda.synchronize()
        out = fn()
        torch.cuda.synchronize()
        if it == 0: ref[name] = out; continue
        rel = max(float((a.double() - b.double()).norm() / max(1e-30, float(b.double().norm()))) for a, b in zip(out, ref[name]))
        if rel > tol:
            bad[name] += 1; worst[name] = max(worst[name], rel)
for name in checks:
    print(f"{name}: wrong in {bad[name]} of {sys.argv[1]} (worst rel {worst[name]:.3g})", flush=True)
