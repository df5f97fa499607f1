#!/usr/bin/env python3
"""Per-layer timing of the MFMA conv kernel on the VGG16 shapes of an SxS image (forward and
data-gradient), HIP events around back-to-back launches.  Usage: python tools/conv_bench.py [S] [iters]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "strotss-tensorflow_amd")):
    sys.path.insert(0, p)
import torch
from nn import _ops


def main():
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    only = sys.argv[3] if len(sys.argv) > 3 else None
    dev = "cuda"
    layers = [("c1_2", S, 64, 64), ("c2_1", S // 2, 64, 128), ("c2_2", S // 2, 128, 128), ("c3_1", S // 4, 128, 256),
              ("c3_2", S // 4, 256, 256), ("c4_1", S // 8, 256, 512), ("c4_2", S // 8, 512, 512),
              ("c5_1", S // 16, 512, 512)]
    g = torch.Generator().manual_seed(0)
    tot_t = tot_f = 0.0
    for name, hw, cin, cout in layers:
        if only and name != only:
            continue
        x = torch.relu(torch.randn(1, hw, hw, cin, generator=g)).to(dev)
        w = (torch.randn(9, cout, cin, generator=g) * 0.05).to(dev)
        wb = (torch.randn(9, cin, cout, generator=g) * 0.05).to(dev)
        b = torch.zeros(cout, device=dev)
        gy = torch.randn(1, hw, hw, cout, generator=g).to(dev)
        out = torch.empty(1, hw, hw, cout, device=dev)
        gin = torch.empty(1, hw, hw, cin, device=dev)
        flops = 2.0 * 9 * cin * cout * hw * hw
        P = 36 if os.environ.get("STROTSS_WINOGRAD_TILE", "2") == "4" else 16
        u = (torch.randn(P, cout, cin, generator=g) * 0.05).to(dev)
        ub = (torch.randn(P, cin, cout, generator=g) * 0.05).to(dev)
        for kind in ("fwd", "dgrad", "wfwd", "wdgrad"):
            if kind.endswith("dgrad") and cin % 64:
                continue
            fn = {"fwd": lambda: _ops.conv3x3_relu_fwd(x, w, b, out=out),
                  "dgrad": lambda: _ops.conv3x3_dgrad(gy, wb, cin, act_in=x, out=gin),
                  "wfwd": lambda: _ops.conv3x3_winograd_fwd(x, u, b, out=out),
                  "wdgrad": lambda: _ops.conv3x3_winograd_dgrad(gy, ub, cin, act_in=x, out=gin)}[kind]
            for _ in range(2):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(iters):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / iters
            if not kind.startswith("w"):
                tot_t += ms; tot_f += flops
            print(f"{name:5s} {kind:6s} hw={hw:5d} cin={cin:4d} cout={cout:4d}  {ms*1e3:9.1f} us  {flops/ms/1e9:7.1f} TFLOP/s")
    if tot_t:
        print(f"total {tot_t:.3f} ms  {tot_f/tot_t/1e9:.1f} TFLOP/s")


if __name__ == "__main__":
    main()
