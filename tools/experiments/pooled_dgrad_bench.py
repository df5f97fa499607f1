#!/usr/bin/env python3
"""dgrad(maxpool2_bwd(gpool, code)): the pooling-backward launch + the data-gradient against the one launch that expands
the pooled gradient in its patch loader.  usage: pooled_dgrad_bench.py [S]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "strotss-tensorflow_amd")):
    sys.path.insert(0, p)
import torch
from nn import _ops

S = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
g = torch.Generator().manual_seed(0)
for name, hw, cin, cout in (("c1_2", S, 64, 64), ("c2_2", S // 2, 128, 128), ("c3_3", S // 4, 256, 256)):
    act = torch.relu(torch.randn(1, hw, hw, cout, generator=g)).cuda()          # the layer's output
    x = torch.relu(torch.randn(1, hw, hw, cin, generator=g)).cuda()             # its input (ReLU mask)
    ub = (torch.randn(36, cin, cout, generator=g) * 0.05).cuda()
    code = torch.empty(1, hw // 2, hw // 2, cout, dtype=torch.uint8, device="cuda")
    _ops.maxpool2_fwd(act, code=code)
    gpool = torch.randn(1, hw // 2, hw // 2, cout, generator=g).cuda()
    bits = _ops.relu_bits(x)
    gfull = torch.empty_like(act); o1 = torch.empty_like(x); o2 = torch.empty_like(x)
    ok = _ops.conv3x3_winograd_dgrad_pooled_ok(hw, hw, cout, cin, ub)
    def two():
        _ops.maxpool2_bwd(act, gpool, out=gfull, code=code)
        _ops.conv3x3_winograd_dgrad(gfull, ub, cin, out=o1, relu_bits=bits)
    def one():
        _ops.conv3x3_winograd_dgrad_pooled(gpool, code, hw, hw, ub, cin, out=o2, relu_bits=bits)
    res = []
    for fn in (two, one) if ok else (two,):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(20):
            fn()
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 20 * 1e3)
    same = bool(torch.equal(o1, o2)) if ok else None
    print(f"{name} hw={hw} {cout}->{cin}: pool_bwd + dgrad {res[0]:.1f} us" + (f", one launch {res[1]:.1f} us, equal={same}" if ok else " (no fused route)"))
