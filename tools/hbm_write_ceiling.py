#!/usr/bin/env python3
"""What do pure streaming writes / reads / copies of first-layer size (268 MB) reach on this GPU?  The yardstick for the
HBM-bound families of bench.py (the first layer's forward pass WRITES 268 MB, its data-gradient READS 268 MB)."""
import torch
dev = torch.device("cuda", 0)
n = 1024 * 1024 * 64
a = torch.empty(n, dtype=torch.float32, device=dev)
b = torch.empty(n, dtype=torch.float32, device=dev)
def timed(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(reps):
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best
mb = n * 4 / 1e6
t = timed(lambda: a.fill_(1.0)); print(f"fill  {mb:.0f} MB written: {t * 1e3:7.1f} us  {mb / t / 1e3:.2f} TB/s")
t = timed(lambda: a.sum());      print(f"sum   {mb:.0f} MB read:    {t * 1e3:7.1f} us  {mb / t / 1e3:.2f} TB/s")
t = timed(lambda: b.copy_(a));   print(f"copy  {mb:.0f} MB + {mb:.0f} MB: {t * 1e3:7.1f} us  {2 * mb / t / 1e3:.2f} TB/s")
