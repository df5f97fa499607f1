#!/usr/bin/env python3
"""One optimisation step as the GPU saw it: the ordered kernel list of a steady-state hipGraph replay with every kernel's
duration and the idle gap in front of it, from a `rocprofv3 --kernel-trace` CSV of `bench.py --scale S --no-families
--no-cpu-baseline --no-pyramid --no-e2e`.  usage: step_trace.py TRACE_DIR [marker-kernel-substring]"""
import csv, glob, os, sys
d = sys.argv[1]
f = next(iter(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)))
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
marker = sys.argv[2] if len(sys.argv) > 2 else "rmsprop_kernel"
ends = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
assert len(ends) > 6, "too few steps in the trace"
a, b = ends[-3] + 1, ends[-2] + 1                      # one step: after an RMSprop launch up to and including the next
step = rows[a:b]
t0 = int(rows[a - 1]["End_Timestamp"])
busy = 0; prev_end = t0
print(f"{len(step)} launches in the step")
for r in step:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    name = name.split("(")[0] if not name.startswith("_Z") else name
    print(f"{(s - prev_end) / 1e3:7.2f} gap {(e - s) / 1e3:8.2f} us  {name[:90]}")
    busy += e - s; prev_end = e
total = prev_end - t0
print(f"step {total / 1e3:.1f} us: kernels {busy / 1e3:.1f} us, gaps {(total - busy) / 1e3:.1f} us")
