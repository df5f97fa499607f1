#!/usr/bin/env python3
"""Per-kernel summary of two rocprofv3 --pmc passes (tools/pmc_sets.sh): set A (cycles, waits, MFMA busy, LDS) and set C
(wave-instruction counts).  One line per kernel name and grid, averaged over its launches.
Usage: python tools/pmc_summary.py DIR_A DIR_C"""
import csv
import glob
import re
import sys
from collections import defaultdict


def load(d):
    cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
    dur = {}
    for f in kt:
        for r in csv.DictReader(open(f)):
            dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    rows = defaultdict(lambda: defaultdict(float))
    meta = {}
    for f in cc:
        for r in csv.DictReader(open(f)):
            did = r["Dispatch_Id"]
            rows[did][r["Counter_Name"]] += float(r["Counter_Value"])
            meta[did] = (r["Kernel_Name"], int(r["Grid_Size"]))
    return rows, meta, dur


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return re.sub(r"\(.*$", "", n)[:78]


def group(rows, meta, dur):
    g = defaultdict(list)
    for did, c in rows.items():
        name, grid = meta[did]
        g[(short(name), grid)].append((c, dur.get(did, 0.0)))
    return g


def main():
    A = group(*load(sys.argv[1]))
    C = group(*load(sys.argv[2]))
    print("rocprofv3 --pmc <set> --kernel-trace -- python3 bench.py --no-cpu-baseline --no-e2e --no-pyramid --no-families --no-graph --steps 6")
    print("(the 1024-px optimisation step, eager launches; one line per kernel name and grid, mean over its launches; counters serialise the")
    print(" kernels, durations are those of the counter run)")
    print("set A: clk = GRBM_GUI_ACTIVE/8/duration; MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES/(GRBM_GUI_ACTIVE/8 * 1024 SIMDs); waits and active as")
    print("       fractions of SQ_WAVE_CYCLES; LDS busy = SQ_LDS_IDX_ACTIVE/256 CUs/cycles; conflict = SQ_LDS_BANK_CONFLICT/SQ_LDS_IDX_ACTIVE")
    print("set C: wave-instruction counts per launch (millions)\n")
    keys = sorted(A, key=lambda k: -sum(d for _, d in A[k]))
    for k in keys:
        ls = A[k]
        n = len(ls)
        m = defaultdict(float)
        for c, d in ls:
            for kk, v in c.items():
                m[kk] += v / n
        d = sum(x for _, x in ls) / n
        if d < 8.0 and n < 12:
            continue
        cyc = m["GRBM_GUI_ACTIVE"] / 8.0
        wc = max(m["SQ_WAVE_CYCLES"], 1.0)
        line = (f"A {k[0]:78s} grid={k[1]:9d} n={n:3d} dur_us={d:7.1f} clk={cyc / max(d, 1e-9) / 1e3:4.2f}GHz "
                f"MfmaUtil={100 * m['SQ_VALU_MFMA_BUSY_CYCLES'] / max(cyc * 1024, 1):5.1f}% WAIT_ANY={100 * m['SQ_WAIT_ANY'] / wc:5.1f}% "
                f"WAIT_INST_ANY={100 * m['SQ_WAIT_INST_ANY'] / wc:5.1f}% ACTIVE_INST_ANY={100 * m['SQ_ACTIVE_INST_ANY'] / wc:5.1f}% "
                f"LDS_busy={100 * m['SQ_LDS_IDX_ACTIVE'] / 256 / max(cyc, 1):5.1f}% "
                f"LDS_conflict/active={100 * m['SQ_LDS_BANK_CONFLICT'] / max(m['SQ_LDS_IDX_ACTIVE'], 1):5.1f}%")
        print(line)
        if k in C:
            lc = C[k]
            mc = defaultdict(float)
            for c, _ in lc:
                for kk, v in c.items():
                    mc[kk] += v / len(lc)
            print(f"C {'':78s} wave-instructions: VALU={mc['SQ_INSTS_VALU'] / 1e6:7.2f}M LDS={mc['SQ_INSTS_LDS'] / 1e6:6.2f}M "
                  f"VMEM_RD={mc['SQ_INSTS_VMEM_RD'] / 1e6:5.2f}M VMEM_WR={mc['SQ_INSTS_VMEM_WR'] / 1e6:5.2f}M SALU={mc['SQ_INSTS_SALU'] / 1e6:6.2f}M")


if __name__ == "__main__":
    main()
