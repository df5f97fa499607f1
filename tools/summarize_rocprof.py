"""Condense rocprofv3 output directories into the small CSVs kept under profiles/.

    python tools/summarize_rocprof.py trace  <dir from --kernel-trace --stats>  profiles/rNN
        -> profiles/rNN_bench_kernel_stats.csv   (rocprofv3's own per-kernel stats, copied)
           profiles/rNN_mfma_kernels_by_grid.csv (MFMA kernels grouped by kernel x grid: median/min/max us)
    python tools/summarize_rocprof.py pmc    <FETCH_SIZE dir> <WRITE_SIZE dir>  profiles/rNN
        -> profiles/rNN_hbm_traffic_by_kernel.csv (per launch, with the gfx950 FETCH_SIZE x2 correction of
           /opt/skills/guides/MI355X_MICROARCH.md)
"""
import csv
import glob
import re
import shutil
import statistics
import sys
from collections import defaultdict

MFMA = ("conv3x3_mfma", "gemm_kc_pipe", "gemm_kernel", "gemm_x3", "conv3x3_c3_fwd_mfma", "winograd43_fused")


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*\)$", "", name)


def one(pattern: str) -> str:
    hits = sorted(glob.glob(pattern, recursive=True))
    if not hits:
        raise SystemExit(f"no file matches {pattern}")
    return hits[0]


def trace(src: str, out: str) -> None:
    shutil.copy(one(f"{src}/**/*_kernel_stats.csv"), f"{out}_bench_kernel_stats.csv")
    groups = defaultdict(list)
    with open(one(f"{src}/**/*_kernel_trace.csv")) as f:
        for r in csv.DictReader(f):
            k = short(r["Kernel_Name"])
            if not k.startswith(MFMA):
                continue
            key = (k, int(r["Grid_Size_X"]), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]), int(r["VGPR_Count"]),
                   int(r["LDS_Block_Size"]))
            groups[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    with open(f"{out}_mfma_kernels_by_grid.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow("kernel grid_x_threads grid_y grid_z vgpr lds_bytes calls median_us min_us max_us".split())
        for key in sorted(groups):
            d = groups[key]
            w.writerow([*key, len(d), round(statistics.median(d), 1), round(min(d), 1), round(max(d), 1)])


def pmc(fetch_dir: str, write_dir: str, out: str) -> None:
    def collect(src, counter):
        acc = defaultdict(list)
        with open(one(f"{src}/**/*_counter_collection.csv")) as f:
            for r in csv.DictReader(f):
                if r["Counter_Name"] != counter:
                    continue
                acc[(short(r["Kernel_Name"]), int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
        return acc
    rd, wr = collect(fetch_dir, "FETCH_SIZE"), collect(write_dir, "WRITE_SIZE")
    with open(f"{out}_hbm_traffic_by_kernel.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "grid_threads", "launches",
                    "read_MB_per_launch(FETCH_SIZE*1024*2: gfx950 half-count correction)",
                    "write_MB_per_launch(WRITE_SIZE*1024)"])
        for key in sorted(rd, key=lambda k: -sum(rd[k])):
            r = statistics.mean(rd[key]) * 1024 * 2 / 1e6
            wv = statistics.mean(wr[key]) * 1024 / 1e6 if key in wr else float("nan")
            w.writerow([key[0], key[1], len(rd[key]), round(r, 1), round(wv, 1)])


if __name__ == "__main__":
    if len(sys.argv) >= 4 and sys.argv[1] == "trace":
        trace(sys.argv[2], sys.argv[3])
    elif len(sys.argv) >= 5 and sys.argv[1] == "pmc":
        pmc(sys.argv[2], sys.argv[3], sys.argv[4])
    else:
        raise SystemExit(__doc__)
