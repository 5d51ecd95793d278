#!/usr/bin/env python3
"""Per-(kernel, grid) averages from a rocprofv3 kernel trace: the split-key attention kernels serve the decoder's
cross-attention (256 x 2048) and self-attention (256 x 256) launches with ONE instantiation, so the per-kernel
averages of `--stats` mix the two shapes.  Where the grid is the same too (it only depends on the query count), the
durations are bimodal (e.g. 14 us / 58 us): rows whose slowest launch takes more than 2.5 x the fastest are split in
two at the geometric mean of the extremes.

    python tools/trace_by_grid.py <..._kernel_trace.csv> [substring, default mha_] > by_grid.csv
"""
import collections
import csv
import re
import sys


def main():
    path = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else "mha_"
    acc = collections.defaultdict(list)
    with open(path, newline="") as f:
        rd = csv.DictReader(f)
        cols = rd.fieldnames
        gx = next(c for c in cols if c.lower().replace("_", "") in ("gridsizex", "gridx", "gridsize"))
        gy = next((c for c in cols if c.lower().replace("_", "") in ("gridsizey", "gridy")), None)
        name = next(c for c in cols if c.lower().replace("_", "") == "kernelname")
        t0 = next(c for c in cols if c.lower().replace("_", "") == "starttimestamp")
        t1 = next(c for c in cols if c.lower().replace("_", "") == "endtimestamp")
        for r in rd:
            nm = r[name]
            if want not in nm:
                continue
            nm = re.sub(r"^void ", "", nm)
            nm = re.sub(r"coda::\(anonymous namespace\)::", "", nm)
            nm = nm.split("(")[0]
            key = (nm, r[gx], r[gy] if gy else "")
            dur = (int(r[t1]) - int(r[t0])) / 1e3
            acc[key].append(dur)
    w = csv.writer(sys.stdout)
    w.writerow(["kernel", "grid_x", "grid_y", "mode", "calls", "avg_us", "min_us", "max_us"])
    rows = []
    for (nm, x, y), ds in acc.items():
        lo, hi = min(ds), max(ds)
        groups = [("all", ds)]
        if hi > 2.5 * lo:
            cut = (lo * hi) ** 0.5
            groups = [("short", [d for d in ds if d < cut]), ("long", [d for d in ds if d >= cut])]
        for mode, g in groups:
            if g:
                rows.append((sum(g), [nm, x, y, mode, len(g), f"{sum(g) / len(g):.1f}", f"{min(g):.1f}", f"{max(g):.1f}"]))
    for _, r in sorted(rows, key=lambda t: -t[0]):
        w.writerow(r)


if __name__ == "__main__":
    main()
