#!/usr/bin/env python3
"""Per-kernel medians of a rocprofv3 --pmc counter_collection CSV: `python tools/pmc_medians.py file.csv [name filter]`."""
import csv
import statistics
import sys
from collections import defaultdict

vals = defaultdict(list)
durs = defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"]
    if len(sys.argv) > 2 and sys.argv[2] not in name:
        continue
    short = name.replace("void ", "").replace("coda::(anonymous namespace)::", "").split("(")[0]
    if "Grid_Size" in r:     # one instantiation serves several shapes: keep them apart by their grid
        short += f" grid {r['Grid_Size']}"
    vals[(short, r["Counter_Name"])].append(float(r["Counter_Value"]))
    durs[(short, r["Counter_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for (k, c), v in sorted(vals.items()):
    print(f"{k:76s} {c:12s} n={len(v):4d} median {statistics.median(v):12.1f}  (dur median {statistics.median(durs[(k, c)]):8.1f} us)")
