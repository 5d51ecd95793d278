#!/usr/bin/env python3
"""Ordered kernel sequence of the last step in a rocprofv3 kernel trace CSV.
    python tools/trace_seq.py <kernel_trace.csv> [start_frac end_frac]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# step boundary: FPS kernel with the large template arg marks the start of a step
starts = [i for i, r in enumerate(rows) if ("fps_t512_kernel<4" in r["Kernel_Name"])]
a, b = starts[-2], starts[-1]
step = rows[a:b]
t0 = int(step[0]["Start_Timestamp"])


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"at::native::", "", n)
    n = re.sub(r"void ", "", n)
    if n.startswith("Cijk"):
        m = re.search(r"MT(\d+x\d+x\d+)", n)
        return "GEMM " + n[:14] + " MT" + (m.group(1) if m else "")
    return n[:110]


lo = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
hi = float(sys.argv[3]) if len(sys.argv) > 3 else 1e9
prev_end = t0
busy = 0
for r in step:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    ms = (s - t0) / 1e6
    busy += e - s
    if lo <= ms <= hi:
        print(f"{ms:8.3f} ms  gap {max(0, s - prev_end) / 1e3:6.1f} us  dur {(e - s) / 1e3:7.1f} us  {short(r['Kernel_Name'])}")
    prev_end = max(prev_end, e)
print(f"step: {len(step)} kernels, span {(prev_end - t0) / 1e6:.3f} ms, busy {busy / 1e6:.3f} ms")
