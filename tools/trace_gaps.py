#!/usr/bin/env python3
"""Idle time of the compute queue inside a training step, from a rocprofv3 kernel trace: per step (from one
assignment kernel to the next) the span, the summed kernel time on the busiest queue, and the gaps between consecutive
kernels -- how much of a GPU-bound step is dispatch latency between launch-sized kernels.

    python tools/trace_gaps.py <..._kernel_trace.csv>
"""
import collections
import csv
import sys


def main():
    rows = []
    with open(sys.argv[1], newline="") as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), (r["Queue_Id"], r.get("Stream_Id", "")),
                         r["Kernel_Name"]))
    rows.sort()
    byq = collections.Counter(r[2] for r in rows)
    mq = byq.most_common(1)[0][0]
    main_q = [r for r in rows if r[2] == mq]
    marks = [i for i, r in enumerate(main_q) if "hungarian" in r[3]]
    print(f"{len(rows)} kernels, queues {dict(byq)}, steps seen {len(marks)}")
    for a, b in list(zip(marks[:-1], marks[1:]))[2:8]:
        seg = main_q[a:b]
        span = seg[-1][1] - seg[0][0]
        busy = sum(r[1] - r[0] for r in seg)
        gaps = [(seg[i + 1][0] - seg[i][1], seg[i][3], seg[i + 1][3]) for i in range(len(seg) - 1)]
        pos = [g for g in gaps if g[0] > 0]
        small = sorted(g[0] for g in pos if g[0] < 20000)
        big = sorted((g for g in pos if g[0] >= 20000), reverse=True)[:6]
        overl = -sum(g[0] for g in gaps if g[0] < 0)
        print(f"step: span {span / 1e6:.2f} ms, kernels {busy / 1e6:.2f} ms in {len(seg)} launches, idle {sum(g[0] for g in pos) / 1e6:.2f} ms "
              f"(overlap {overl / 1e6:.2f}); gaps < 20 us: {len(small)} totalling {sum(small) / 1e6:.2f} ms, median "
              f"{small[len(small) // 2] / 1e3:.2f} us, p90 {small[int(len(small) * 0.9)] / 1e3:.2f} us")
        for g, a_, b_ in big:
            print(f"    {g / 1e3:8.1f} us between {a_[:60]} -> {b_[:60]}")
        # where the small gaps are: by the kernel that follows
        acc = collections.Counter()
        for g, a_, b_ in pos:
            if g < 20000:
                acc[b_.split("(")[0][-50:]] += g
        print("    small-gap time by following kernel:", [(k, round(v / 1e3)) for k, v in acc.most_common(8)])


if __name__ == "__main__":
    main()
