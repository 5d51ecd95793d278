#!/usr/bin/env python3
"""Dev: where a step's wall time goes, from a rocprofv3 kernel trace (CSV) of `CODA_BENCH_LEGS=headline bench.py`.

    python tools/trace_gaps.py <run_kernel_trace.csv> [tail fraction, default 0.5]

Looks at the last part of the trace (steady state), per queue/stream: busy time, and for the union over all queues
the idle time (no kernel running anywhere); then the gap histogram between consecutive kernels on the busiest queue
and the kernels with the largest summed duration there.
"""
import csv
import sys
from collections import defaultdict


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
    ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Stream_Id", r.get("Queue_Id", "0")), r["Kernel_Name"])
                 for r in rows), key=lambda e: e[0])
    t0, t1 = ev[0][0], max(e[1] for e in ev)
    cut = t1 - (t1 - t0) * frac
    ev = [e for e in ev if e[0] >= cut]
    span = (max(e[1] for e in ev) - ev[0][0]) / 1e6
    print(f"window {span:.2f} ms, {len(ev)} kernels")
    by_q = defaultdict(list)
    for e in ev:
        by_q[e[2]].append(e)
    for q, es in sorted(by_q.items(), key=lambda kv: -sum(e[1] - e[0] for e in kv[1])):
        print(f"  queue {q}: {len(es)} kernels, busy {sum(e[1] - e[0] for e in es) / 1e6:.2f} ms ({100 * sum(e[1] - e[0] for e in es) / 1e6 / span:.1f} %)")
    # union busy
    busy, cur_s, cur_e = 0, None, None
    for s, e, _, _ in ev:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    print(f"union busy {busy / 1e6:.2f} ms = {100 * busy / 1e6 / span:.1f} % of the window; idle {span - busy / 1e6:.2f} ms")
    main_q = max(by_q, key=lambda q: sum(e[1] - e[0] for e in by_q[q]))
    es = by_q[main_q]
    gaps = [(es[i + 1][0] - es[i][1]) / 1e3 for i in range(len(es) - 1)]
    pos = [g for g in gaps if g > 0]
    print(f"busiest queue {main_q}: {len(pos)} positive gaps, sum {sum(pos) / 1e3:.2f} ms, median {sorted(pos)[len(pos) // 2]:.2f} us")
    for lo, hi in ((0, 2), (2, 5), (5, 10), (10, 20), (20, 50), (50, 200), (200, 1e9)):
        sel = [g for g in pos if lo <= g < hi]
        print(f"    gaps {lo:>4}-{hi:<6g} us: {len(sel):5d}, sum {sum(sel) / 1e3:.2f} ms")
    # which kernels FOLLOW the big gaps
    after = defaultdict(lambda: [0, 0.0])
    for i, g in enumerate(gaps):
        if g >= 10:
            k = es[i + 1][3].replace("void ", "").replace("coda::(anonymous namespace)::", "").split("(")[0][:70]
            after[k][0] += 1
            after[k][1] += g
    print("  kernels after gaps >= 10 us (count, summed gap ms):")
    for k, (c, g) in sorted(after.items(), key=lambda kv: -kv[1][1])[:25]:
        print(f"    {c:5d} {g / 1e3:7.2f}  {k}")


if __name__ == "__main__":
    main()
