#!/usr/bin/env python3
"""dev: the fused ball query + grouping operator at B = 8 / 64, (a) one call at a time with a synchronisation in
between (what bench.py's `avg_launch_ms_alone` measures: launch latency and an idle GPU's clocks included) and (b) 200
calls back to back (per-call device time at full clocks).  CODA_BQ_QUERY=wave: the one-wave-per-centre query."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coda_neurips2023_amd.pointnet2 import _ext  # noqa: E402
from coda_neurips2023_amd.synthetic_scenes import make_batch  # noqa: E402

dev = torch.device("cuda:0")


def synced(fn, reps=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e))
    return float(np.median(ts)) * 1e3


def burst(fn, reps=200):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


for B, n in ((8, 20000), (64, 20000), (8, 40000)):
    pcs = [make_batch(8, n, seed=2000 + i)[0] for i in range(B // 8)]
    xyz = torch.from_numpy(np.concatenate(pcs)).to(dev)
    inds = _ext.furthest_point_sampling(xyz, 2048)
    new = torch.gather(xyz, 1, inds.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    f = lambda: _ext.query_and_group_xyz(new, xyz, 0.2, 64, True, channels_last=True)  # noqa: E731
    print(f"B={B} n={n} query={os.environ.get('CODA_BQ_QUERY', 'lanes8')}: "
          f"synced {synced(f):.1f} us   back-to-back {burst(f):.1f} us", flush=True)
