#!/usr/bin/env python3
"""cProfile of the host side of the bench step (launch-bound analysis).  Dev tool."""
import cProfile
import os
import pstats
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from coda_neurips2023_amd.synthetic_scenes import make_batch  # noqa: E402

dev = torch.device("cuda:0")
model, step_fn, _, _ = bench.build_workload("model", dev)
opt = torch.optim.AdamW(model.parameters(), lr=1e-4, fused=True)
pool = []
for i in range(4):
    pc, mn, mx = make_batch(bench.B_PER_GPU, bench.N_POINTS, seed=1 + i)
    pool.append({"point_clouds": torch.from_numpy(pc).to(dev), "point_cloud_dims_min": torch.from_numpy(mn).to(dev),
                 "point_cloud_dims_max": torch.from_numpy(mx).to(dev)})


def one(i):
    model.prefetch_sampling(pool[(i + 1) % 4], wait_for=None)
    opt.zero_grad(set_to_none=True)
    step_fn(model, pool[i % 4]).backward()
    opt.step()


for i in range(6):
    one(i)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for i in range(10):
    one(i + 6)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("cumtime").print_stats("repo|optim", 40)
