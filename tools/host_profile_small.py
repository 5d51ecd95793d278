#!/usr/bin/env python3
"""cProfile of the host side of one training step on ONE small scene (same launch count as the bench step, negligible
GPU work: what the profile shows is host time).  Dev tool.  Usage: python tools/host_profile_small.py [tottime|cumtime]"""
import cProfile
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from coda_neurips2023_amd.synthetic_scenes import make_batch  # noqa: E402

bench.B_PER_GPU = 1
dev = torch.device("cuda:0")
model, step_fn, _, _ = bench.build_workload("model", dev)
opt, clip = bench.make_optimizer(model.parameters())
pool = []
for i in range(4):
    pc, mn, mx = make_batch(1, 2048, seed=1 + i)
    pool.append({"point_clouds": torch.from_numpy(pc).to(dev), "point_cloud_dims_min": torch.from_numpy(mn).to(dev),
                 "point_cloud_dims_max": torch.from_numpy(mx).to(dev)})


def one(i):
    model.prefetch_sampling(pool[(i + 1) % 4], wait_for=None)
    opt.zero_grad(set_to_none=True)
    step_fn(model, pool[i % 4]).backward()
    clip()
    opt.step()


for i in range(8):
    one(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(20):
    one(i)
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"host enqueue {1e3 * (t1 - t0) / 20:.2f} ms/step (no profiler)")
pr = cProfile.Profile()
pr.enable()
for i in range(20):
    one(i)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats(sys.argv[1] if len(sys.argv) > 1 else "tottime").print_stats(45)
