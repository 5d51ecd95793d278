"""True host cost of one training step: the bench step on ONE small scene (same launch count, negligible GPU work),
so wall time per step = host time per step.  Usage: [CODA_STACK=python] python tools/host_time_small.py"""
import os
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
opt = torch.optim.AdamW(model.parameters(), lr=1e-4, fused=True)
pool = []
for i in range(4):
    pc, mn, mx = make_batch(1, 2048, seed=1 + i)
    pool.append({"point_clouds": torch.from_numpy(pc).to(dev), "point_cloud_dims_min": torch.from_numpy(mn).to(dev),
                 "point_cloud_dims_max": torch.from_numpy(mx).to(dev)})


def one(i):
    model.prefetch_sampling(pool[(i + 1) % 4], wait_for=None)
    opt.zero_grad(set_to_none=True)
    step_fn(model, pool[i % 4]).backward()
    opt.step()


for i in range(8):
    one(i)
torch.cuda.synchronize()
n = 40
t0 = time.perf_counter()
for i in range(n):
    one(i + 8)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"CODA_STACK={os.environ.get('CODA_STACK', 'c')}: host enqueue {1e3 * (t1 - t0) / n:.2f} ms/step, "
      f"wall {1e3 * (t2 - t0) / n:.2f} ms/step (1 scene, 2048 points: launch-bound)")
