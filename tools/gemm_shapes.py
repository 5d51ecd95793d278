#!/usr/bin/env python3
"""Which GEMM problems does one training step issue through gemm.py / torch?  Dev tool."""
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from coda_neurips2023_amd import gemm  # noqa: E402
from coda_neurips2023_amd.synthetic_scenes import make_batch  # noqa: E402

dev = torch.device("cuda:0")
model, step_fn, _, _ = bench.build_workload("model", dev)
pc, mn, mx = make_batch(bench.B_PER_GPU, bench.N_POINTS, seed=1)
batch = {"point_clouds": torch.from_numpy(pc).to(dev), "point_cloud_dims_min": torch.from_numpy(mn).to(dev),
         "point_cloud_dims_max": torch.from_numpy(mx).to(dev)}
for _ in range(2):
    model.zero_grad(set_to_none=True)
    step_fn(model, batch).backward()
counts = collections.Counter()
real = gemm._run


def spy(transa, transb, m, n, k, a, b, out, bias, accumulate):
    counts[("coda_gemm", transa, transb, m, n, k, bias is not None, bool(accumulate), a.stride(0), b.stride(0),
            None if out is None else out.stride(0))] += 1
    return real(transa, transb, m, n, k, a, b, out, bias, accumulate)


gemm._run = spy
for name in ("mm", "bmm", "addmm", "baddbmm", "matmul"):
    orig = getattr(torch, name)

    def wrap(*a, _o=orig, _n=name, **kw):
        counts[("torch." + _n,) + tuple(tuple(t.shape) for t in a if torch.is_tensor(t))] += 1
        return _o(*a, **kw)
    setattr(torch, name, wrap)
model.zero_grad(set_to_none=True)
step_fn(model, batch).backward()
torch.cuda.synchronize()
for key, c in sorted(counts.items(), key=lambda kv: -kv[1]):
    print(c, key)
