#!/usr/bin/env python3
"""Host time of one training step split by autograd node: perf_counter around the forward and the backward of every
custom autograd.Function of the package (the backward runs on the engine's thread, where cProfile does not look), on
one small scene (launch count of the bench step, negligible GPU work).  The rest of run_backward's wall time is the
engine itself and torch-native nodes.  Dev tool."""
import collections
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from coda_neurips2023_amd.synthetic_scenes import make_batch  # noqa: E402
from coda_neurips2023_amd import (align_loss, attention_core, box_decode, box_loss, fused_blocks, fused_bn_mlp,  # noqa: E402
                                  fused_layers, linear_fn)
from coda_neurips2023_amd.pointnet2 import fused_sa_mlp, pointnet2_utils  # noqa: E402

acc = collections.defaultdict(float)
cnt = collections.Counter()


def wrap(cls):
    for name in ("forward", "backward"):
        fn = getattr(cls, name)
        raw = fn.__func__ if hasattr(fn, "__func__") else fn

        def timed(*a, _raw=raw, _key=f"{cls.__name__}.{name}", **k):
            t = time.perf_counter()
            try:
                return _raw(*a, **k)
            finally:
                acc[_key] += time.perf_counter() - t
                cnt[_key] += 1
        setattr(cls, name, staticmethod(timed))


for mod in (align_loss, attention_core, box_decode, box_loss, fused_blocks, fused_bn_mlp, fused_layers, linear_fn, fused_sa_mlp,
            pointnet2_utils):
    for v in list(vars(mod).values()):
        if isinstance(v, type) and issubclass(v, torch.autograd.Function) and v is not torch.autograd.Function \
                and v.__module__ == mod.__name__:
            wrap(v)

bench.B_PER_GPU = 1
dev = torch.device("cuda:0")
model, step_fn, _, _ = bench.build_workload("model", dev)
opt, clip = bench.make_optimizer(model.parameters())
pool = []
for i in range(4):
    pc, mn, mx = make_batch(1, 2048, seed=1 + i)
    pool.append({"point_clouds": torch.from_numpy(pc).to(dev), "point_cloud_dims_min": torch.from_numpy(mn).to(dev),
                 "point_cloud_dims_max": torch.from_numpy(mx).to(dev)})
seg = collections.defaultdict(float)


def one(i):
    t0 = time.perf_counter()
    model.prefetch_sampling(pool[(i + 1) % 4], wait_for=None)
    opt.zero_grad(set_to_none=True)
    t1 = time.perf_counter()
    loss = step_fn(model, pool[i % 4])
    t2 = time.perf_counter()
    loss.backward()
    t3 = time.perf_counter()
    clip()
    opt.step()
    t4 = time.perf_counter()
    seg["prefetch+zero_grad"] += t1 - t0
    seg["forward+criterion"] += t2 - t1
    seg["backward"] += t3 - t2
    seg["clip+optimizer"] += t4 - t3


for i in range(8):
    one(i)
torch.cuda.synchronize()
acc.clear(); cnt.clear(); seg.clear()
N = 20
for i in range(N):
    one(i)
torch.cuda.synchronize()
print("segments, ms per step:", {k: round(1e3 * v / N, 3) for k, v in seg.items()}, "total", round(1e3 * sum(seg.values()) / N, 3))
fw = sum(v for k, v in acc.items() if k.endswith(".forward"))
bw = sum(v for k, v in acc.items() if k.endswith(".backward"))
print(f"inside custom Functions: forward {1e3 * fw / N:.3f} ms, backward {1e3 * bw / N:.3f} ms per step")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print(f"  {k:32s} {cnt[k] / N:6.1f} calls/step  {1e3 * v / N:7.3f} ms/step  {1e6 * v / cnt[k]:7.1f} us/call")
