"""Which GEMM of one training step takes which route (x3 kernels / library or own fp32 kernels via gemm._run), and which
large products bypass gemm.py altogether (torch.mm / bmm / addmm seen by a TorchDispatchMode)."""
import os
import sys

import torch
from torch.utils._python_dispatch import TorchDispatchMode

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from coda_neurips2023_amd import gemm  # noqa: E402

dev = torch.device("cuda:0")
model, step_fn, _, _ = bench.build_workload("model", dev)
opt, clip = bench.make_optimizer(model.parameters())
from coda_neurips2023_amd.synthetic_scenes import make_batch  # noqa: E402
pc, mn, mx = make_batch(bench.B_PER_GPU, bench.N_POINTS, seed=1)
batch = {"point_clouds": torch.from_numpy(pc).to(dev), "point_cloud_dims_min": torch.from_numpy(mn).to(dev),
         "point_cloud_dims_max": torch.from_numpy(mx).to(dev)}


def one():
    opt.zero_grad(set_to_none=True)
    step_fn(model, batch).backward()
    clip()
    opt.step()


for _ in range(2):
    one()
torch.cuda.synchronize()


class Seen(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.calls = {}

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.overloadpacket.__name__
        if name in ("mm", "bmm", "addmm", "baddbmm", "linear", "matmul"):
            shapes = tuple(tuple(a.shape) for a in args if torch.is_tensor(a))
            key = (name, shapes)
            self.calls[key] = self.calls.get(key, 0) + 1
        return func(*args, **(kwargs or {}))


gemm.route_log = {}
before = gemm.x3_calls
with Seen() as seen:
    one()
torch.cuda.synchronize()
print(f"x3 products in the step: {gemm.x3_calls - before}; cached weight-piece sets: {len(gemm._planes)}")
print("-- through gemm._run (route, transa, transb, m, n, k): calls, GFLOP")
for key, cnt in sorted(gemm.route_log.items(), key=lambda kv: -kv[0][3] * kv[0][4] * kv[0][5] * kv[1]):
    print(f"  {key}: {cnt}  {2e-9 * key[3] * key[4] * key[5] * cnt:7.2f}")
print("-- torch GEMM ops seen by the dispatcher (not through gemm.py): calls")
for (name, shapes), cnt in sorted(seen.calls.items(), key=lambda kv: -kv[1]):
    print(f"  {name} {shapes}: {cnt}")
