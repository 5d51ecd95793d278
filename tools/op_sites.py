#!/usr/bin/env python3
"""Which lines of this package issue the PyTorch-native ops of one training step?  A TorchDispatchMode logs every aten
op with the innermost frame inside the package (ops run by the autograd engine's own thread for torch-native backward
nodes have no Python frame: "<autograd engine>").  Dev tool; torch.profiler's stacks are empty on this build.

    python tools/op_sites.py [--top 60]
"""
import argparse
import collections
import os
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

SKIP = ("aten.view", "aten._unsafe_view", "aten.t.", "aten.transpose", "aten.permute", "aten.expand", "aten.slice",
        "aten.select", "aten.unsqueeze", "aten.squeeze", "aten.detach", "aten.alias", "aten.as_strided", "aten.reshape",
        "aten.unbind", "aten.split", "aten.empty", "aten.unflatten", "aten.lift_fresh", "aten.is_", "aten.stride",
        "aten.sym_", "aten.size", "aten._local_scalar_dense", "aten.narrow", "aten.chunk", "aten.unfold", "aten.diagonal")


class Log(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.sites = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not name.startswith(SKIP):
            site = "<autograd engine>"
            for fr in reversed(traceback.extract_stack(limit=40)):
                fn = fr.filename
                if (("coda_neurips2023_amd" in fn or fn.endswith("bench.py")) and "tools/" not in fn):
                    site = f"{os.path.relpath(fn, ROOT)}:{fr.lineno}"
                    break
            numel = 0
            for a in args:
                if torch.is_tensor(a):
                    numel = max(numel, a.numel())
            self.sites[(name, site, "small" if numel < (1 << 16) else "large")] += 1
        return func(*args, **(kwargs or {}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--top", type=int, default=80)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    mod, step_fn, desc, _ = bench.build_model_workload(dev)
    pool = []
    for i in range(2):
        pc, mn, mx = bench.make_batch(bench.B_PER_GPU, bench.N_POINTS, seed=4321 + i)
        pool.append({"point_clouds": torch.from_numpy(pc).to(dev), "point_cloud_dims_min": torch.from_numpy(mn).to(dev),
                     "point_cloud_dims_max": torch.from_numpy(mx).to(dev)})
    opt, clip = bench.make_optimizer(mod.parameters())

    def one(i):
        opt.zero_grad(set_to_none=True)
        if hasattr(mod, "prefetch_sampling"):
            mod.prefetch_sampling(pool[(i + 1) % 2])
        step_fn(mod, pool[i % 2]).backward()
        clip()
        opt.step()

    for i in range(3):
        one(i)
    torch.cuda.synchronize()
    log = Log()
    with log:
        one(3)
    torch.cuda.synchronize()
    total = sum(log.sites.values())
    print(f"{total} device ops in the step (views and metadata ops not counted)")
    for (name, site, size), n in log.sites.most_common(a.top):
        print(f"{n:5d}  {name:42s} {size:6s} {site}")


if __name__ == "__main__":
    main()
