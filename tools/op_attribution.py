"""Which Python lines launch the small PyTorch kernels of the training step?  Runs a few headline steps under
torch.profiler with stacks and prints device time / launch count per (operator, first frame inside this package).

    python tools/op_attribution.py [--steps 3] [--top 60]
"""
import argparse
import collections
import os
import sys

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--top", type=int, default=60)
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
        step_fn(mod, pool[i % 2]).backward()
        clip()
        opt.step()

    for i in range(4):
        one(i)
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        for i in range(a.steps):
            one(i)
        torch.cuda.synchronize()
    out_dir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    prof.export_chrome_trace(os.path.join(out_dir, "step_trace.json.gz"))
    agg = collections.defaultdict(lambda: [0.0, 0])
    for ev in prof.key_averages(group_by_stack_n=12):
        dt = getattr(ev, "self_device_time_total", None)
        if dt is None:
            dt = ev.self_cuda_time_total
        if dt <= 0:
            continue
        frame = "?"
        for f in ev.stack:
            if "coda_neurips2023_amd" in f or "bench.py" in f:
                frame = f.split("coda_neurips2023_amd/")[-1] if "coda_neurips2023_amd/" in f else f.split("/")[-1]
                break
        key = (ev.key, frame)
        agg[key][0] += dt
        agg[key][1] += ev.count
    rows = sorted(agg.items(), key=lambda kv: -kv[1][0])
    tot = sum(v[0] for v in agg.values())
    print(f"device time in profiled ops: {tot / a.steps / 1e3:.3f} ms/step")
    for (op, frame), (dt, cnt) in rows[:a.top]:
        if op.startswith("aten::") or "Backward" in op or op.startswith("Optimizer") or True:
            print(f"{dt / a.steps:9.1f} us/step  n/step {cnt / a.steps:6.1f}  {op[:48]:48s} {frame[:90]}")


if __name__ == "__main__":
    main()
