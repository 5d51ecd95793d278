#!/usr/bin/env python3
"""GPU time of the regions of one training step of the bench model (HIP events at region
boundaries; backward boundaries via tensor hooks).  Dev tool."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from coda_neurips2023_amd.synthetic_scenes import make_batch  # noqa: E402

dev = torch.device("cuda:0")
model, step_fn, _, _ = bench.build_workload("model", dev)
opt = torch.optim.AdamW(model.parameters(), lr=1e-4, fused=True)
pc, mn, mx = make_batch(bench.B_PER_GPU, bench.N_POINTS, seed=1)
batch = {"point_clouds": torch.from_numpy(pc).to(dev), "point_cloud_dims_min": torch.from_numpy(mn).to(dev),
         "point_cloud_dims_max": torch.from_numpy(mx).to(dev)}

marks = []


def mark(name):
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    marks.append((name, e))


def fwd_hooks(mod, name):
    mod.register_forward_pre_hook(lambda m, a: mark(name + ":start"))

    def post(m, a, o):
        mark(name + ":end")
        t = o
        while isinstance(t, (tuple, list)):
            t = [x for x in t if torch.is_tensor(x) and x.requires_grad][-1] if any(
                torch.is_tensor(x) and x.requires_grad for x in t) else t[0]
        if torch.is_tensor(t) and t.requires_grad:
            t.register_hook(lambda g: mark("bwd_reached_output_of:" + name))

    mod.register_forward_hook(post)


for n in ["pre_encoder", "encoder", "encoder_to_decoder_projection", "decoder", "query_projection"]:
    fwd_hooks(getattr(model, n), n)
for i, layer in enumerate(model.decoder.layers):
    if i in (0, 7):
        fwd_hooks(layer, f"dec_layer{i}")
for i, layer in enumerate(model.encoder.layers):
    fwd_hooks(layer, f"enc_layer{i}")

acc = {}
order = []
for it in range(8):
    marks.clear()
    opt.zero_grad(set_to_none=True)
    mark("step:start")
    loss = step_fn(model, batch)
    mark("fwd:end")
    loss.backward()
    mark("bwd:end")
    opt.step()
    mark("opt:end")
    torch.cuda.synchronize()
    if it < 3:
        continue
    for (n0, e0), (n1, e1) in zip(marks[:-1], marks[1:]):
        k = f"{n0} -> {n1}"
        if k not in acc:
            acc[k] = 0.0
            order.append(k)
        acc[k] += e0.elapsed_time(e1) / 5
    acc["TOTAL"] = acc.get("TOTAL", 0.0) + marks[0][1].elapsed_time(marks[-1][1]) / 5
for k in order:
    print(f"{acc[k]:8.3f} ms  {k}")
print(f"{acc['TOTAL']:8.3f} ms  TOTAL")
