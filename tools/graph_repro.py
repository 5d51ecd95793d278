#!/usr/bin/env python3
"""Dev: bisect the two open problems of step_graph.GraphedTail (round 1: end-capture crash after an eager
autograd step; one wrong loss on a later replay).  Each variant runs in its own process (a crash must not take
the others down) and prints one line; the parent prints the matrix.

    python tools/graph_repro.py            # all variants
    python tools/graph_repro.py <variant>  # one variant, in this process

Variant = <what>:<eager step before the capture 0|1>
  mgc      the same toy tail through torch.cuda.make_graphed_callables (PyTorch's own fwd/bwd capture)
  torch    pure-PyTorch toy tail (Linear / LayerNorm / ReLU / BatchNorm1d), no kernel of this repository
  modules  tiny detector, module-by-module layer path (CODA_LAYERS=modules): torch ops + the attention core
  layers   tiny detector, fused kernels as separate autograd nodes (CODA_LAYER_NODES=ops, CODA_DECODER_NODE=layers)
  fused    tiny detector, default fused path (one node per encoder layer, one for the decoder)
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WHAT = ["mgc", "torch", "modules", "layers", "fused"]
ENV = {"mgc": {}, "torch": {}, "modules": {"CODA_LAYERS": "modules"},
       "layers": {"CODA_LAYER_NODES": "ops", "CODA_DECODER_NODE": "layers"}, "fused": {}}


def run_variant(what, eager_first):
    import numpy as np
    import torch
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from coda_neurips2023_amd.step_graph import GraphedTail
    dev = torch.device("cuda:0")
    torch.manual_seed(0)

    if what == "mgc":
        model = torch.nn.Sequential(torch.nn.Linear(64, 128), torch.nn.LayerNorm(128), torch.nn.ReLU(),
                                    torch.nn.Linear(128, 8)).to(dev).train()
        xs = [torch.randn(512, 64, device=dev) for _ in range(3)]
        if eager_first:
            model(xs[0]).square().mean().backward()
            model.zero_grad(set_to_none=True)
        ref = []
        for x in xs:
            model.zero_grad(set_to_none=True)
            loss = model(x).square().mean()
            loss.backward()
            ref.append((float(loss), [p.grad.clone() for p in model.parameters()]))
        graphed = torch.cuda.make_graphed_callables(model, (xs[0].clone(),))
        worst = 0.0
        for i in (1, 2, 0, 0, 2, 1):
            model.zero_grad(set_to_none=True)
            loss = graphed(xs[i]).square().mean()
            loss.backward()
            worst = max(worst, abs(float(loss) - ref[i][0]) / abs(ref[i][0]))
            for p, r in zip(model.parameters(), ref[i][1]):
                worst = max(worst, float((p.grad - r).abs().max() / (r.abs().max() + 1e-12)))
        print(f"RESULT {what}:{int(eager_first)} worst relative difference eager vs replay = {worst:.3e} "
              f"-> {'PASS' if worst < 1e-4 else 'MISMATCH'}", flush=True)
        return
    if what == "torch":
        model = torch.nn.Sequential(torch.nn.Linear(64, 128), torch.nn.LayerNorm(128), torch.nn.ReLU(),
                                    torch.nn.Linear(128, 128), torch.nn.BatchNorm1d(128), torch.nn.ReLU(),
                                    torch.nn.Linear(128, 8)).to(dev).train()
        params = list(model.parameters())
        batches = [(torch.randn(512, 64, device=dev).requires_grad_(True),) for _ in range(3)]

        def fn(x):
            return model(x).square().mean()

        def eager(b):
            return fn(*b)
    else:
        from golden.weights import fill_deterministic
        from test_model_structure import tiny_args

        from coda_neurips2023_amd.dataset_config import HotPathDatasetConfig
        from coda_neurips2023_amd.model_3detr import build_model
        g = np.load(os.path.join(ROOT, "tests", "golden", "model_tiny.npz"))
        model, _ = build_model(tiny_args(), HotPathDatasetConfig())
        fill_deterministic(model, seed=9)
        model.to(dev).train()
        sa_ids = {id(p) for p in model.pre_encoder.parameters()}
        params = [p for p in model.parameters() if id(p) not in sa_ids]
        pc0 = torch.from_numpy(g["pc"]).to(dev)

        def loss_of(st):
            return (st["sem_cls_logits"].square().mean() + st["center_normalized"].mean()
                    + st["size_normalized"].square().mean() + st["angle_logits"].abs().mean()
                    + st["text_correlation_embedding"].square().mean())

        def fn(xyz, feat, inds, pc, dmin, dmax):
            batch = {"point_clouds": pc, "point_cloud_dims_min": dmin, "point_cloud_dims_max": dmax}
            return loss_of(model(batch, pre_encoded=(xyz, feat, inds))["stacked_outputs"])

        batches = []
        for i in range(3):
            pc = (pc0.roll(i * 37, dims=1) * (1.0 + 0.01 * i)).contiguous()
            with torch.no_grad():
                xyz, feat, inds = model.run_pre_encoder(pc)
            batches.append((xyz, feat.clone().requires_grad_(True), inds, pc, pc.amin(1).contiguous(),
                            pc.amax(1).contiguous()))

        def eager(b):
            return fn(*b)

    if eager_first:  # an ordinary eager training step before the capture
        eager(batches[0]).backward()
        for p in params:
            p.grad = None
    ref = []
    for b in batches:
        loss = eager(b)
        grads = torch.autograd.grad(loss, [t for t in b if t.requires_grad] + params, allow_unused=True)
        ref.append((float(loss), [None if g_ is None else g_.clone() for g_ in grads]))
    tail = GraphedTail(fn, list(batches[0]), params)
    worst = 0.0
    for order in ([1, 2, 0], [0, 0, 2, 1], [2, 1, 0]):
        for i in order:
            b = batches[i]
            loss, in_grads = tail.replay(*[t.detach() for t in b])
            lref, gref = ref[i]
            worst = max(worst, abs(float(loss) - lref) / (abs(lref) + 1e-12))
            got = [g_ for g_ in in_grads if g_ is not None] + [p.grad for p in tail._params]
            for a, r in zip(got, gref):
                if r is not None and a is not None:
                    worst = max(worst, float((a - r).abs().max() / (r.abs().max() + 1e-12)))
    tail.close()
    print(f"RESULT {what}:{int(eager_first)} worst relative difference eager vs replay = {worst:.3e} "
          f"-> {'PASS' if worst < 1e-4 else 'MISMATCH'}", flush=True)


def main():
    if len(sys.argv) > 1:
        what, eager_first = sys.argv[1].split(":")
        run_variant(what, eager_first == "1")
        return
    rows = []
    for what in WHAT:
        for eager_first in (0, 1):
            env = dict(os.environ, **ENV[what])
            try:
                r = subprocess.run([sys.executable, os.path.abspath(__file__), f"{what}:{eager_first}"], env=env,
                                   capture_output=True, text=True, timeout=180)
                line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")]
                status = line[0] if line else f"RESULT {what}:{eager_first} exit code {r.returncode} " \
                                              f"({(r.stderr.strip().splitlines() or ['no stderr'])[-1][:160]})"
            except subprocess.TimeoutExpired:
                status = f"RESULT {what}:{eager_first} TIMEOUT"
            rows.append(status)
            print(status, flush=True)


if __name__ == "__main__":
    main()
