"""dev: how far the library GEMMs of the query projection (GenericMLP: Linear + ReLU + Linear + ReLU on 8 x nq tokens) are
from float64, and how many ReLU decisions differ -- for the configs[3] share (dec_dim 512, 128 queries) whose
query_projection gradients moved by 5e-3 when the library's kernel choice changed (tests/test_full_step_gpu.py)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from golden import step_inputs as SI  # noqa: E402
from golden.weights import fill_deterministic  # noqa: E402
from coda_neurips2023_amd.dataset_config import HotPathDatasetConfig  # noqa: E402
from coda_neurips2023_amd.model_3detr import build_model  # noqa: E402

dev = torch.device("cuda:0")
for case in ("configs2", "configs3"):
    batch, seam = SI.build(case)
    args = SI.recipe(case)
    model, _ = build_model(args, HotPathDatasetConfig(), text_features_fg_norm=seam["text"].to(dev))
    fill_deterministic(model, seed=SI.WEIGHT_SEED)
    model.to(dev).train()
    qp = model.query_projection
    seen = {}
    real = qp.forward_tokens

    def spy(x):
        seen["x"] = x.detach()
        return real(x)

    qp.forward_tokens = spy
    with torch.no_grad():
        model({k: v.to(dev) for k, v in batch.items()}, curr_epoch=0)
    x = seen["x"]
    lin = [m for m in qp.layers if isinstance(m, (torch.nn.Conv1d, torch.nn.Linear))]
    w = [m.weight.squeeze(-1) for m in lin]
    b = [m.bias for m in lin]
    xd = x.double().cpu()
    z1 = torch.nn.functional.linear(x, w[0], b[0])
    z1d = torch.nn.functional.linear(xd, w[0].double().cpu(), b[0].double().cpu())
    h, hd = torch.relu(z1), torch.relu(z1d)
    z2 = torch.nn.functional.linear(h, w[1], b[1])
    z2d = torch.nn.functional.linear(hd, w[1].double().cpu(), b[1].double().cpu())
    z2m = torch.nn.functional.linear(hd.float().to(dev), w[1], b[1])  # layer 2 alone on the exact hidden
    for name, g, d in (("layer 1", z1, z1d), ("layer 2", z2, z2d), ("layer 2 (exact input)", z2m, z2d)):
        err = (g.double().cpu() - d).abs()
        flips = ((g.cpu() > 0) != (d > 0)).sum().item()
        print(f"{case} {tuple(x.shape)} {name}: max |err| {err.max():.3e} (rms of the values {d.pow(2).mean().sqrt():.3e}), "
              f"relative L2 {err.norm() / d.norm():.3e}, ReLU decisions that differ: {flips} of {d.numel()}")
