"""Is the x3 GEMM's error biased?  Mean SIGNED error (in units of the RMS error and relative to the mean |y|) of the x3
and the library fp32 GEMM against float64, on outputs that are all positive, all negative, and mixed."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coda_neurips2023_amd import gemm  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
with torch.no_grad():
    for k in (128, 256, 1024, 4096):
        for kind in ("pos", "neg", "mixed"):
            m, n = 16384, 256
            a = torch.randn(m, k, generator=g, dtype=torch.float64)
            w = torch.randn(n, k, generator=g, dtype=torch.float64)
            if kind != "mixed":
                a, w = a.abs(), w.abs() * (1 if kind == "pos" else -1)
            a32, w32 = a.float().to(dev), torch.nn.Parameter(w.float().to(dev))
            ref = a32.double() @ w32.double().t()
            scale = float(ref.abs().mean())
            out = {}
            gemm.set_x3(False)
            out["lib"] = gemm.linear(a32, w32).double() - ref
            gemm.set_x3(True, force=True)
            out["x3"] = gemm.linear(a32, w32).double() - ref
            msg = []
            for name, d in out.items():
                msg.append(f"{name}: mean {float(d.mean()) / scale:+.2e} rms {float(d.pow(2).mean().sqrt()) / scale:.2e} "
                           f"colsum-err {float(d.sum(0).abs().max()) / scale / m:.2e}")
            print(f"k={k:5d} {kind:5s} " + " | ".join(msg))
