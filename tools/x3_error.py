"""Error of the x3 GEMM and of the library's fp32 GEMM against float64 on the step's shapes: max and RMS, relative to the
largest output; operands as in tests/test_gemm_x3_gpu.py plus activation-like (ReLU-ed, non-zero mean) rows."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coda_neurips2023_amd import gemm  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)


def errs(c, ref):
    d = (c.double() - ref)
    return float(d.abs().max() / ref.abs().max()), float(d.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())


with torch.no_grad():
    for kind in ("normal", "relu"):
        for m, n, k in [(16384, 256, 256), (16384, 768, 256), (16384, 256, 2048), (16384, 2048, 256), (16384, 256, 128)]:
            a = torch.randn(m, k, generator=g, dtype=torch.float64)
            if kind == "relu":
                a = a.clamp_min(0) + 0.1
            w = torch.randn(n, k, generator=g, dtype=torch.float64) / k ** 0.5
            a32, w32 = a.float().to(dev), torch.nn.Parameter(w.float().to(dev))
            ref = a32.double() @ w32.double().t()
            gemm.set_x3(False)
            nat = errs(gemm.linear(a32, w32), ref)
            gemm.set_x3(True, force=True)
            x3 = errs(gemm.linear(a32, w32), ref)
            print(f"{kind:6s} {m} x {n} x {k}: native max {nat[0]:.2e} rms {nat[1]:.2e} | x3 max {x3[0]:.2e} rms {x3[1]:.2e} "
                  f"| ratio max {x3[0] / nat[0]:.2f} rms {x3[1] / nat[1]:.2f}")
