"""Which part of the pipelined x3 GEMM kernel bounds it?  CODA_X3_DBG=<mask> builds variants without W loads (1), A loads (2),
MFMAs (4), A split + LDS stores (8), W LDS stores (16); this prints the time of two shapes for the mask in the environment."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coda_neurips2023_amd import gemm  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
for m, n, k in [(98304, 256, 256), (16384, 2048, 256), (16384, 256, 2048)]:
    a = torch.randn(m, k, generator=g).to(dev)
    w = torch.nn.Parameter(torch.randn(n, k, generator=g).to(dev))
    out = torch.empty(m, n, device=dev)
    with torch.no_grad():
        for _ in range(3):
            gemm._run(0, 1, m, n, k, a, w, out, None, False)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            gemm._run(0, 1, m, n, k, a, w, out, None, False)
        e1.record()
        torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 20 * 1e3
    print(f"DBG={os.environ.get('CODA_X3_DBG', '0'):>3} {m} x {n} x {k}: {t:8.1f} us  ({2.0 * m * n * k / t / 1e6:6.1f} TF/s-equivalent)")
