"""fp32 GEMM three ways on the step's large shapes: hipBLASLt (coda_gemm_f32), the nine-product bf16 evaluation
(coda_gemm_x3_f32), torch.mm; GPU time per call (HIP events over 20 back-to-back calls)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coda_neurips2023_amd import gemm  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
SHAPES = [("nt", 16384, 768, 256), ("nt", 16384, 256, 256), ("nt", 16384, 128, 256), ("nt", 16384, 256, 128),
          ("nt", 16384, 2048, 256), ("nn", 16384, 256, 768), ("nn", 16384, 256, 2048), ("nn", 16384, 256, 256),
          ("tn", 256, 256, 16384), ("tn", 768, 256, 16384), ("tn", 2048, 256, 16384), ("nt", 2048, 256, 256),
          ("nt", 98304, 256, 256)]


def timed(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for lay, m, n, k in SHAPES:
    ta, tb = {"nt": (0, 1), "nn": (0, 0), "tn": (1, 0)}[lay]
    a = torch.randn((k, m) if ta else (m, k), generator=g).to(dev)
    b = torch.randn((n, k) if tb else (k, n), generator=g).to(dev)
    out = torch.empty(m, n, device=dev)
    gemm.set_x3(False)
    t_lib = timed(lambda: gemm._run(ta, tb, m, n, k, a, b, out, None, False))
    t_x3 = timed(lambda: gemm.gemm_x3(ta, tb, m, n, k, a, b, out))
    fl = 2.0 * m * n * k
    print(f"{lay} {m:6d} x {n:5d} x {k:6d}: library {t_lib:8.1f} us ({fl / t_lib / 1e6:6.1f} TF/s)   "
          f"x3 {t_x3:8.1f} us ({fl / t_x3 / 1e6:6.1f} TF/s)   {t_lib / t_x3:5.2f}x")
