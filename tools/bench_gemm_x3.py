"""The step's large token-wise products two ways: hipBLASLt fp32 (coda_gemm_f32) and the six-product bf16 evaluation
(coda_gemm_x3_nt_f32, weight pieces cached); GPU time per call (HIP events over 20 back-to-back calls)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coda_neurips2023_amd import gemm  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
# (layout, m, n, k): nt = y = x W^T (W n x k), nn = dx = dy W (W k x n)
SHAPES = [("nt", 16384, 768, 256), ("nt", 16384, 256, 256), ("nt", 16384, 128, 256), ("nt", 16384, 256, 128),
          ("nt", 16384, 2048, 256), ("nt", 16384, 512, 256), ("nn", 16384, 256, 768), ("nn", 16384, 256, 2048),
          ("nn", 16384, 256, 256), ("nn", 16384, 128, 256), ("nn", 16384, 256, 128), ("nt", 98304, 256, 256),
          ("nt", 4096, 256, 256)]


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


with torch.no_grad():
    for lay, m, n, k in SHAPES:
        tb = 1 if lay == "nt" else 0
        a = torch.randn(m, k, generator=g).to(dev)
        w = torch.nn.Parameter(torch.randn((n, k) if tb else (k, n), generator=g).to(dev))
        out = torch.empty(m, n, device=dev)
        gemm.set_x3(False)
        t_lib = timed(lambda: gemm._run(0, tb, m, n, k, a, w, out, None, False))
        ref = out.clone()
        gemm.set_x3(True, force=True)
        t_x3 = timed(lambda: gemm._run(0, tb, m, n, k, a, w, out, None, False))
        diff = float((out - ref).abs().max() / ref.abs().max())
        fl = 2.0 * m * n * k
        print(f"{lay} {m:6d} x {n:5d} x {k:6d}: library {t_lib:8.1f} us ({fl / t_lib / 1e6:6.1f} TF/s)   "
              f"x3 {t_x3:8.1f} us ({fl / t_x3 / 1e6:6.1f} TF/s)   {t_lib / t_x3:5.2f}x   max diff {diff:.1e}")


gemm.set_x3(True, force=True, tn=True)
# weight gradients: dW = dY^T X over 16 384 token rows -- the library path (row chunks as one batched GEMM + a sum) against
# the x3 partial sums per token slice + the same sum
with torch.no_grad():
    for t, co, ci in [(16384, 256, 256), (16384, 768, 256), (16384, 128, 256), (16384, 256, 128), (16384, 2048, 256),
                      (16384, 512, 512)]:
        dy = torch.randn(t, co, generator=g).to(dev)
        x = torch.randn(t, ci, generator=g).to(dev)

        def lib():
            return torch.bmm(dy.view(8, t // 8, co).transpose(1, 2), x.view(8, t // 8, ci)).sum(0)

        def x3():
            return gemm.x3_tn_partials(dy, x).sum(0)

        ref = dy.double().t() @ x.double()
        t_lib, t_x3 = timed(lib), timed(x3)
        e_lib = float((lib().double() - ref).abs().max() / ref.abs().max())
        e_x3 = float((x3().double() - ref).abs().max() / ref.abs().max())
        fl = 2.0 * t * co * ci
        print(f"tn {co:5d} x {ci:5d} x {t:6d}: library {t_lib:8.1f} us ({fl / t_lib / 1e6:6.1f} TF/s)   "
              f"x3 {t_x3:8.1f} us ({fl / t_x3 / 1e6:6.1f} TF/s)   {t_lib / t_x3:5.2f}x   err vs f64: lib {e_lib:.1e} x3 {e_x3:.1e} "
              f"slices {gemm.x3_tn_partials(dy, x).shape[0]}")
