#!/usr/bin/env python3
"""Own fp32-MFMA GEMM kernel (coda_sgemm_f32) vs the library path (coda_gemm_f32 = hipBLASLt)
on the shapes a training step issues (tools/gemm_shapes.py): device time (HIP events over a queued batch of
launches) and host time per call.  Dev tool."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from coda_neurips2023_amd import _lib, gemm  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.load()


def timeit(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    host = (time.perf_counter() - t0) / reps
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3, host * 1e6


def own(transb, a, b, out, bias=None, acc=False):
    m, k = a.shape
    n = b.shape[0] if transb else b.shape[1]
    st = lib.coda_sgemm_f32(1 if transb else 0, m, n, k, a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), out.data_ptr(),
                            out.stride(0), bias.data_ptr() if bias is not None else None, 1 if acc else 0,
                            _lib.current_stream_handle())
    assert st == 0, st


for (m, n, k) in [(2048, 256, 256), (2048, 512, 256), (16384, 256, 256), (16384, 768, 256), (16384, 128, 256),
                  (16384, 256, 128), (16384, 2048, 256), (16384, 256, 2048)]:
    a = torch.randn(m, k, device=dev)
    w_nt = torch.randn(n, k, device=dev)
    w_nn = torch.randn(k, n, device=dev)
    bias = torch.randn(n, device=dev)
    out = torch.empty(m, n, device=dev)
    ref = (a.double() @ w_nt.double().t() + bias.double()).float()
    own(True, a, w_nt, out, bias)
    e1 = float((out - ref).abs().max() / ref.abs().max())
    own(False, a, w_nn, out)
    e2 = float((out - (a.double() @ w_nn.double()).float()).abs().max() / ref.abs().max())
    flops = 2.0 * m * n * k
    t_own_nt, h_own = timeit(lambda: own(True, a, w_nt, out, bias))
    t_lib_nt, h_lib = timeit(lambda: gemm._run(0, 1, m, n, k, a, w_nt, out, bias, False))
    t_own_nn, _ = timeit(lambda: own(False, a, w_nn, out))
    t_lib_nn, _ = timeit(lambda: gemm._run(0, 0, m, n, k, a, w_nn, out, None, False))
    print(f"{m:6d}x{n:4d}x{k:4d}  NT own {t_own_nt:7.1f} us ({flops / t_own_nt / 1e6:5.1f} TF/s) lib {t_lib_nt:7.1f} us | "
          f"NN own {t_own_nn:7.1f} lib {t_lib_nn:7.1f} | host own {h_own:4.1f} lib {h_lib:4.1f} us | err {e1:.1e} {e2:.1e}")

print("--- weight gradients: out (co x ci) = dy^T x")
for (rows, co, ci) in [(2048, 256, 256), (2048, 512, 256), (16384, 256, 256), (16384, 768, 256), (16384, 128, 256)]:
    dy = torch.randn(rows, co, device=dev)
    x = torch.randn(rows, ci, device=dev)
    ref = (dy.double().t() @ x.double()).float()
    e = float((gemm.mm_tn(dy, x) - ref).abs().max() / ref.abs().max())
    t_l, h_l = timeit(lambda: gemm.mm_tn(dy, x))  # one library GEMM (the own TN kernel was removed in round 4)
    from coda_neurips2023_amd.linear_fn import tn_gemm
    t_c, h_c = timeit(lambda: tn_gemm(dy, x))     # row chunks + a sum: what the step uses for long reductions
    print(f"{rows:6d} {co:4d}x{ci:4d}  gemm.mm_tn {t_l:7.1f} us  linear_fn.tn_gemm {t_c:7.1f} us | host {h_l:4.1f} {h_c:4.1f} | err {e:.1e}")
