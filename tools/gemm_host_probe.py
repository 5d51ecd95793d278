#!/usr/bin/env python3
"""Dev probe: host time per launch-sized GEMM, torch.addmm vs coda_gemm_f32 (gemm.linear)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coda_neurips2023_amd import gemm
dev = torch.device("cuda:0")
x = torch.randn(2048, 256, device=dev); w = torch.randn(256, 256, device=dev); b = torch.randn(256, device=dev)
out = torch.empty(2048, 256, device=dev)
def bench(fn, n=2000):
    for _ in range(50): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    th = (time.perf_counter() - t0) / n * 1e6
    torch.cuda.synchronize(); tw = (time.perf_counter() - t0) / n * 1e6
    return "host %.1f us  wall %.1f us" % (th, tw)
print("torch.addmm        ", bench(lambda: torch.addmm(b, x, w.t())))
print("torch.mm           ", bench(lambda: torch.mm(x, w.t())))
print("torch.mm out=      ", bench(lambda: torch.mm(x, w.t(), out=out)))
print("gemm.linear bias   ", bench(lambda: gemm.linear(x, w, b)))
print("gemm.linear        ", bench(lambda: gemm.linear(x, w)))
print("gemm.linear out=   ", bench(lambda: gemm.linear(x, w, None, out)))
print("torch.empty        ", bench(lambda: torch.empty((2048, 256), device=dev)))
