#!/usr/bin/env python3
"""Dev probe: the decoder's memory key/value projections as ONE GEMM into a packed (tokens, 8E) buffer (today) vs
EIGHT GEMMs into a layer-major (8, tokens, E) buffer, forward and the three backward products."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from coda_neurips2023_amd import gemm  # noqa: E402
from coda_neurips2023_amd.linear_fn import tn_gemm  # noqa: E402

dev = torch.device("cuda:0")
R, E, NL = 16384, 256, 8
x = torch.randn(R, E, device=dev)
w = torch.randn(NL * E, E, device=dev)
b = torch.randn(NL * E, device=dev)
dk = torch.randn(R, NL * E, device=dev)
dk_lm = torch.randn(NL, R, E, device=dev)
k_lm = torch.empty(NL, R, E, device=dev)
dx = torch.empty(R, E, device=dev)


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) * 1e3)
    return np.median(ts)


def fwd_split():
    for l in range(NL):
        gemm.linear(x, w[l * E:(l + 1) * E], b[l * E:(l + 1) * E], out=k_lm[l])


def dx_split():
    for l in range(NL):
        gemm.mm(dk_lm[l], w[l * E:(l + 1) * E], out=dx, accumulate=l > 0)


def dw_split():
    return [tn_gemm(dk_lm[l], x) for l in range(NL)]


print(f"forward  packed {timeit(lambda: gemm.linear(x, w, b)):7.1f} us   layer-major 8x {timeit(fwd_split):7.1f} us")
print(f"d input  packed {timeit(lambda: gemm.mm(dk, w)):7.1f} us   layer-major 8x {timeit(dx_split):7.1f} us")
print(f"d weight packed {timeit(lambda: tn_gemm(dk, x)):7.1f} us   layer-major 8x {timeit(dw_split):7.1f} us")
