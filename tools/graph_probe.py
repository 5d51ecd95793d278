#!/usr/bin/env python3
"""Dev probe: host and wall cost per kernel of a hipGraph replay vs eager launches on this stack
(1000 launch-sized kernels: torch elementwise adds and coda add_ln calls)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coda_neurips2023_amd import fused_layers

dev = torch.device("cuda:0")
x = torch.randn(2048, 256, device=dev)
norm = torch.nn.LayerNorm(256).to(dev)


def body(n):
    y = x
    for i in range(n // 2):
        y = y + 1.0
        y = fused_layers.add_ln(y, norm)[1]
    return y


def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    t_host = (time.perf_counter() - t0) / reps
    torch.cuda.synchronize()
    t_wall = (time.perf_counter() - t0) / reps
    return t_host * 1e3, t_wall * 1e3


N = 1000
with torch.no_grad():
    print("eager   host/wall ms per %d kernels: %.2f / %.2f" % ((N,) + timeit(lambda: body(N))))
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        body(N)
    torch.cuda.current_stream().wait_stream(s)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        out = body(N)
    print("replay  host/wall ms per %d kernels: %.2f / %.2f" % ((N,) + timeit(gr.replay)))
