#!/usr/bin/env python3
"""Weight-gradient GEMMs dW = dY^T X with a huge reduction dim: torch.mm vs chunked bmm."""
import torch


def t(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


dev = torch.device("cuda:0")
for p, co, ci in [(1048576, 256, 128), (1048576, 128, 64), (16384, 768, 256), (16384, 256, 256), (16384, 128, 256),
                  (16384, 256, 128), (16384, 512, 256)]:
    dy = torch.randn(p, co, device=dev)
    x = torch.randn(p, ci, device=dev)
    ref = torch.mm(dy.t(), x)
    base = t(lambda: torch.mm(dy.t(), x))
    line = f"P={p:8d} {co:4d}x{ci:4d}  mm {base:8.1f} us"
    for rows in (512, 1024, 2048, 4096, 16384):
        if p % rows or rows >= p:
            continue
        nc = p // rows
        fn = lambda: torch.bmm(dy.view(nc, rows, co).transpose(1, 2), x.view(nc, rows, ci)).sum(0)  # noqa: E731
        err = float((fn() - ref).abs().max() / ref.abs().max())
        line += f" | chunk{rows}: {t(fn):7.1f} us (err {err:.0e})"
    print(line, flush=True)
