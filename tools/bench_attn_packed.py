"""Decoder cross-attention (256 x 2048) with K / V as column slices of the packed (tokens, 8 E) projections of all
layers (what the decoder stack hands the kernels) against dense K / V.  Per-kernel times (library event timing)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coda_neurips2023_amd import attention_core as core  # noqa: E402

dev = torch.device("cuda:0")
l, s, b, h, d, nl = 256, 2048, 8, 4, 64, 8
q = torch.randn(l, b, h, d, device=dev, requires_grad=True)
go = torch.randn(l, b, h, d, device=dev)
for name in ("dense", "packed (row stride 8E)"):
    if name == "dense":
        k = torch.randn(s, b, h, d, device=dev, requires_grad=True)
        v = torch.randn(s, b, h, d, device=dev, requires_grad=True)
        kk, vv = k, v
    else:
        kp = torch.randn(s, b, nl, h, d, device=dev, requires_grad=True)
        vp = torch.randn(s, b, nl, h, d, device=dev, requires_grad=True)
        kk, vv = kp[:, :, 3], vp[:, :, 3]
    for _ in range(3):
        out, _ = core.attention(q, kk, vv, None, 0.125, 0.1, False)
        out.backward(go)
    torch.cuda.synchronize()
    core.enable_kernel_timing(0)
    for _ in range(20):
        out, _ = core.attention(q, kk, vv, None, 0.125, 0.1, False)
        out.backward(go)
    rec = core.collect_kernel_timing()
    core.disable_kernel_timing()
    line = f"{name:24s}:"
    for kind in ("fwd", "dkv", "dq"):
        ms = sorted(rec[(kind, l, s)])
        line += f"  {kind} {1e3 * ms[len(ms) // 2]:6.1f} us"
    print(line)
