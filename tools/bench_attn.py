#!/usr/bin/env python3
"""Attention kernel timings on the model's shapes, per kernel (HIP events around each launch inside the
library, coda_mha_timing_*), for both MFMA operand types.  Dev tool.

    python tools/bench_attn.py [fp32|bf16|both] [dropout_p]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from coda_neurips2023_amd import attention_core as core  # noqa: E402

dev = torch.device("cuda:0")
# executed MFMA flops per (l * s * d * b * h); dqg: dQ = dS K alone; bwdf: S, dP, dV, dK, dQ in one kernel
FLOPS = {"fwd": 4, "dkv": 8, "dq": 6, "dqg": 2, "bwdf": 10}


def run(l, s, b=8, h=4, d=64, p=0.1, reps=20):
    q = torch.randn(l, b, h, d, device=dev, requires_grad=True)
    k = torch.randn(s, b, h, d, device=dev, requires_grad=True)
    v = torch.randn(s, b, h, d, device=dev, requires_grad=True)
    go = torch.randn(l, b, h, d, device=dev)
    for _ in range(3):
        out, _ = core.attention(q, k, v, None, 0.125, p, False)
        out.backward(go)
    torch.cuda.synchronize()
    core.enable_kernel_timing(0)
    for _ in range(reps):
        out, _ = core.attention(q, k, v, None, 0.125, p, False)
        out.backward(go)
    rec = core.collect_kernel_timing()
    core.disable_kernel_timing()
    line = f"L={l:5d} S={s:5d} d={d:3d} p={p}:"
    for kind in core.TIMING_KINDS:
        if (kind, l, s) not in rec:   # delta: formed inside the dQ kernel on the fp32 path
            continue
        ms = sorted(rec[(kind, l, s)])
        med = ms[len(ms) // 2]
        line += f"  {kind} {1e3 * med:7.1f} us"
        if kind in FLOPS:
            line += f" ({FLOPS[kind] * l * s * d * b * h / med / 1e9:6.1f} TF/s)"
    print(line)


which = sys.argv[1] if len(sys.argv) > 1 else "both"
p = float(sys.argv[2]) if len(sys.argv) > 2 else 0.1
for dt in (["fp32", "bf16", "bf16x3"] if which == "both" else [which]):
    print(f"--- MFMA operands: {dt}")
    with core.mfma_dtype(dt):
        for shape in [(2048, 2048), (256, 2048), (256, 256), (512, 2048), (512, 512)]:
            run(*shape, p=p)
        run(256, 2048, d=128, h=4, p=p)
        run(128, 2048, d=128, h=4, p=p)   # the scripts' variant: dec_dim 512, 128 queries
        run(128, 128, d=128, h=4, p=p)
