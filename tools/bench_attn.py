#!/usr/bin/env python3
"""Attention kernel timings on the model's shapes (HIP events, same process / same box). Dev tool."""
import sys, os
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from coda_neurips2023_amd import attention_core as core

dev = torch.device("cuda:0")


def run(l, s, b=8, h=4, d=64, p=0.1, reps=20):
    q = torch.randn(l, b, h, d, device=dev, requires_grad=True)
    k = torch.randn(s, b, h, d, device=dev, requires_grad=True)
    v = torch.randn(s, b, h, d, device=dev, requires_grad=True)
    go = torch.randn(l, b, h, d, device=dev)
    for _ in range(3):
        out, _ = core.attention(q, k, v, None, 0.125, p, False)
        out.backward(go)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    tf = tb = 0.0
    for _ in range(reps):
        ev[0].record()
        out, _ = core.attention(q, k, v, None, 0.125, p, False)
        ev[1].record()
        out.backward(go)
        ev[2].record()
        torch.cuda.synchronize()
        tf += ev[0].elapsed_time(ev[1])
        tb += ev[1].elapsed_time(ev[2])
    flops = 4.0 * l * s * d * b * h
    print(f"L={l} S={s} p={p}: fwd {1e3 * tf / reps:7.1f} us ({flops / (tf / reps) / 1e9:5.1f} TF/s)  "
          f"bwd {1e3 * tb / reps:7.1f} us ({2.5 * flops / (tb / reps) / 1e9:5.1f} TF/s useful)")


for shape in [(2048, 2048), (256, 2048), (256, 256)]:
    run(*shape)
